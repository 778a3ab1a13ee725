"""The tail of the reference's bulk driver, on the device engine:
`perform_inference_and_write_output` (src/bulk.rs:82-209) from the built store onwards --
EMInfo assembly, `em` / `em_par` by thread count (:155-159), aux counts (:161), `.quant` /
`.ambig_info.tsv` / `.meta_info.json` (:168-174), bootstraps -> `.infreps.pq` (:178-193),
assignment probabilities -> `.prob` (:196-207).  Alignment parsing, filtering and the coverage
model come before this (oem_builder_* / the caller); KDE is not supported.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import writers
from .em import bootstrap, em, em_par
from .types import EMInfo, InMemoryAlignmentStore, TranscriptInfo


def read_short_quant_vec(short_read_path: str, txps_name: Sequence[str]) -> np.ndarray:
    """`--short-quant` (util/read_function.rs:9-85): a salmon-style `quant.sf` TSV (columns Name, Length,
    EffectiveLength, TPM, NumReads) projected onto the transcript order of the header -> the EM's
    `init_abundances` (bulk.rs:118-120).  A name in the file that the header lacks is an error; a
    header transcript missing from the file gets 0."""
    import csv
    with open(short_read_path, newline="") as fh:
        rdr = csv.DictReader(fh, delimiter="\t")
        need = {"Name", "Length", "EffectiveLength", "TPM", "NumReads"}
        if rdr.fieldnames is None or not need.issubset(rdr.fieldnames):
            raise ValueError(f"{short_read_path}: expected the columns {sorted(need)}")
        records = {}
        for rec in rdr:
            int(rec["Length"]); float(rec["EffectiveLength"]); float(rec["TPM"])   # deserialised (and checked) as in the reference
            records[rec["Name"]] = float(rec["NumReads"])
    names = set(txps_name)
    if not all(k in names for k in records):
        raise ValueError("There were transcripts in the short read quantification file that didn't appear in "
                         "the BAM header; cannot proceed.")
    return np.array([records.get(n, 0.0) for n in txps_name], dtype=np.float64)


@dataclass
class BulkArgs:
    """The `Args` fields this stage reads (prog_opts.rs): defaults are the reference's."""
    output: str
    threads: int = 3                      # prog_opts.rs:543
    max_em_iter: int = 1000               # :532
    convergence_thresh: float = 1e-3      # :536
    num_bootstraps: int = 0
    write_assignment_probs: bool = False
    display_thresh: float = 1e-6
    seed: int = 0                         # the reference seeds from the OS (em.rs:274)
    device: int = 0
    extra_info: dict = field(default_factory=dict)


def perform_inference_and_write_output(store: InMemoryAlignmentStore, txps_name: Sequence[str],
                                       txp_lens: Sequence[int], args: BulkArgs,
                                       init_abundances: Optional[np.ndarray] = None,
                                       read_names: Optional[Sequence[str]] = None) -> np.ndarray:
    txps = [TranscriptInfo.with_len(int(l)) for l in txp_lens]
    emi = EMInfo(eq_map=store, txp_info=txps, max_iter=args.max_em_iter,
                 convergence_thresh=args.convergence_thresh, init_abundances=init_abundances,
                 device=args.device)
    counts = em_par(emi, args.threads) if args.threads > 4 else em(emi, args.threads)   # bulk.rs:155-159
    dev = store.device_store(len(txps), args.device)
    unique, total = dev.aux_counts()                                                     # bulk.rs:161
    info = {"num_aligned_reads": store.num_aligned_reads(), "total_alignments": store.total_len(),
            "em_max_iter": args.max_em_iter, "em_convergence_thresh": args.convergence_thresh,
            "threads": args.threads, "num_bootstraps": args.num_bootstraps, "output": args.output,
            "filter_options": {"model_coverage": store.filter_opts.model_coverage},
            "em_iterations": emi.last_run_info.niter if emi.last_run_info else None}
    info.update(args.extra_info)
    writers.write_output(args.output, info, txps_name, txp_lens, counts, unique, total)  # bulk.rs:168-174
    if args.num_bootstraps > 0:                                                          # bulk.rs:178-193
        breps = bootstrap(emi, args.num_bootstraps, args.threads, seed=args.seed)
        writers.write_infrep_file(args.output, breps)
    if args.write_assignment_probs:                                                      # bulk.rs:196-207
        if read_names is None:
            raise ValueError("cannot write assignment probabilities without valid vector of read names")
        probs = dev.assignment_probs(counts, args.display_thresh)
        writers.write_out_prob(args.output, store.boundaries, store.alignments, probs, read_names, txps_name,
                               args.display_thresh)
    return counts
