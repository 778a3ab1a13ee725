"""Result files in the reference's on-disk formats.

Host-side formatting only; every number written here comes out of the device engine
(`em` / `em_par` / `bootstrap` / `DeviceStore.aux_counts` / `DeviceStore.assignment_probs`).

  write_output               write_function.rs:72-148   <out>.meta_info.json, .quant, .ambig_info.tsv
  write_infrep_file          write_function.rs:199-209, parquet_utils.rs:15-44, bulk.rs:181-193
  write_out_prob             write_function.rs:226-340  <out>.prob
  write_single_cell_output   write_function.rs:25-69    <out>.count.mtx, .features.txt (+ .barcodes.txt,
                             single_cell.rs:176-178)

Numbers are printed the way Rust's `{}` prints them (shortest digits that round-trip, never an
exponent, no trailing ".0" -- `rust_display`), so a `.quant` written here is byte-identical to the
reference's for equal counts.
"""
from __future__ import annotations

import json
import math
import os
from typing import Iterable, Optional, Sequence

import numpy as np


def rust_display(x, f32: bool = False) -> str:
    """`format!("{}", x)` of an f64 (or f32): shortest round-trip digits, positional notation."""
    x = np.float32(x) if f32 else np.float64(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def with_additional_extension(output: str, ext: str) -> str:
    """path_tools::WithAdditionalExtension: append, never replace."""
    return str(output) + ext


def _make_parent(output: str) -> None:  # write_function.rs:32-40
    parent = os.path.dirname(str(output))
    if parent:
        os.makedirs(parent, exist_ok=True)


def write_output(output: str, info: dict, names: Sequence[str], lens: Sequence[int], counts,
                 unique_counts, total_counts) -> None:
    """write_function.rs:72-148: `.meta_info.json`, `.quant` (tname, len, num_reads) and
    `.ambig_info.tsv` (unique, ambig = total - unique saturating, total)."""
    if not (len(names) == len(lens) == len(counts) == len(unique_counts) == len(total_counts)):
        raise ValueError("write_output: per-transcript columns differ in length")
    _make_parent(output)
    with open(with_additional_extension(output, ".meta_info.json"), "w") as fh:
        json.dump(info, fh, indent=2)                       # serde_json to_writer_pretty
    with open(with_additional_extension(output, ".quant"), "w") as fh:
        fh.write("tname\tlen\tnum_reads\n")
        for n, l, c in zip(names, lens, counts):
            fh.write(f"{n}\t{int(l)}\t{rust_display(c)}\n")
    with open(with_additional_extension(output, ".ambig_info.tsv"), "w") as fh:
        fh.write("unique_reads\tambig_reads\ttotal_reads\n")
        for u, t in zip(unique_counts, total_counts):
            u, t = int(u), int(t)
            fh.write(f"{u}\t{max(t - u, 0)}\t{t}\n")


def write_infrep_file(output: str, breps) -> str:
    """bulk.rs:181-193 + parquet_utils.rs:15-44: one non-nullable f64 column `bootstrap.{i}` per
    replicate, one row per transcript; zstd, format v2, plain encoding, statistics on."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    breps = np.asarray(breps, dtype=np.float64)
    if breps.ndim != 2:
        raise ValueError("breps must be n_boot x n_txps")
    fields = [pa.field(f"bootstrap.{i}", pa.float64(), nullable=False) for i in range(breps.shape[0])]
    table = pa.Table.from_arrays([pa.array(np.ascontiguousarray(b)) for b in breps], schema=pa.schema(fields))
    path = with_additional_extension(output, ".infreps.pq")
    _make_parent(output)
    pq.write_table(table, path, compression="zstd", version="2.6", use_dictionary=False,
                   write_statistics=True, data_page_version="2.0")
    return path


def prob_display_decimals(display_thresh: float) -> int:
    """write_function.rs:218-224: ceil(-log10(thresh)) clamped to [3, 9]; 9 for degenerate input."""
    if display_thresh > 0.0 and math.isfinite(display_thresh):
        return int(min(max(math.ceil(-math.log10(display_thresh)), 3.0), 9.0))
    return 9


def write_out_prob(output: str, row_ptr, tid, probs, read_names: Iterable[str], txp_names: Sequence[str],
                   display_thresh: float, compressed: bool = False) -> str:
    """write_function.rs:226-340.  `probs` is what `DeviceStore.assignment_probs(counts,
    display_thresh)` returns: the renormalised probability of every printed alignment, -1 for
    the ones below the threshold (the arithmetic of :283-318 runs on the device)."""
    if compressed:
        raise NotImplementedError(".prob.lz4 needs an lz4 frame encoder, which this image lacks")
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    tid = np.asarray(tid)
    probs = np.asarray(probs, dtype=np.float64)
    d = prob_display_decimals(display_thresh)
    path = with_additional_extension(output, ".prob")
    _make_parent(output)
    with open(path, "w", buffering=1 << 20) as fh:
        fh.write(f"{len(txp_names)}\t{len(row_ptr) - 1}\n")
        for t in txp_names:
            fh.write(f"{t}\n")
        for r, name in enumerate(read_names):
            b, e = row_ptr[r], row_ptr[r + 1]
            keep = probs[b:e] >= 0.0
            ids = tid[b:e][keep]
            pv = probs[b:e][keep]
            fh.write(f"{name.rstrip(chr(0))}\t{len(ids)}\t" + "\t".join(str(int(x)) for x in ids) + "\t"
                     + "\t".join(f"{x:.{d}f}" for x in pv) + "\n")
    return path


def cell_triplets(cell_counts) -> tuple:
    """single_cell.rs:155-160: per cell keep v > 0 as f32 -> (row_ids, col_ids, vals) of the
    cells x transcripts count matrix."""
    cell_counts = np.asarray(cell_counts, dtype=np.float64)
    rows, cols = np.nonzero(cell_counts > 0.0)
    return rows.astype(np.uint32), cols.astype(np.uint32), cell_counts[rows, cols].astype(np.float32)


def write_single_cell_output(output: str, info: dict, feature_names: Sequence[str], barcodes: Optional[Sequence[str]],
                             n_cells: int, row_ids, col_ids, vals) -> None:
    """write_function.rs:25-69: `.meta_info.json`, `.count.mtx` (MatrixMarket coordinate real
    general, 1-based, f32 values in triplet order as sprs::io::write_matrix_market emits them) and
    `.features.txt`; `.barcodes.txt` in row order (single_cell.rs:176-178)."""
    _make_parent(output)
    with open(with_additional_extension(output, ".meta_info.json"), "w") as fh:
        json.dump(info, fh, indent=2)
    with open(with_additional_extension(output, ".count.mtx"), "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate real general\n% written by sprs\n")
        fh.write(f"{int(n_cells)} {len(feature_names)} {len(vals)}\n")
        for r, c, v in zip(row_ids, col_ids, vals):
            fh.write(f"{int(r) + 1} {int(c) + 1} {rust_display(v, f32=True)}\n")
    with open(with_additional_extension(output, ".features.txt"), "w") as fh:
        for n in feature_names:
            fh.write(f"{n}\n")
    if barcodes is not None:
        with open(with_additional_extension(output, ".barcodes.txt"), "w") as fh:
            for b in barcodes:
                fh.write(f"{b}\n")
