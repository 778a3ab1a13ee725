"""Seeded synthetic alignment stores for the BASELINE.json configurations.

Generator of SURVEY.md section 8d / BASELINE.md section 4 (there is no BAM or
read data in the reference's test_data/, and no network):

  * true abundance  a_t ~ LogNormal(0, 2), normalised;
  * transcripts grouped into "genes" of size 1 + Geom(1/4), contiguous ids;
  * per read: primary t0 ~ Categorical(a); k = clip(1 + Poisson(kbar-1), 1, 100)
    (cap = --best-n default 100, prog_opts.rs:428); the other k-1 targets are 80 %
    from the primary's gene and 20 % uniform over all transcripts; duplicates within
    a read are dropped (targets are distinct);
  * weights follow oarfish_types.rs:1107-1113: p = expf((s - best) / 5) as **f32**,
    with d = best - s in {0, 1, ...}: d = 0 for the primary, d ~ Geom(0.15) for the
    others, truncated so that s / best >= 0.95 (default score threshold,
    prog_opts.rs:458) for a best score drawn uniformly from [500, 3000];
  * optional coverage column: positive f64, normalised to sum 1 per read, as
    normalize_probability.rs:61-69 leaves it.

Rows come out in generation order (no sorting).  Reads are generated in fixed
chunks of ``CHUNK`` reads, chunk c from ``default_rng([seed, c])``, so a store is
a pure function of (seed, n_reads, n_txps, kbar, coverage).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Optional

import numpy as np

CHUNK = 1 << 18
BASE_SEED = 20260928


@dataclass
class SyntheticStore:
    row_ptr: np.ndarray          # u64 [R+1]
    tid: np.ndarray              # u32 [nnz]
    as_prob: np.ndarray          # f32 [nnz]
    cov_prob: Optional[np.ndarray]  # f64 [nnz] or None
    n_txps: int
    abundance: np.ndarray        # f64 [T] ground truth (sums to 1)
    gene_of: np.ndarray          # i32 [T]

    @property
    def n_reads(self) -> int:
        return len(self.row_ptr) - 1

    @property
    def nnz(self) -> int:
        return len(self.tid)


def _genes(n_txps: int, rng: np.random.Generator):
    sizes = []
    tot = 0
    while tot < n_txps:
        s = 1 + rng.geometric(0.25, size=max(1024, n_txps // 3)) - 1  # 1 + Geom(1/4) on {0,1,..}
        sizes.append(s)
        tot += int(s.sum())
    sizes = np.concatenate(sizes)
    ends = np.cumsum(sizes)
    n_genes = int(np.searchsorted(ends, n_txps, side="left")) + 1
    sizes = sizes[:n_genes].copy()
    sizes[-1] -= int(ends[n_genes - 1]) - n_txps
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    gene_of = np.repeat(np.arange(n_genes, dtype=np.int32), sizes)
    return starts, sizes.astype(np.int64), gene_of


def _families(n_genes: int, rng: np.random.Generator, adjacent: bool):
    """Paralog families of three genes: consecutive triples of a random permutation of the genes (scattered over the
    annotation, as an annotation numbers them), or of the genes in id order (``adjacent``: the numbering a co-mapping
    renumbering at store creation would produce).  Returns (order, pos): family f = order[3 f : 3 f + 3], pos = inverse."""
    order = np.arange(n_genes, dtype=np.int64) if adjacent else rng.permutation(n_genes).astype(np.int64)
    pos = np.empty(n_genes, dtype=np.int64)
    pos[order] = np.arange(n_genes, dtype=np.int64)
    return order, pos


def _chunk(c: int, n: int, seed: int, n_txps: int, kbar: float, cdf, g_start, g_size, gene_of,
           coverage: bool, gaps: str = "geometric", fam=None):
    rng = np.random.default_rng([seed, c])
    # primary transcript ~ Categorical(a)
    t0 = np.searchsorted(cdf, rng.random(n), side="right").astype(np.int64)
    np.minimum(t0, n_txps - 1, out=t0)
    k = np.clip(1 + rng.poisson(max(kbar - 1.0, 0.0), size=n), 1, 100).astype(np.int64)
    k = np.minimum(k, n_txps)
    tot = int(k.sum())
    row = np.repeat(np.arange(n, dtype=np.int64), k)
    first = np.concatenate([[0], np.cumsum(k)[:-1]])
    is_primary = np.zeros(tot, dtype=bool)
    is_primary[first] = True
    # targets: slot 0 is the primary; of the other k-1 slots Binomial(k-1, 0.8) are
    # "same gene" slots, the rest uniform over all transcripts.  Same-gene slots walk
    # the gene's other members from a random rotation (distinct by construction) and,
    # once the gene is exhausted, spill to the ids that follow it (neighbouring
    # genes), so reads keep their k alignments and their locality.
    slot = np.arange(tot, dtype=np.int64) - first[row]
    n_gene = rng.binomial(k - 1, 0.8)[row]
    is_gene = (slot >= 1) & (slot <= n_gene)
    g = gene_of[t0][row]
    gs, gz = g_start[g], g_size[g]
    i = slot - 1
    rot = (rng.random(n) * 1e9).astype(np.int64)[row]
    others = np.maximum(gz - 1, 1)
    member = (t0[row] - gs + 1 + (rot + i) % others) % gz
    spill = gs + gz + (i - (gz - 1))
    spill = np.where(spill >= n_txps, gs - 1 - (spill - n_txps), spill)
    in_gene = np.where(i < gz - 1, gs + member, spill)
    in_gene = np.clip(in_gene, 0, n_txps - 1)
    if fam is None:
        anywhere = rng.integers(0, n_txps, size=tot)
    else:
        # far hits recur: a transcript of another gene of the read's paralog family (three genes), not anywhere
        order, pos = fam
        n_genes = len(order)
        ps = pos[g]
        # (the last family is incomplete when the gene count is no multiple of three: the index wraps inside the family's
        # real size, so its reads still draw their far hit from ANOTHER gene -- a family of one has only itself)
        fsize = np.minimum(3, n_genes - 3 * (ps // 3))
        step = np.where(fsize == 3, 1 + rng.integers(0, 2, size=tot), 1)
        other = 3 * (ps // 3) + (ps % 3 + step) % fsize
        g2 = order[other]
        anywhere = g_start[g2] + (rng.random(tot) * g_size[g2]).astype(np.int64)
    t = np.where(is_gene, in_gene, anywhere)
    t[is_primary] = t0
    # score deficits d (best - s): 0 for the primary, truncated geometric otherwise
    if gaps == "uniform":
        # Long reads: a best score anywhere in [500, 20 000] and the other alignments' deficits UNIFORM on everything
        # the reference's score_threshold of 0.95 lets through (oarfish_types.rs:1107-1118) -- up to 1000 distinct
        # integer gaps, of which exp(-gap / 5) keeps ~520 apart in f32 before it reaches 0: the store with more than
        # 256 distinct weights that the byte-coded weight table cannot take (oem_layout_dict.hip: 16-bit indices).
        best = rng.integers(500, 20001, size=n)
        dmax = np.floor(0.05 * best).astype(np.int64)[row]
        d = np.floor(rng.random(tot) * (dmax + 1)).astype(np.int64)
    else:
        best = rng.integers(500, 3001, size=n)
        dmax = np.floor(0.05 * best).astype(np.int64)[row]
        d = np.minimum(rng.geometric(0.15, size=tot) - 1, dmax)
    d[is_primary] = 0
    # distinct targets per read: keep the smallest deficit of each (row, tid)
    order = np.lexsort((d, t, row))
    row, t, d = row[order], t[order], d[order]
    keep = np.ones(tot, dtype=bool)
    keep[1:] = (row[1:] != row[:-1]) | (t[1:] != t[:-1])
    row, t, d = row[keep], t[keep], d[keep]
    # oarfish_types.rs:1112-1113: ((fscore - mscore) / score_prob_denom).exp() in f32
    p = np.exp((-d.astype(np.float32)) / np.float32(5.0)).astype(np.float32)
    lens = np.bincount(row, minlength=n).astype(np.uint64)
    cov = None
    if coverage:
        cov = rng.uniform(0.05, 1.0, size=len(t))
        s = np.add.reduceat(cov, np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64))
        cov = cov / np.repeat(s, lens.astype(np.int64))
    return lens, t.astype(np.uint32), p, cov


def make_store(n_reads: int, n_txps: int, kbar: float = 8.0, seed: int = BASE_SEED,
               coverage: bool = False, threads: int = 8, gaps: str = "geometric", far: str = "uniform") -> SyntheticStore:
    """``gaps``: "geometric" (score deficits Geom(0.15), best score 500..3000 -- ~100 distinct
    weights) or "uniform" (long reads: deficits uniform on [0, 0.05 best], best score 500..20 000 -- ~520 distinct
    weights; see _chunk).
    ``far``: where a read's alignments outside its gene go (20 % of the non-primary ones): "uniform" -- anywhere in the
    annotation, SURVEY.md section 8d's generator and the BASELINE headline; "paralog" -- a transcript of another gene of
    the read's paralog FAMILY (three genes scattered over the annotation): far hits recur, as multi-mapping reads'
    do; "paralog_adjacent" -- the same families numbered next to each other (the annotation a co-mapping renumbering
    of the transcripts at store creation would produce)."""
    if gaps not in ("geometric", "uniform"):
        raise ValueError("gaps must be 'geometric' or 'uniform'")
    if far not in ("uniform", "paralog", "paralog_adjacent"):
        raise ValueError("far must be 'uniform', 'paralog' or 'paralog_adjacent'")
    rng0 = np.random.default_rng([seed, 0xA11CE])
    a = rng0.lognormal(0.0, 2.0, size=n_txps)
    a /= a.sum()
    cdf = np.cumsum(a)
    cdf /= cdf[-1]
    g_start, g_size, gene_of = _genes(n_txps, rng0)
    fam = None if far == "uniform" else _families(len(g_start), np.random.default_rng([seed, 0xFA111E5]), far == "paralog_adjacent")
    n_chunks = (n_reads + CHUNK - 1) // CHUNK
    sizes = [min(CHUNK, n_reads - c * CHUNK) for c in range(n_chunks)]

    def run(c):
        return _chunk(c, sizes[c], seed, n_txps, kbar, cdf, g_start, g_size, gene_of, coverage, gaps, fam)

    if n_chunks > 1 and threads > 1:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            parts = list(ex.map(run, range(n_chunks)))
    else:
        parts = [run(c) for c in range(n_chunks)]
    lens = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, dtype=np.uint64)
    row_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=row_ptr[1:])
    tid = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, dtype=np.uint32)
    as_prob = np.concatenate([p[2] for p in parts]) if parts else np.zeros(0, dtype=np.float32)
    cov = np.concatenate([p[3] for p in parts]) if coverage and parts else None
    return SyntheticStore(row_ptr, tid, as_prob, cov, n_txps, a, gene_of)


# BASELINE.json configs (C2, C3/C4, C5)
CONFIGS = {
    "c2": dict(n_reads=1_000_000, n_txps=60_000, kbar=8.0),
    "c3": dict(n_reads=10_000_000, n_txps=200_000, kbar=8.0),
    "c5_cell": dict(n_reads=50_000, n_txps=60_000, kbar=8.0),
}


def make_config(name: str, coverage: bool = False, seed_offset: int = 0, **over) -> SyntheticStore:
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    return make_store(seed=BASE_SEED + seed_offset, coverage=coverage, **cfg)


def make_cells(n_cells: int, reads_per_cell: int, n_txps: int, kbar: float = 8.0,
               seed: int = BASE_SEED + 5, expressed_frac: Optional[float] = None, first_cell: int = 0, threads: int = 1):
    """C5: concatenated per-cell stores.  Cell c is a pure function of (seed, c), so ranks that each
    generate a block of cells (``first_cell``) produce pieces of one experiment.

    ``expressed_frac`` (None: every cell draws from all ``n_txps`` transcripts, the benchmark's workload): a cell
    expresses a random subset of that fraction of the annotation -- its store is generated over the subset and mapped
    back through the sorted subset ids, so genes stay runs of neighbouring ids (single-cell data: a cell touches a
    few per cent to a fifth of the transcripts; what the per-cell transcript compaction of the batched store is for)."""
    def one(c):
        cs = seed * 1000 + first_cell + c
        if expressed_frac is None:
            return make_store(reads_per_cell, n_txps, kbar, seed=cs, threads=1)
        n_sub = min(n_txps, max(int(n_txps * expressed_frac), 8))
        sub = np.sort(np.random.default_rng([cs, 0xCE11]).choice(n_txps, n_sub, replace=False)).astype(np.uint32)
        st = make_store(reads_per_cell, n_sub, kbar, seed=cs, threads=1)
        return SyntheticStore(st.row_ptr, sub[st.tid], st.as_prob, None, n_txps, None, None)

    if threads > 1 and n_cells > 1:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            stores = list(ex.map(one, range(n_cells)))
    else:
        stores = [one(c) for c in range(n_cells)]
    rps, tids, ps = [np.zeros(1, dtype=np.uint64)], [], []
    cell_off = np.zeros(n_cells + 1, dtype=np.uint64)
    base = 0
    for c, st in enumerate(stores):
        rps.append(st.row_ptr[1:] + np.uint64(base))
        tids.append(st.tid)
        ps.append(st.as_prob)
        base += st.nnz
        cell_off[c + 1] = cell_off[c] + np.uint64(st.n_reads)
    return (cell_off, np.concatenate(rps), np.concatenate(tids) if tids else np.zeros(0, np.uint32),
            np.concatenate(ps) if ps else np.zeros(0, np.float32))


def cell_shift(rep: int, n_txps: int) -> int:
    """Transcript-id rotation of replica ``rep`` of a cell in ``replicate_cells``."""
    return (rep * 7919) % n_txps


def replicate_cells(cells, n_txps: int, n_cells: int):
    """BASELINE configs[4] whole (5 k cells x 50 k reads, 2 G alignments) without generating 5 k cells in Python
    (~0.1 s of one core per cell): ``cells`` = (cell_off, row_ptr, tid, as_prob) of n_base generated cells, and cell
    ``r * n_base + c`` of the result is cell c with every transcript id rotated by ``cell_shift(r, n_txps)``
    (t -> (t + shift) mod T).  A relabelling of the transcripts is an exact symmetry of the EM, so the copies are
    distinct problems for the engine (other ids, other tiles, other windows) whose answers are known in terms of each
    other: counts[r * n_base + c][(t + shift) mod T] == counts[c][t] up to summation order."""
    cell_off_b, row_ptr_b, tid_b, p_b = cells
    n_base = len(cell_off_b) - 1
    reps = -(-n_cells // n_base)
    nnz_b, reads_b = len(tid_b), len(row_ptr_b) - 1
    tid = np.empty(nnz_b * reps, dtype=np.uint32)
    for r in range(reps):
        seg = tid[r * nnz_b:(r + 1) * nnz_b]
        np.add(tid_b, np.uint32(cell_shift(r, n_txps)), out=seg)
        np.subtract(seg, np.uint32(n_txps), out=seg, where=seg >= n_txps)
    row_ptr = np.empty(reads_b * reps + 1, dtype=np.uint64)
    cell_off = np.empty(n_base * reps + 1, dtype=np.uint64)
    for r in range(reps):
        row_ptr[r * reads_b:(r + 1) * reads_b] = row_ptr_b[:-1] + np.uint64(r * nnz_b)
        cell_off[r * n_base:(r + 1) * n_base] = np.asarray(cell_off_b[:-1], dtype=np.uint64) + np.uint64(r * reads_b)
    row_ptr[-1] = np.uint64(reps * nnz_b)
    cell_off[-1] = np.uint64(reps * reads_b)
    p = np.tile(p_b, reps)
    if n_cells < n_base * reps:   # trim to whole cells
        r1 = int(cell_off[n_cells])
        a1 = int(row_ptr[r1])
        return cell_off[:n_cells + 1].copy(), row_ptr[:r1 + 1].copy(), tid[:a1].copy(), p[:a1].copy()
    return cell_off, row_ptr, tid, p


def make_sirv_store(tag: str = "C", n_reads: int = 20_000, seed: int = BASE_SEED + 1,
                    coverage: bool = False, table_path: Optional[str] = None) -> SyntheticStore:
    """BASELINE config[0] stand-in: a SIRV-shaped store (T = 69 / 44 / 100 for the C / I / O
    annotations).  The reference's test_data has no BAM or reads, so reads are synthesised over
    the real SIRV isoform structure (tests/golden/sirv_txps.json: gene, length and pairwise exonic
    overlap of every transcript, derived from the reference's GTFs by scripts/make_sirv_fixture.py):
    a read's primary is drawn from a log-normal mix; every isoform of the same gene that shares
    exonic sequence with it is a secondary alignment with probability overlap/length, scored with
    a deficit that grows as the overlap shrinks."""
    import json
    import os
    if table_path is None:
        table_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                  "tests", "golden", "sirv_txps.json")
    tab = json.load(open(table_path))[tag]
    T = len(tab["names"])
    length = np.asarray(tab["length"], dtype=np.float64)
    ov = np.zeros((T, T))
    for k, v in tab["overlap"].items():
        i, j = (int(x) for x in k.split(","))
        ov[i, j] = ov[j, i] = v
    rng = np.random.default_rng([seed, ord(tag)])
    a = rng.lognormal(0.0, 1.5, size=T)
    a /= a.sum()
    t0 = rng.choice(T, size=n_reads, p=a)
    frac = np.clip(ov[t0] / length[t0][:, None], 0.0, 1.0)         # [n_reads, T]
    take = rng.random((n_reads, T)) < frac
    take[np.arange(n_reads), t0] = True
    d = np.minimum(np.floor((1.0 - frac) * 12.0) + rng.geometric(0.3, size=(n_reads, T)) - 1, 60)
    d[np.arange(n_reads), t0] = 0
    rows, cols = np.nonzero(take)
    p = np.exp((-d[rows, cols].astype(np.float32)) / np.float32(5.0)).astype(np.float32)
    lens = np.bincount(rows, minlength=n_reads).astype(np.uint64)
    row_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=row_ptr[1:])
    cov = None
    if coverage:
        cov = rng.uniform(0.05, 1.0, size=len(cols))
        ssum = np.add.reduceat(cov, row_ptr[:-1].astype(np.int64))
        cov = cov / np.repeat(ssum, lens.astype(np.int64))
    return SyntheticStore(row_ptr, cols.astype(np.uint32), p, cov, T, a, np.asarray(tab["gene"], dtype=np.int32))
