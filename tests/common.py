"""Shared helpers of the test-suite (tests are the only place oracle/ is used)."""
import glob
import os

import numpy as np

from oarfish_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["n_txps"] = int(d["n_txps"][0])
    d.setdefault("cov_prob", None)
    d.setdefault("init", None)
    runs = []
    n = 0
    while f"run{n}_params" in d:
        mi, ct, gate = d[f"run{n}_params"]
        niter, npass, conv, rel = d[f"run{n}_info"]
        runs.append(dict(max_iter=int(mi), conv_thresh=float(ct), gate=int(gate), counts=d[f"run{n}_counts"],
                         niter=int(niter), n_passes=int(npass), converged=bool(conv), rel_diff=float(rel)))
        n += 1
    d["runs"] = runs
    return d


def assert_counts_close(got, want, n_reads, n_txps, rtol=1e-4, what=""):
    """BASELINE.json north_star: abundances match to <= 1e-4 relative.
    |a-b| <= rtol * max(|b|, 1e-5 * R/T) avoids 0/0 on absent transcripts."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    floor = 1e-5 * max(n_reads, 1) / max(n_txps, 1)
    tol = rtol * np.maximum(np.abs(want), floor)
    bad = np.abs(got - want) > tol
    assert not bad.any(), (
        f"{what}: {bad.sum()} of {len(want)} transcripts differ; worst rel "
        f"{np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)):.3e}")


def weights_with(n_distinct, size, rng):
    vals = ((1.0 + np.arange(n_distinct)) / (n_distinct + 3.0)).astype(np.float32)
    p = vals[rng.integers(0, n_distinct, size=size)]
    p[:n_distinct] = vals
    return p


def tile_test_store(kind, seed):
    """(row_ptr, tid, as_prob, n_txps) of the shapes the tile kernels are tested on."""
    rng = np.random.default_rng(seed)
    if kind == "dense":        # many reads per transcript: hot anchors, 8 window copies
        st = synth.make_store(150_000, 2_000, seed=seed)
    elif kind == "sparse":     # wide windows, few reads per tile window entry
        st = synth.make_store(60_000, 40_000, seed=seed)
    elif kind == "long":       # reads with 20+ local alignments: the reload loops behind the register sets
        st = synth.make_store(30_000, 3_000, kbar=24.0, seed=seed)
    elif kind == "remote":     # mostly-remote: more records than a thread's six register slots (overflow path)
        n, T = 40_000, 30_000
        k = rng.integers(6, 14, n)
        row_ptr = np.concatenate([[0], np.cumsum(k)]).astype(np.uint64)
        tid = np.empty(int(row_ptr[-1]), np.uint32)
        for i in range(n):     # distinct transcripts per read, all over the annotation
            tid[int(row_ptr[i]):int(row_ptr[i + 1])] = rng.choice(T, int(k[i]), replace=False)
        p = np.exp(-rng.integers(0, 30, len(tid)) / 5.0).astype(np.float32)
        return row_ptr, tid, p, T
    elif kind == "tiny":       # one tile, three slices
        st = synth.make_store(150, 40, seed=seed)
    elif kind == "ragged":     # a last tile with a few reads, single-alignment reads, one very long read
        st = synth.make_store(20_000 + 7, 900, seed=seed)
    else:
        raise ValueError(kind)
    return st.row_ptr, st.tid, st.as_prob, st.n_txps
