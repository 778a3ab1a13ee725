"""Shared helpers of the test-suite (tests are the only place oracle/ is used)."""
import glob
import os

import numpy as np

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
