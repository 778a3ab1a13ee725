"""Independent NumPy f64 restatement of the oarfish EM (TEST INFRASTRUCTURE ONLY).

Written separately from oracle/oem_oracle.c (vectorised: ``np.add.reduceat`` for
the per-read denominators, ``np.bincount`` for the per-transcript sums) so that
the two restatements cross-check each other; neither shares code with the other.
Summation order differs from the serial reference, so agreement is to ~1e-12
relative, with exactly equal iteration counts away from knife-edge convergence.

Reference: src/em.rs:87-133 (m_step), :144-255 (do_em), :399 (em_par gate),
src/util/constants.rs:1-2, src/bootstrap.rs:7-16.
"""
from __future__ import annotations

import numpy as np

MIN_READ_THRESH = 1e-5   # constants.rs:1
EM_DENOM_THRESH = 1e-30  # constants.rs:2


def m_step(row_ptr, tid, as_prob, cov_prob, prev, n_txps, row_w=None):
    """One E/M pass (em.rs:87-133); ``row_w`` = integer multiplicity per read."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n_reads = len(row_ptr) - 1
    if n_reads == 0:
        return np.zeros(n_txps, dtype=np.float64)
    # em.rs:107-111: prev[t] * (p as f64) * cov * 1.0, multiplied left to right
    num = prev[tid] * as_prob.astype(np.float64)
    if cov_prob is not None:
        num = num * cov_prob
    lens = np.diff(row_ptr)
    denom = np.add.reduceat(num, row_ptr[:-1]) if len(num) else np.zeros(n_reads)
    # reduceat on an empty slice returns the next element; the store never has
    # empty rows (oarfish_types.rs:724,735-737) but guard anyway.
    denom = np.where(lens > 0, denom, 0.0)
    ok = denom > EM_DENOM_THRESH  # em.rs:115
    scale = np.zeros(n_reads, dtype=np.float64)
    scale[ok] = 1.0
    if row_w is not None:
        scale = scale * np.asarray(row_w, dtype=np.float64)
    safe = np.where(ok, denom, 1.0)
    row_of = np.repeat(np.arange(n_reads), lens)
    inc = num / safe[row_of] * scale[row_of]  # em.rs:128
    return np.bincount(tid, weights=inc, minlength=n_txps).astype(np.float64)  # em.rs:129


def do_em(row_ptr, tid, as_prob, cov_prob, n_txps, init=None, max_iter=1000, conv_thresh=1e-3,
          min_iter_gate=50, row_w=None):
    """em.rs:144-255.  Returns (counts, niter, n_passes, converged, rel_diff)."""
    tid = np.asarray(tid, dtype=np.int64)
    as_prob = np.asarray(as_prob, dtype=np.float32)
    n_reads = len(row_ptr) - 1
    if init is not None:
        prev = np.array(init, dtype=np.float64)              # em.rs:160-162
    else:
        prev = np.full(n_txps, n_reads / n_txps, dtype=np.float64)  # em.rs:165-166
    niter, n_passes, converged, last_rel = 0, 0, False, 0.0
    while niter < max_iter:                                   # em.rs:181
        curr = m_step(row_ptr, tid, as_prob, cov_prob, prev, n_txps, row_w)
        n_passes += 1
        m = prev > MIN_READ_THRESH                            # em.rs:195
        rel = 0.0                                             # em.rs:169/234
        if m.any():
            rel = max(0.0, float(np.max((curr[m] - prev[m]) / prev[m])))  # em.rs:196-199
        prev = curr                                           # em.rs:204-207
        last_rel = rel
        if rel < conv_thresh and niter > min_iter_gate:       # em.rs:212 / :399
            converged = True
            break
        niter += 1                                            # em.rs:218
    prev = np.where(prev < MIN_READ_THRESH, 0.0, prev)        # em.rs:238-242
    out = m_step(row_ptr, tid, as_prob, cov_prob, prev, n_txps, row_w)  # em.rs:245-252
    n_passes += 1
    return out, niter, n_passes, converged, last_rel


def sample_weights(n_reads: int, rng: np.random.Generator):
    """bootstrap.rs:7-16 in multiplicity form: counts of n draws from Uniform[0,n)."""
    inds = rng.integers(0, n_reads, size=n_reads)
    return np.bincount(inds, minlength=n_reads).astype(np.uint32)
