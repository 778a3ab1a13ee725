#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference holds no golden vectors for src/em.rs / src/bootstrap.rs and
cannot be built or imported here (Rust), so the vectors
are *constructed*:

  * closed-form cases whose answer is known analytically;
  * seeded random stores whose expected outputs come from the independent NumPy
    restatement (oracle/oracle_np.py); the C restatement (oracle/oem_oracle.c)
    must agree with them (checked here at generation time and again in
    tests/test_oracle.py).

Each fixture is one .npz of inputs + expected outputs (data only).
Run:  python scripts/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import c_oracle, oracle_np  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def csr_from_rows(rows):
    """rows: list of (tids, probs)"""
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
    tid, p = [], []
    for i, (t, w) in enumerate(rows):
        tid.extend(t)
        p.extend(w)
        row_ptr[i + 1] = len(tid)
    return row_ptr, np.asarray(tid, dtype=np.uint32), np.asarray(p, dtype=np.float32)


def random_store(rng, n_reads, n_txps, kmax=6, coverage=False, zero_frac=0.0):
    k = rng.integers(1, kmax + 1, size=n_reads)
    a = rng.lognormal(0, 1.5, size=n_txps)
    a /= a.sum()
    rows, covs = [], []
    for i in range(n_reads):
        kk = min(int(k[i]), n_txps)
        t0 = rng.choice(n_txps, p=a)
        near = (t0 + rng.integers(-3, 4, size=4 * kk)) % n_txps
        far = rng.integers(0, n_txps, size=kk)
        cand = [t0] + list(near) + list(far)
        seen, ts = set(), []
        for c in cand:
            if int(c) not in seen:
                seen.add(int(c))
                ts.append(int(c))
            if len(ts) == kk:
                break
        d = np.concatenate([[0], rng.geometric(0.15, size=len(ts) - 1) - 1]).astype(np.float32)
        rows.append((ts, np.exp(-d / np.float32(5.0)).astype(np.float32)))
    row_ptr, tid, p = csr_from_rows(rows)
    cov = None
    if coverage:
        cov = rng.uniform(0.05, 1.0, size=len(tid))
        if zero_frac > 0:
            cov[rng.random(len(tid)) < zero_frac] = 0.0
        for i in range(n_reads):
            s, e = int(row_ptr[i]), int(row_ptr[i + 1])
            tot = cov[s:e].sum()
            cov[s:e] /= tot if tot > 0 else 1.0  # normalize_probability.rs:61-69
    return row_ptr, tid, p, cov


def expected(row_ptr, tid, p, cov, T, runs, row_w=None, init=None):
    """runs: list of (max_iter, conv_thresh, gate).  Expected from NumPy; C must agree."""
    s = c_oracle.Store(row_ptr, tid, p, cov, T)
    out = {}
    for n, (mi, ct, gate) in enumerate(runs):
        cnt, niter, npass, conv, rel = oracle_np.do_em(row_ptr, tid, p, cov, T, init=init, max_iter=mi,
                                                       conv_thresh=ct, min_iter_gate=gate, row_w=row_w)
        c_cnt, info = c_oracle.do_em(s, init=init, max_iter=mi, conv_thresh=ct, min_iter_gate=gate,
                                     row_w=row_w)
        assert info.niter == niter and info.n_passes == npass and info.converged == conv, (info, niter, npass, conv)
        tol = 1e-9 * max(1.0, np.abs(cnt).max())
        assert np.abs(c_cnt - cnt).max() <= tol, np.abs(c_cnt - cnt).max()
        out[f"run{n}_params"] = np.array([mi, ct, gate], dtype=np.float64)
        out[f"run{n}_counts"] = cnt
        out[f"run{n}_info"] = np.array([niter, npass, int(conv), rel], dtype=np.float64)
    return out


def save(name, row_ptr, tid, p, cov, T, extra):
    d = dict(row_ptr=row_ptr, tid=tid, as_prob=p, n_txps=np.array([T], dtype=np.int64))
    if cov is not None:
        d["cov_prob"] = cov
    d.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, "R=%d nnz=%d T=%d" % (len(row_ptr) - 1, len(tid), T))


def main():
    os.makedirs(OUT, exist_ok=True)
    std_runs = [(0, 1e-3, 50), (1, 0.0, 50), (10, 0.0, 50), (100, 0.0, 50), (1000, 1e-3, 50),
                (1000, 1e-3, 1), (1000, 1e-2, 1), (1000, 1e-8, 50)]

    # 1. closed form: 2 transcripts; n_A unique to A, n_B unique to B, m ambiguous with
    #    equal weights => theta_A = n_A + m*n_A/(n_A+n_B)  
    nA, nB, m = 30, 10, 20
    rows = [([0], [1.0])] * nA + [([1], [1.0])] * nB + [([0, 1], [0.5, 0.5])] * m
    row_ptr, tid, p = csr_from_rows(rows)
    ex = expected(row_ptr, tid, p, None, 2, [(1000, 1e-12, 50)])
    ex["closed_form"] = np.array([nA + m * nA / (nA + nB), nB + m * nB / (nA + nB)])
    assert np.allclose(ex["run0_counts"], ex["closed_form"], rtol=1e-9)
    save("closed_form_two_txps", row_ptr, tid, p, None, 2, ex)

    # 2. unique-only store: counts are the integer histogram after the first pass
    rng = np.random.default_rng(7)
    T = 17
    t = rng.integers(0, T, size=500)
    rows = [([int(x)], [float(np.float32(rng.uniform(0.1, 1.0)))]) for x in t]
    row_ptr, tid, p = csr_from_rows(rows)
    ex = expected(row_ptr, tid, p, None, T, [(1, 0.0, 50), (100, 1e-3, 50)])
    ex["closed_form"] = np.bincount(t, minlength=T).astype(np.float64)
    assert np.allclose(ex["run0_counts"], ex["closed_form"]) and np.allclose(ex["run1_counts"], ex["closed_form"])
    save("unique_only", row_ptr, tid, p, None, T, ex)

    # 3. single read, single alignment; and a read whose weight underflows the denominator
    row_ptr, tid, p = csr_from_rows([([3], [0.25])])
    ex = expected(row_ptr, tid, p, None, 5, [(100, 1e-3, 50)])
    ex["closed_form"] = np.array([0, 0, 0, 1.0, 0])
    save("single_read", row_ptr, tid, p, None, 5, ex)

    # 4. seeded random stores, without and with the coverage column (some zero cov probs)
    for name, seed, R, T, cov, zf in [("random_a", 11, 1500, 200, False, 0.0),
                                      ("random_b", 12, 3000, 64, False, 0.0),
                                      ("random_cov", 13, 2000, 150, True, 0.0),
                                      ("random_cov_zeros", 14, 1200, 90, True, 0.15)]:
        rng = np.random.default_rng(seed)
        row_ptr, tid, p, c = random_store(rng, R, T, coverage=cov, zero_frac=zf)
        ex = expected(row_ptr, tid, p, c, T, std_runs)
        save(name, row_ptr, tid, p, c, T, ex)

    # 5. initial abundances supplied (-q short-read seeding, bulk.rs:125-127), incl. zeros:
    #    a transcript seeded at 0 stays at 0 (theta_t = 0 is absorbing)
    rng = np.random.default_rng(21)
    row_ptr, tid, p, c = random_store(rng, 1000, 80)
    init = rng.uniform(0, 30, size=80)
    init[rng.random(80) < 0.2] = 0.0
    ex = expected(row_ptr, tid, p, None, 80, [(50, 0.0, 50), (1000, 1e-3, 50)], init=init)
    ex["init"] = init
    save("random_init", row_ptr, tid, p, None, 80, ex)

    # 6. the stopping gates: a store that converges (rel < 1e-2) between pass 3 and 52
    #    => em_par's gate (niter>1) and em's gate (niter>50) stop at different iterations
    rng = np.random.default_rng(31)
    row_ptr, tid, p, c = random_store(rng, 800, 12, kmax=3)
    ex = expected(row_ptr, tid, p, None, 12, [(1000, 5e-2, 1), (1000, 5e-2, 50), (1000, 5e-2, 10)])
    assert ex["run0_info"][0] < 51 <= ex["run1_info"][0], (ex["run0_info"], ex["run1_info"])
    save("gate_cases", row_ptr, tid, p, None, 12, ex)

    # 7. signed rel-diff: one transcript only ever loses mass, so its (negative) diffs must
    #    not count; convergence is decided by the floor at 0 (em.rs:169,199)
    rows = [([0, 1], [1.0, 0.2])] * 40 + [([0], [1.0])] * 40
    row_ptr, tid, p = csr_from_rows(rows)
    init = np.array([1.0, 79.0])
    ex = expected(row_ptr, tid, p, None, 2, [(3, 1e-3, 1), (1000, 1e-3, 1), (1000, 1e-3, 50)], init=init)
    ex["init"] = init
    save("signed_reldiff", row_ptr, tid, p, None, 2, ex)

    # 8. bootstrap: explicit index vectors (duplicates, missing rows) in multiplicity form
    rng = np.random.default_rng(41)
    row_ptr, tid, p, c = random_store(rng, 600, 40)
    B = 4
    W = np.zeros((B, 600), dtype=np.uint32)
    for b in range(B):
        W[b] = oracle_np.sample_weights(600, rng)
    W[3, :] = 0
    W[3, :10] = 60  # extreme resample: ten reads, sixty copies each
    s = c_oracle.Store(row_ptr, tid, p, None, 40)
    outs = []
    for b in range(B):
        inds = np.repeat(np.arange(600, dtype=np.uint64), W[b])
        via_inds, i1 = c_oracle.do_em(s, inds=inds, max_iter=1000, conv_thresh=1e-3)
        via_w, i2 = c_oracle.do_em(s, row_w=W[b], max_iter=1000, conv_thresh=1e-3)
        npy = oracle_np.do_em(row_ptr, tid, p, None, 40, max_iter=1000, conv_thresh=1e-3, row_w=W[b])
        assert i1.niter == i2.niter == npy[1]
        assert np.abs(via_inds - via_w).max() < 1e-9 and np.abs(via_w - npy[0]).max() < 1e-9
        outs.append(npy[0])
    save("bootstrap_inject", row_ptr, tid, p, None, 40,
         dict(row_w=W, boot_counts=np.stack(outs), boot_params=np.array([1000, 1e-3])))

    # 9. edge shapes: a ragged store with one 100-alignment read (--best-n cap), zero-weight
    #    alignments and a transcript nobody maps to
    rng = np.random.default_rng(51)
    rows = [(list(range(100)), list(np.exp(-rng.geometric(0.15, size=100).astype(np.float32) / 5)))]
    rows += [([int(x)], [1.0]) for x in rng.integers(0, 120, size=50)]
    rows += [([5, 7, 9], [0.0, 1.0, 0.0])] * 5
    row_ptr, tid, p = csr_from_rows(rows)
    ex = expected(row_ptr, tid, p, None, 130, [(100, 0.0, 50), (1000, 1e-3, 50)])
    save("ragged_edge", row_ptr, tid, p, None, 130, ex)


if __name__ == "__main__":
    main()
