#!/usr/bin/env python3
"""Throughput of the coverage model: host functions (one core) vs oem_coverage_probs_device.
usage: coverage_bench.py [n_reads] [n_txps]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
rng = np.random.default_rng(1)
txp_len = rng.integers(400, 6000, size=T).astype(np.uint64)
lens = rng.integers(1, 16, size=R)
rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
a = rng.lognormal(0, 2, size=T); a /= a.sum()
t0 = rng.choice(T, size=R, p=a)
tid = ((np.repeat(t0, lens) + (np.arange(len(lens).__class__(rp[-1])) if False else np.arange(int(rp[-1])) - np.repeat(rp[:-1].astype(np.int64), lens))) % T).astype(np.uint32)
L = txp_len[tid].astype(np.int64)
start = (rng.random(len(tid)) * np.maximum(L - 350, 1)).astype(np.int64)
end = np.minimum(start + rng.integers(100, 3000, size=len(tid)), L).astype(np.uint32)
start = start.astype(np.uint32)
nnz = len(tid)
out = np.zeros(nnz)
L_ = _lib.lib()
for model, name in ((0, "logistic"), (1, "binomial")):
    for rep in range(2):
        t = time.perf_counter()
        _lib.check(L_.oem_coverage_probs_device(rp.ctypes.data, tid.ctypes.data, start.ctypes.data, end.ctypes.data,
                                                txp_len.ctypes.data, R, nnz, T, 100, model, 2.0, 0, out.ctypes.data))
        dt = time.perf_counter() - t
    print(f"device {name}: {nnz} alignments in {dt:.3f} s (incl. upload + read-back) = {nnz / dt * 1e-6:.0f} M alignments/s; "
          f"row sums ok: {np.allclose(np.add.reduceat(out, rp[:-1].astype(np.int64)), 1.0)}")
