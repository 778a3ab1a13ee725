#!/usr/bin/env python3
"""Pass / iteration time of the EM on a store WITH the coverage column (f64 weights, em.rs:107-111)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore
_lib.lib()
R, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 200_000)
for cov in (False, True):
    st = synth.make_store(R, T, 8.0, coverage=cov, threads=32)
    d = DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob, T)
    hbm, alg = d.bytes()
    d.time_em_iters(5)
    k = d.time_m_step(50); it = d.time_em_iters(100) / 100
    d.bootstrap(2, seed=9, max_iter=2)   # untimed: allocates the batch buffers
    nb = 16
    t = time.perf_counter(); b, bi = d.bootstrap(nb, seed=1); tb = time.perf_counter() - t
    print(f"coverage={cov}: pass {k:.4f} ms, iteration {it:.4f} ms, algorithmic {alg/1e6:.0f} MB -> {alg/k/1e6:.0f} GB/s ({alg/k/1e6/8000:.3f} of 8 TB/s); bootstraps {nb/tb:.2f}/s ({nb} replicates)")
    d.close()
