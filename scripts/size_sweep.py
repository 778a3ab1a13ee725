#!/usr/bin/env python3
"""EM pass time / roofline fraction / store creation time against store size on one GPU
(k-bar = 8, seeded generator of oarfish_amd.synth).  usage: size_sweep.py "R:T R:T ..." """
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore
_lib.lib()
print("| reads | transcripts | alignments | create s | pass ms | iteration ms | algorithmic GB/s | frac of 8 TB/s | HBM MB |")
print("|---|---|---|---|---|---|---|---|---|")
for spec in sys.argv[1].split():
    R, T = (int(x) for x in spec.split(":"))
    st = synth.make_store(R, T, 8.0, threads=min(32, os.cpu_count() or 8))
    t = time.perf_counter()
    d = DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T)
    tc = time.perf_counter() - t
    hbm, alg = d.bytes()
    d.time_em_iters(5)
    k_ms = d.time_m_step(50)
    it_ms = d.time_em_iters(100) / 100
    print(f"| {R} | {T} | {st.nnz} | {tc:.2f} | {k_ms:.4f} | {it_ms:.4f} | {alg / k_ms / 1e6:.0f} | {alg / k_ms / 1e6 / 8000:.3f} | {hbm / 1e6:.0f} |", flush=True)
    d.close()
    del st
