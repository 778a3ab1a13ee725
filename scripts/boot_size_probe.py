#!/usr/bin/env python3
"""Batched bootstrap pass against store size at a fixed transcript count: would a pass cut into halves (each half's
queue small enough for the Infinity Cache) cost less per read?  usage: boot_size_probe.py "R R ..." [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import synth
from oarfish_amd.types import DeviceStore
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
for R in (int(x) for x in sys.argv[1].split()):
    st = synth.make_store(R, T, 8.0, threads=min(32, os.cpu_count() or 8))
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T) as d:
        d.time_bootstrap_passes(20)
        ms, slots, nbytes = min(d.time_bootstrap_passes(40) for _ in range(3))
        d.time_m_step(100)
        pm = min(d.time_m_step(100) for _ in range(3))
        print(f"{R} reads x {T}: batched pass {ms:.4f} ms = {ms / R * 1e9:.1f} ps per read ({slots} slots); "
              f"point-estimate pass {pm:.4f} ms = {pm / R * 1e9:.1f} ps per read", flush=True)
    del st
