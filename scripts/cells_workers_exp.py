#!/usr/bin/env python3
"""Per-cell call over several groups (2 500 cells = 625 generated x 4 transcript-id rotations): host workers that draw
groups (test-only library: OEM_CELLS_WORKERS).  usage: cells_workers_exp.py [n_cells]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oarfish_amd
from oarfish_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
T = 60_000
base = synth.make_cells(625, 50_000, T, seed=3, threads=min(32, os.cpu_count() or 4))
co, rp, tid, p = synth.replicate_cells(base, T, n)
with _lib.testing():
    c2 = int(co[2]); a2 = int(rp[c2])
    oarfish_amd.em_cells(co[:3], rp[:c2 + 1], tid[:a2], p[:a2], None, T, max_iter=5)
    for w in (2, 3, 4, 2, 3, 4):
        os.environ["OEM_CELLS_WORKERS"] = str(w)
        t = time.perf_counter()
        out, infos = oarfish_amd.em_cells(co, rp, tid, p, None, T, max_iter=1000, convergence_thresh=1e-3)
        dt = time.perf_counter() - t
        print(f"workers {w}: {n} cells in {dt:.3f} s = {n / dt:.1f} cells/s, sum {out.sum():.1f}")
        del out
