#!/usr/bin/env python3
"""Cost attribution of the per-cell loop: 100 all-cells-live passes (max_iter 100, threshold 0: every cell runs every
pass whatever the switches do to the numbers) with parts of the tile kernel switched off (OEM_TILE_EXP, test-only
library).  usage: cells_exp.py [n_cells]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
import oarfish_amd
from oarfish_amd import synth, _lib
from oarfish_amd.em import cells_last_timing
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 625
cell_off, row_ptr, tid, p = synth.make_cells(n_cells, 50_000, 60_000, seed=3, threads=min(32, os.cpu_count() or 4))
_lib.testing().__enter__()
_lib.lib()
for mask in [int(m) for m in os.environ.get('OEM_EXP_MASKS', '0,1,16,4,8,2,19,31').split(',')]:
    os.environ["OEM_TILE_EXP"] = str(mask)
    out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, 60_000, max_iter=100, convergence_thresh=0.0)
    ms, passes = cells_last_timing()
    print(f"OEM_TILE_EXP={mask:2d}: {ms / max(passes, 1) * 1e3:8.1f} us per all-live pass ({passes} passes)")
