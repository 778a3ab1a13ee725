#!/usr/bin/env python3
"""Times the batched per-cell EM against the one-cell-at-a-time path (OEM_SERIAL_CELLS=1).
usage: cells_bench.py <cells> <reads per cell> <transcripts> [expressed fraction of the annotation per cell]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
import oarfish_amd
from oarfish_amd import synth
n_cells, rpc, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frac = float(sys.argv[4]) if len(sys.argv) > 4 else None
cell_off, row_ptr, tid, p = synth.make_cells(n_cells, rpc, T, seed=3, threads=min(32, os.cpu_count() or 4), expressed_frac=frac)
from oarfish_amd import _lib
if os.environ.get("OEM_SERIAL_CELLS") or os.environ.get("OEM_USE_TESTING_LIB") == "1":   # knobs live in the test-only library
    _lib.testing().__enter__()
_lib.lib()  # load the library (and the process's HIP runtime) outside the timed region
c2 = int(cell_off[2]); a2 = int(row_ptr[c2])
oarfish_amd.em_cells(cell_off[:3], row_ptr[:c2 + 1], tid[:a2], p[:a2], None, T, max_iter=5)   # HIP runtime start-up (~0.2 s) outside too
t = time.perf_counter()
out, infos = oarfish_amd.em_cells(cell_off, row_ptr, tid, p, None, T, max_iter=1000, convergence_thresh=1e-3)
dt = time.perf_counter() - t
print(f"{'serial' if os.environ.get('OEM_SERIAL_CELLS') else 'batched'}: {n_cells} cells x {rpc} reads, T={T}{'' if frac is None else f', {frac:g} of it expressed per cell'}: {dt:.3f} s "
      f"({n_cells/dt:.1f} cells/s), mean passes {np.mean([i.n_passes for i in infos]):.0f}, sum {out.sum():.1f}")
