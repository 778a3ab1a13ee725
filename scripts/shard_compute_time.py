#!/usr/bin/env python3
"""Per-GPU compute time of one EM iteration for the row shard a rank would own at N = 1, 2, 4, 8
(no collective: one GPU, the shard sizes only).  The driver's SCALE run adds the all-reduce."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
if os.environ.get("OEM_USE_TESTING_LIB") == "1":   # (knobs: OEM_TILE_ROWS ...)
    _lib.testing().__enter__()
only = [int(x) for x in os.environ.get("SHARD_NS", "1,2,4,8").split(",")]
full = synth.make_store(10_000_000, 200_000, 8.0, threads=32)
for n in only:
    r1 = full.n_reads // n
    a1 = int(full.row_ptr[r1])
    d = DeviceStore(full.row_ptr[:r1 + 1], full.tid[:a1], full.as_prob[:a1], None, full.n_txps)
    d.time_m_step(300)
    ms = min(d.time_em_iters(300) for _ in range(3)) / 300     # (an un-attached store: the single-device loop, stopping rule one pass behind)
    pm = min(d.time_m_step(200) for _ in range(3))             # (what a row shard's iteration has before its exchange kernels: tile kernel + fold)
    print(f"N={n}: shard {r1} reads, {a1} alignments: pass {pm*1e3:.1f} us, single-device iteration {ms*1e3:.1f} us -> {1e3/ms:.0f} it/s upper bound")
    d.close()
