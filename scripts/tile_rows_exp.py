#!/usr/bin/env python3
"""Reads per tile against store size (test-only library: OEM_TILE_ROWS overrides oem_layout.h's tile_rows_for).
Prefixes of the C3 store (the row shards of N = 1..8 ranks and smaller) and the C2 store, each laid out with
1024 / 512 / 256 / 128 reads per tile: HIP-event time of one E/M pass and of one loop iteration.
usage: tile_rows_exp.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore

rows_list = [int(a) for a in sys.argv[1:]] or [1024, 512, 256, 128]
full = synth.make_store(10_000_000, 200_000, 8.0, threads=32)
c2 = synth.make_config("c2")
cases = [("c3/1", full, full.n_reads), ("c3/2", full, full.n_reads // 2), ("c3/4", full, full.n_reads // 4),
         ("c3/8", full, full.n_reads // 8), ("c2", c2, c2.n_reads), ("c2/2", c2, c2.n_reads // 2), ("c2/4", c2, c2.n_reads // 4),
         ("c2/8", c2, c2.n_reads // 8), ("c2/32", c2, c2.n_reads // 32)]
if os.environ.get("OEM_EXP_SMALL"):
    cases = [c for c in cases if c[2] <= 1_000_000]
with _lib.testing():
    for name, st, r1 in cases:
        a1 = int(st.row_ptr[r1])
        out = []
        for rows in rows_list:
            os.environ["OEM_TILE_ROWS"] = str(rows)
            with DeviceStore(st.row_ptr[:r1 + 1], st.tid[:a1], st.as_prob[:a1], None, st.n_txps) as d:
                d.time_m_step(20)
                pm = min(d.time_m_step(50) for _ in range(3))
                it = min(d.time_em_iters(200) for _ in range(3)) / 200
                out.append(f"{rows}: {d.info(_lib.OEM_INFO_TILES)} tiles, pass {pm * 1e3:.1f} us, iteration {it * 1e3:.1f} us")
        print(f"{name} ({r1} reads): " + " | ".join(out), flush=True)
