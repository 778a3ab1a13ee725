#!/usr/bin/env python3
"""Where a tile's lifetime goes: per-phase wall-clock account of k_em_tile from in-kernel timestamps
(test-only library, oem_debug_tile_probe: wave 0 of every workgroup stamps the 100 MHz device clock at its
phase boundaries).  usage: tile_probe.py [c3|c2]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
PHASES = ["descriptor load + slice addresses", "issue loads, theta window -> LDS, clear windows", "barrier 1",
          "remote gathers land + denominator atomics (phase A)", "barrier 2",
          "slice 0 (operands land + fold)", "slice 1", "slice 2", "slice 3", "barrier 3", "queue stores + window flush"]
with _lib.testing():
    L = _lib.lib()
    L.oem_debug_tile_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    st = synth.make_config(wl)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        pm = d.time_m_step(20)
        # number of tiles: probe into a generous buffer, trim the rows that stayed zero
        cap = st.n_reads // 64 + 1024
        buf = np.zeros((cap, 16), dtype=np.uint64)
        _lib.check(L.oem_debug_tile_probe(d.handle, buf.ctypes.data, buf.size)) if False else None
        # the hook checks n_out >= n_tiles * 16; cap is an upper bound of n_tiles
        rc = L.oem_debug_tile_probe(d.handle, buf.ctypes.data, buf.size)
        _lib.check(rc)
t = buf[buf[:, 0] != 0].astype(np.int64)
n = len(t)
us = 0.01  # 100 MHz
life = (t[:, 11] - t[:, 0]) * us
span = (t[:, 11].max() - t[:, 0].min()) * us
print(f"{wl}: {n} tiles; pass (HIP events, unprobed) {pm * 1e3:.1f} us; probed tile kernel spans {span:.1f} us; "
      f"tile lifetime mean {life.mean():.2f} us (p10 {np.percentile(life, 10):.2f}, p50 {np.percentile(life, 50):.2f}, "
      f"p90 {np.percentile(life, 90):.2f}); mean resident workgroups {life.sum() / span:.0f} of 1280 slots")
print(f"{'phase (wave 0 of the workgroup)':62s} {'mean us':>8s} {'p10':>7s} {'p50':>7s} {'p90':>7s} {'share':>6s}")
for i, name in enumerate(PHASES):
    dt = (t[:, i + 1] - t[:, i]) * us
    print(f"{name:62s} {dt.mean():8.2f} {np.percentile(dt, 10):7.2f} {np.percentile(dt, 50):7.2f} {np.percentile(dt, 90):7.2f} "
          f"{dt.mean() / life.mean():6.1%}")
start = (t[:, 0] - t[:, 0].min()) * us
order = np.argsort(start)
print("dispatch: first start of tiles #0/#1279/#1280/#2560/#5120/last:",
      [round(float(start[order[min(k, n - 1)]]), 1) for k in (0, 1279, 1280, 2560, 5120, n - 1)])
# occupancy over the kernel's life: resident workgroups in ten slices of its span, and what the ramp and the tail cost
end = (t[:, 11] - t[:, 0].min()) * us
edges = np.linspace(0.0, span, 11)
occ = [float(np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])]
print("resident workgroups by tenth of the span:", [round(o) for o in occ])
last_start = float(start.max())
print(f"last tile starts at {last_start:.1f} us of {span:.1f}: the tail after it is {span - last_start:.1f} us "
      f"({(span - last_start) / span:.1%} of the span) at a mean of "
      f"{float(np.clip(end - np.maximum(start, last_start), 0, None).sum() / max(span - last_start, 1e-9)):.0f} resident workgroups")
