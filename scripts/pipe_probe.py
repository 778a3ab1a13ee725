#!/usr/bin/env python3
"""Where a tile's time goes in the PIPELINED tile walk (oem_tile_pipe.hip): per-phase wall-clock account from
in-kernel timestamps (test-only library, oem_debug_tile_probe: wave 0 of every workgroup stamps the 100 MHz device
clock at the phase boundaries of every tile it walks).  usage: pipe_probe.py [c3] [weight_coding]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
coding = int(sys.argv[2]) if len(sys.argv) > 2 else 0
PHASES = ["wait at barrier 1", "clear denominators of k+1", "-",
          "slice 0 fold + request slice 0 of k+1", "slice 1", "slice 2", "slice 3", "descriptor k+2 + wait at barrier 2",
          "G: gathers of k+1 issued", "B: queue stores", "window flush", "W(k+1), gathers land, A(k+1), rotate"]
os.environ.setdefault("OEM_TILE_PIPE", "1")
with _lib.testing():
    L = _lib.lib()
    L.oem_debug_tile_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    st = synth.make_config(wl)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps, weight_coding=coding) as d:
        pm = d.time_m_step(20)
        cap = st.n_reads // 64 + 1024
        buf = np.zeros((cap, 16), dtype=np.uint64)
        _lib.check(L.oem_debug_tile_probe(d.handle, buf.ctypes.data, buf.size))
t = buf[buf[:, 0] != 0].astype(np.int64)
n = len(t)
us = 0.01  # 100 MHz
life = (t[:, 12] - t[:, 0]) * us
span = (t[:, 12].max() - t[:, 0].min()) * us
print(f"{wl}: {n} tiles; pass (HIP events, unprobed) {pm * 1e3:.1f} us; probed tile kernel spans {span:.1f} us; "
      f"tile iteration mean {life.mean():.2f} us (p10 {np.percentile(life, 10):.2f}, p50 {np.percentile(life, 50):.2f}, "
      f"p90 {np.percentile(life, 90):.2f}); mean busy workgroups {life.sum() / span:.0f}")
print(f"{'phase (wave 0 of the workgroup)':62s} {'mean us':>8s} {'p10':>7s} {'p50':>7s} {'p90':>7s} {'share':>6s}")
for i, name in enumerate(PHASES):
    dt = (t[:, i + 1] - t[:, i]) * us
    print(f"{name:62s} {dt.mean():8.2f} {np.percentile(dt, 10):7.2f} {np.percentile(dt, 50):7.2f} {np.percentile(dt, 90):7.2f} "
          f"{dt.mean() / life.mean():6.1%}")
# static stride: tile i belongs to workgroup i % G, G = ceil(n / ceil(n / slots))
slots = int(os.environ.get("OEM_PIPE_SLOTS", 1024))
rounds = -(-n // slots)
G = -(-n // rounds)
full = buf[:n].astype(np.int64)
tot = np.zeros(G)
np.add.at(tot, np.arange(n) % G, (full[:, 12] - full[:, 0]) * us)
print(f"grid {G} x {rounds} tiles: busy time per workgroup mean {tot.mean():.1f} us, p10 {np.percentile(tot, 10):.1f}, "
      f"p90 {np.percentile(tot, 90):.1f}, max {tot.max():.1f} -> {tot.mean() / tot.max():.1%} of the slowest")
end = (t[:, 12] - t[:, 0].min()) * us
print("last tile ends: p50 / p90 / max of the per-tile end times:", [round(float(np.percentile(end, q)), 1) for q in (50, 90, 100)])
