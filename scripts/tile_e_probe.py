#!/usr/bin/env python3
"""Per-phase wall-clock account of k_em_tile_e (the batched bootstrap's tile kernel) from in-kernel timestamps
(test-only library).  usage: tile_e_probe.py [c3]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
with _lib.testing():
    L = _lib.lib()
    L.oem_debug_tile_e_probe_begin.argtypes = [C.c_uint64]
    L.oem_debug_tile_e_probe_end.argtypes = [C.c_void_p, C.c_uint64]
    st = synth.make_config(wl)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        ms, slots, _ = d.time_bootstrap_passes(10)
        cap = st.n_reads // 64 + 1024
        _lib.check(L.oem_debug_tile_e_probe_begin(cap))
        d.time_bootstrap_passes(5)          # the last launch's stamps stay
        buf = np.zeros((cap, 16), dtype=np.uint64)
        _lib.check(L.oem_debug_tile_e_probe_end(buf.ctypes.data, buf.size))
t = buf[buf[:, 0] != 0].astype(np.int64)
us = 0.01
life = (t[:, 9] - t[:, 0]) * us
span = (t[:, 9].max() - t[:, 0].min()) * us
print(f"{wl}: {len(t)} tiles; batched pass (unprobed) {ms * 1e3:.1f} us for {slots} slots; probed k_em_tile_e spans {span:.1f} us; "
      f"tile lifetime mean {life.mean():.2f} us (p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}); "
      f"mean resident workgroups {life.sum() / span:.0f} of 512 slots")
seq = [(0, 1, "descriptor + slice addresses"), (1, 2, "issue all loads, remote x, theta window -> LDS, clear"), (2, 3, "barrier 1"),
       (3, 4, "remote denominators (phase A)"), (4, 5, "barrier 2"), (5, 6, "first slice (the widest): both passes, 4 slots"),
       (6, 7, "second slice (the narrowest)"), (7, 8, "barrier 3"),
       (8, 9, "queue stores + window flush")]
print(f"{'phase (wave 0 of the workgroup)':56s} {'mean us':>8s} {'p10':>7s} {'p50':>7s} {'p90':>7s} {'share':>6s}")
for a, b, name in seq:
    dt = (t[:, b] - t[:, a]) * us
    print(f"{name:56s} {dt.mean():8.2f} {np.percentile(dt, 10):7.2f} {np.percentile(dt, 50):7.2f} {np.percentile(dt, 90):7.2f} {dt.mean() / life.mean():6.1%}")
