#!/usr/bin/env python3
"""How much of a pass's fold (k_remote_fold) hides under a tile kernel?  Test-only library, oem_debug_overlap_probe:
tile kernel alone, fold alone, the pass, and fold i on a second stream behind tile i while tile i + 1 runs.
usage: overlap_probe.py [c3|c2] [launches]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
with _lib.testing():
    L = _lib.lib()
    L.oem_debug_overlap_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    st = synth.make_config(wl)
    with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
        for rnd in range(3):
            out = np.zeros(7)
            _lib.check(L.oem_debug_overlap_probe(d.handle, n, out.ctypes.data))
            print(f"{wl} x{n}: tile {out[0]:.1f} us, fold {out[1]:.1f} us, pass (one stream) {out[2]:.1f} us, "
                  f"fold on a second stream under the next tile kernel {out[3]:.1f} us per iteration "
                  f"(hidden: {out[2] - out[3]:.1f} of {out[1]:.1f} us); the fold in 256-thread workgroups: alone {out[4]:.1f} us, "
                  f"tile + it on one stream {out[6]:.1f} us, on the second stream {out[5]:.1f} us per iteration "
                  f"(hidden: {out[6] - out[5]:.1f} of {out[4]:.1f} us)")
