#!/usr/bin/env python3
"""HIP-event time of one E/M pass and of one loop iteration on a BASELINE-shaped store (A/B target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ctx = _lib.testing() if os.environ.get("OEM_USE_TESTING_LIB") == "1" else None
if ctx:
    ctx.__enter__()
st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    d.time_m_step(20)
    pm = min(d.time_m_step(50) for _ in range(3))
    it = min(d.time_em_iters(100) for _ in range(3)) / 100
    hbm, alg = d.bytes()
    print(f"{wl}: pass {pm:.4f} ms ({alg / pm / 1e6:.0f} GB/s, {alg / pm / 1e6 / 8000:.3f} of 8 TB/s), iteration {it:.4f} ms")
