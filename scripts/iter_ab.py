#!/usr/bin/env python3
"""A/B helper: plain pass against loop iterations (200 / 890 / 200 / 890) on C3 or C2.  usage: iter_ab.py [c3|c2]  (OEM_AB_DIR)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import synth
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    d.time_m_step(300)
    pm = min(d.time_m_step(200) for _ in range(3))
    ns = (200, 890, 200, 890) if wl == "c3" else (1000, 1000, 1000, 1000)
    it = [d.time_em_iters(n) / n * 1e3 for n in ns]
    print(f"{wl}: pass {pm*1e3:.2f} us; iterations {ns}: " + ", ".join(f"{x:.2f}" for x in it))
