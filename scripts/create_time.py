#!/usr/bin/env python3
"""Wall time of oem_store_create on a BASELINE-shaped store, repeated (OEM_VERBOSE=1 prints the stages)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
st = synth.make_config(wl)
_lib.lib()
for rep in range(4):
    t = time.perf_counter()
    d = DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
    dt = time.perf_counter() - t
    print(f"create #{rep}: {dt * 1e3:.1f} ms", flush=True)
    d.close()
