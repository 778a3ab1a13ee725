import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oarfish_amd import _lib, synth
from oarfish_amd.types import DeviceStore
_lib.lib()
for R, T in ((100_000, 200_000), (200_000, 200_000), (250_000, 300_000), (400_000, 200_000), (120_000, 60_000), (2_000_000, 4_000_000)):
    st = synth.make_store(R, T, 8.0, threads=32)
    out = []
    for cap in (512, 2048):
        d = DeviceStore(st.row_ptr, st.tid, st.as_prob, None, T, window_cap=cap)   # oem_store_opts.window_cap
        d.time_em_iters(10)
        out.append(d.time_em_iters(200) / 200)
        d.close()
    print(f"R={R} T={T} density {R/T:.1f}: iteration ms cap512 {out[0]:.4f} cap2048 {out[1]:.4f} -> {'wide' if out[1] < out[0] else 'narrow'} wins")
