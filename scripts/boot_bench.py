#!/usr/bin/env python3
"""Times the batched bootstrap on a C3-shaped store: per batched pass (HIP events) and end to end."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
import numpy as np
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_boot = int(sys.argv[2]) if len(sys.argv) > 2 else 16
use_testing = os.environ.get("OEM_USE_TESTING_LIB") == "1"
ctx = _lib.testing() if use_testing else None
if ctx:
    ctx.__enter__()
st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    ms, slots, nbytes = d.time_bootstrap_passes(20)
    print(f"batched pass: {ms:.4f} ms for {slots} replicates = {ms / slots * 1e3:.1f} us per replicate-pass; "
          f"algorithmic {nbytes / 1e6:.0f} MB -> {nbytes / ms / 1e6:.0f} GB/s = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s")
    d.bootstrap(2, seed=99, max_iter=2)
    t = time.perf_counter()
    out, infos = d.bootstrap(n_boot, seed=1, max_iter=1000, conv_thresh=1e-3)
    dt = time.perf_counter() - t
    print(f"{n_boot} bootstraps in {dt:.2f} s = {n_boot / dt:.2f} /s, mean passes {np.mean([i.n_passes for i in infos]):.0f}")
    pm = d.time_m_step(20)
    print(f"point-estimate pass {pm:.4f} ms")
