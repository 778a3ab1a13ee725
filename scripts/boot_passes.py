#!/usr/bin/env python3
"""A few batched bootstrap passes on a C3-shaped store (profiling target).  usage: boot_passes.py [c3|c2] [n] [cov]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = _lib.testing() if os.environ.get("OEM_USE_TESTING_LIB") == "1" else None
if ctx:
    ctx.__enter__()
cov = len(sys.argv) > 3 and sys.argv[3] == "cov"   # f64 weights (the coverage model): k_em_tile_e<.., double, ..>
if cov:
    st = synth.make_store(10_000_000, 200_000, 8.0, coverage=True, threads=min(32, os.cpu_count() or 8))
else:
    st = synth.make_config(wl)
wc = int(os.environ.get("OEM_WEIGHT_CODING", "0"))   # 1: the f32 weight stream (oem_store_opts.weight_coding)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob if cov else None, st.n_txps, weight_coding=wc) as d:
    ms, slots, nbytes = d.time_bootstrap_passes(n)
    print(f"batched pass: {ms:.4f} ms for {slots} replicates = {ms / slots * 1e3:.1f} us per replicate-pass; "
          f"{nbytes / ms / 1e6:.0f} GB/s algorithmic")
