#!/usr/bin/env python3
"""The launches bench.py times, in bench.py's order, for a kernel trace (scripts/loop_trace.sh): W warm-up iterations,
K timed iterations, 50 back-to-back passes (oem_time_m_step), then 200 iterations -- separated by host syncs so that
scripts/loop_trace_summary.py can tell the segments apart.  usage: loop_trace.py [c3|c2] [steps] [warmup]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = _lib.testing() if os.environ.get("OEM_USE_TESTING_LIB") == "1" else None
if ctx:
    ctx.__enter__()
st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    time.sleep(0.05)
    d.time_em_iters(warm)
    time.sleep(0.002)
    t = time.perf_counter(); ms = d.time_em_iters(steps); host = (time.perf_counter() - t) * 1e3
    print(f"{wl}: {steps} iterations: device {ms:.4f} ms = {ms / steps:.5f} ms/step, host {host:.4f} ms = {host / steps:.5f} ms/step")
    time.sleep(0.002)
    pm = d.time_m_step(50)
    print(f"{wl}: pass (50 back to back) {pm:.5f} ms")
    time.sleep(0.002)
    ms = d.time_em_iters(200)
    print(f"{wl}: 200 iterations: device {ms / 200:.5f} ms/step")
    time.sleep(0.002)
    t = time.perf_counter(); ms = d.time_em_iters(steps); host = (time.perf_counter() - t) * 1e3
    print(f"{wl}: {steps} iterations again (device busy just before): device {ms / steps:.5f} ms/step, host {host / steps:.5f} ms/step")
