#!/usr/bin/env python3
"""oem_em_run end to end against the same passes timed on the device: where the wall time of a run to convergence goes.
usage: emrun_probe.py [c3|c2]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401
from oarfish_amd import synth
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    d.time_m_step(300)
    pm = min(d.time_m_step(100) for _ in range(3))
    n = 890 if wl == "c3" else 1000
    it = min(d.time_em_iters(n) for _ in range(2))
    best_fix, best_conv = 1e9, 1e9
    for _ in range(3):
        t = time.perf_counter(); cnt, info = d.em_run(None, n, 0.0, 50); best_fix = min(best_fix, time.perf_counter() - t)
        t = time.perf_counter(); cnt, info2 = d.em_run(None, 1000, 1e-3, 50); best_conv = min(best_conv, time.perf_counter() - t)
    print(f"{wl}: pass {pm * 1e3:.2f} us; {n} iterations on the device (oem_time_em_iters) {it:.2f} ms = {it / n * 1e3:.2f} us each; "
          f"oem_em_run({n}, thresh 0) {best_fix * 1e3:.2f} ms = {best_fix / info.n_passes * 1e6:.2f} us per pass; "
          f"oem_em_run to convergence ({info2.niter} iterations, {info2.n_passes} passes) {best_conv * 1e3:.2f} ms = {best_conv / info2.n_passes * 1e6:.2f} us per pass")
