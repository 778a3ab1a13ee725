import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import _ab  # noqa
from oarfish_amd import synth
from oarfish_amd.types import DeviceStore
wl = sys.argv[1]
st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    d.em_run(None, 100, 0.0, 50)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); cnt, info = d.em_run(None, 1000, 1e-3 if wl == "c3" else 0.0, 50); dt = time.perf_counter() - t
        best = min(best, dt)
    print(f"{wl}: em_run {info.niter} iterations in {best*1e3:.2f} ms = {best/info.n_passes*1e6:.2f} us per pass")
