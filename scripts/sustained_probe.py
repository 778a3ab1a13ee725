#!/usr/bin/env python3
"""Is a pass of a long run slower because the run is long (power) or because theta has moved (data)?  The same pass with
the uniform start vector for 100 / 900 / 3000 launches back to back, and real iterations 1-200 against 700-900."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oarfish_amd import synth
from oarfish_amd.types import DeviceStore
st = synth.make_config("c3")
with DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps) as d:
    d.time_m_step(300)
    for n in (100, 900, 3000, 100, 900):
        print(f"uniform theta, {n} passes back to back: {d.time_m_step(n) * 1e3:.2f} us per pass")
    for n in (100, 200, 890, 200, 890):
        print(f"real iterations 1..{n}: {d.time_em_iters(n) / n * 1e3:.2f} us each")
    # passes over the CONVERGED abundances: m_step launches with theta = the result of a run
    cnt, info = d.em_run(None, 1000, 1e-3, 50)
    t = time.perf_counter()
    for _ in range(20):
        d.m_step(cnt)
    print(f"(m_step API with converged theta incl. 2 x 1.6 MB transfers: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call)")
