#!/usr/bin/env python3
"""HIP-event time of one E/M pass and of one loop iteration on a BASELINE-shaped store (A/B target).
usage: pass_time.py [c3|c2] [geometric|uniform|coverage|paralog|paralog_adjacent] [weight_coding 0|1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _ab  # noqa: E401,E402,F401  (OEM_AB_DIR: A/B against a snapshot build)
from oarfish_amd import synth, _lib
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
kind = sys.argv[2] if len(sys.argv) > 2 else "geometric"
coding = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = _lib.testing() if os.environ.get("OEM_USE_TESTING_LIB") == "1" else None
if ctx:
    ctx.__enter__()
cfg = synth.CONFIGS[wl]
if kind == "coverage":
    st = synth.make_store(coverage=True, threads=16, **cfg)
elif kind == "uniform":
    st = synth.make_store(gaps="uniform", threads=16, **cfg)
elif kind in ("paralog", "paralog_adjacent"):
    st = synth.make_store(far=kind, threads=16, **cfg)
else:
    st = synth.make_config(wl)
with DeviceStore(st.row_ptr, st.tid, st.as_prob, st.cov_prob if kind == "coverage" else None, st.n_txps, weight_coding=coding) as d:
    d.time_m_step(20)
    pm = min(d.time_m_step(50) for _ in range(3))
    it = min(d.time_em_iters(100) for _ in range(3)) / 100
    hbm, alg = d.bytes()
    print(f"{wl} {kind} coding={coding} dict={d.info(_lib.OEM_INFO_WEIGHT_DICT_ENTRIES)} remote={d.info(_lib.OEM_INFO_REMOTE_ALIGNMENTS)}: pass {pm:.4f} ms ({alg / pm / 1e6:.0f} GB/s, "
          f"{alg / pm / 1e6 / 8000:.3f} of 8 TB/s), iteration {it:.4f} ms")
