#!/usr/bin/env python3
"""Experiment: do K independent bootstrap chains on K HIP streams overlap (one chain's streaming fold /
rel-diff kernels under another's tile kernel)?  K resident copies of the store, one thread each."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oarfish_amd import synth
from oarfish_amd.types import DeviceStore
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
st = synth.make_config(wl)
stores = []
for K in (1, 2, 3, 4):
    while len(stores) < K:
        d = DeviceStore(st.row_ptr, st.tid, st.as_prob, None, st.n_txps)
        d.bootstrap(2, seed=99, max_iter=2)
        stores.append(d)
    res = [None] * K
    def run(i, d):
        res[i] = d.bootstrap(per, seed=1, first_replica=i * per)
    t = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, d)) for i, d in enumerate(stores[:K])]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t
    print(f"{K} chain(s) x {per} replicates: {dt:.2f} s = {K * per / dt:.2f} /s", flush=True)
for d in stores:
    d.close()
