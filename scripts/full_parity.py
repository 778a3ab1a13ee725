#!/usr/bin/env python3
"""Full-size parity run (checker: the C restatement under oracle/, serial do_em = em::em).

usage: full_parity.py <workload c2|c3> <out.json>
Runs the device EM and the oracle on the SAME full-size synthetic store with the reference's defaults
(max_iter 1000, conv_thresh 1e-3, gate 50) to their own convergence, and at a fixed iteration count,
and writes the worst relative abundance difference |a-b| / max(|b|, 1e-5*R/T) of each comparison.
Too slow for the test suite (the serial oracle needs minutes at C3); results are kept under profiles/.
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oarfish_amd.types import DeviceStore
from oracle import c_oracle

wl, out_path = sys.argv[1], sys.argv[2]
cfg = bench.WORKLOADS[wl]
row_ptr, tid, p, r0, r1 = bench.make_shard(cfg, 0, 1)
R, T = len(row_ptr) - 1, cfg["n_txps"]
floor = 1e-5 * R / T
res = {"workload": wl, "n_reads": R, "n_txps": T, "nnz": int(len(tid)), "tolerance": 1e-4, "runs": []}
o = c_oracle.Store(row_ptr, tid, p, None, T)
with DeviceStore(row_ptr, tid, p, None, T) as d:
    runs = [("fixed_60_iterations", dict(max_iter=60, conv_thresh=0.0)),
            ("defaults_to_convergence", dict(max_iter=1000, conv_thresh=1e-3))]
    if wl == "c2":   # BASELINE configs[1] as SURVEY.md 8d states it: 1000 iterations, no early exit
        runs.append(("fixed_1000_iterations", dict(max_iter=1000, conv_thresh=0.0)))
    for name, kw in runs:
        t = time.perf_counter(); got, gi = d.em_run(None, kw["max_iter"], kw["conv_thresh"], 50); tg = time.perf_counter() - t
        t = time.perf_counter(); want, wi = c_oracle.do_em(o, **kw); tc = time.perf_counter() - t
        err = np.abs(got - want) / np.maximum(np.abs(want), floor)
        res["runs"].append({"run": name, "device_niter": gi.niter, "oracle_niter": wi.niter,
                            "device_seconds": tg, "oracle_seconds_1core": tc,
                            "worst_rel_diff": float(err.max()), "n_over_1e-4": int((err > 1e-4).sum()),
                            "sum_device": float(got.sum()), "sum_oracle": float(want.sum())})
        print(res["runs"][-1], flush=True)
json.dump(res, open(out_path, "w"), indent=1)
