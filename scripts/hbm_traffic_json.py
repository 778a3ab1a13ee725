#!/usr/bin/env python3
"""Turn the separate FETCH_SIZE / WRITE_SIZE / calibration passes of scripts/collect_profiles.sh
into profiles/<tag>_<wl>_hbm_traffic.json (HBM bytes per launch, per kernel; the file bench.py's
roofline.traffic is read from).
usage: hbm_traffic_json.py <collect dir> <workload> <out.json> <tag> [boot|cells]
  (boot: the kernels of the batched bootstrap pass, collected on scripts/boot_passes.py;
   cells: the per-cell loop of scripts/cells_bench.py -- per-launch means AND totals over the loop's launches,
   since the traffic of a pass falls as cells finish)"""
import collections, csv, hashlib, json, os, re, sys

root, wl, out = sys.argv[1], sys.argv[2], sys.argv[3]
KNOWN_KB = 1.5 * 1024 * 1024  # scripts/microbench/stream reads 1.5 GiB per launch


COUNTS = {}


def kernel_source_sha():
    """The kernel sources the counters were collected on (bench.py compares it with the tree it runs in)."""
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("oem_tile_kernels.hip", "oem_tile_common.h", "oem_lane_runs.h", "oem_batch_kernels.hip", "oem_layout.h"):
        with open(os.path.join(root_dir, "oarfish_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def means(path):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            if m:
                acc[m.group(1)].append(float(r["Counter_Value"]))
    COUNTS[path] = {k: len(v) for k, v in acc.items()}
    return {k: sum(v) / len(v) for k, v in acc.items()}


cal = means(f"{root}/cal/cal_counter_collection.csv")["k_stream"]
tag = [a for a in sys.argv[4:5]] or ["r01"]
boot = len(sys.argv) > 5 and sys.argv[5] == "boot"
cells = len(sys.argv) > 5 and sys.argv[5] == "cells"
rd = means(f"{root}/pf/{tag[0]}_counter_collection.csv")
wr = means(f"{root}/pw/{tag[0]}_counter_collection.csv")
factor = KNOWN_KB / cal
kern = {}
names = ("k_em_tile_e", "k_remote_fold_b", "k_reldiff_b") if boot else ("k_em_tile", "k_remote_fold", "k_reldiff_swap_clear")
if cells:
    names = ("k_em_tile", "k_multi_fold_reldiff", "k_multi_decide")
names = tuple(k for k in names if k in rd and k in wr)   # (single-device stores no longer launch the sweep kernel)
for k in names:
    kern[k] = {"read": rd[k] * 1024 * factor, "write": wr[k] * 1024}
if cells:
    n = COUNTS[f"{root}/pf/{tag[0]}_counter_collection.csv"]
    doc = {"workload": wl, "command": "python scripts/cells_bench.py 625 50000 60000",
           "kernel_source_sha": kernel_source_sha(), "fetch_factor": factor, "per_launch_mean_bytes": kern, "launches": {k: n[k] for k in names},
           "loop_total_bytes": sum((kern[k]["read"] + kern[k]["write"]) * n[k] for k in names)}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))
    sys.exit(0)
doc = {
    "workload": wl,
    "kernel_source_sha": kernel_source_sha(),
    "command": (f"python scripts/boot_passes.py {wl} 20  (4-slot batched passes, all slots running, one chain)" if boot else
                f"python bench.py --workload {wl} --steps 50 --warmup 5 --no-cpu-baseline --no-f32-compare --bootstraps 0 --cells 0"),
    "fetch_calibration": {
        "kernel": "scripts/microbench/stream (rows of 64 lanes, 4/8/12/16 B per lane)",
        "known_KB_per_launch": KNOWN_KB, "FETCH_SIZE_KB": cal, "factor": factor,
        "note": "gfx950 FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md, HBM section): corrected "
                "bytes = FETCH_SIZE*1024*factor; WRITE_SIZE is taken as reported (uncalibrated)"},
    "per_launch_bytes": kern,
    ("per_batched_pass_bytes_total" if boot else "per_pass_bytes_total"): sum(v["read"] + v["write"] for v in kern.values()),
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["per_launch_bytes"]), sum(v["read"] + v["write"] for v in kern.values()))
