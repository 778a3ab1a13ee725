#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter, mean over dispatches."""
import csv, glob, re, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"  {k:42s} {c:24s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
