#!/usr/bin/env python3
"""Per-launch begin / end stamps of a rocprofv3 --kernel-trace run of scripts/loop_trace.py, as segments of back-to-back
passes: per segment the kernels' durations, the gaps between launches and the span per pass; for short segments the
per-launch series.  usage: loop_trace_summary.py <kernel_trace.csv> [name substring of the pass kernel, default k_em_tile]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "k_em_tile"
def short(n):
    n = n.replace("void ", "").replace("oem::", "").replace("(anonymous namespace)::", "")
    return n.split("<")[0].split("(")[0]
# segments: launches closer than 200 us to the previous launch's end
segs, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 200_000:
        segs.append(cur); cur = []
    cur.append(r)
if cur:
    segs.append(cur)
print(f"{len(rows)} launches, {len(segs)} segments (split where the device idled > 200 us)")
for si, seg in enumerate(segs):
    n_pass = sum(1 for r in seg if key in r[2])
    if n_pass < 3:
        continue
    first = next(i for i, r in enumerate(seg) if key in r[2])
    body = seg[first:]
    span = (body[-1][1] - body[0][0]) / 1e3
    by = {}
    for r in body:
        by.setdefault(short(r[2]), []).append((r[1] - r[0]) / 1e3)
    gaps = [(body[i + 1][0] - body[i][1]) / 1e3 for i in range(len(body) - 1)]
    print(f"\nsegment {si}: {n_pass} passes, {len(body)} launches, span {span:.1f} us = {span / n_pass:.2f} us per pass")
    for k, v in by.items():
        print(f"   {k:34s} n={len(v):4d}  avg {sum(v) / len(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}  sum {sum(v):9.1f}")
    if gaps:
        print(f"   gaps between launches              n={len(gaps):4d}  avg {sum(gaps) / len(gaps):8.2f} us  min {min(gaps):8.2f}  max {max(gaps):8.2f}  sum {sum(gaps):9.1f}")
    if n_pass <= 32:
        print("   per launch (start offset us, duration us, gap to next us):")
        for i, r in enumerate(body):
            g = gaps[i] if i < len(gaps) else 0.0
            print(f"     {short(r[2]):30s} {(r[0] - body[0][0]) / 1e3:9.1f} {(r[1] - r[0]) / 1e3:8.2f} {g:7.2f}")
