#!/usr/bin/env python3
"""Summary of a rocprofv3 PC-sampling CSV: samples per kernel, and for the kernel named on the command line (default
k_em_tile) per instruction: share of samples, issued vs stalled, stall reasons.
usage: pc_sample_summary.py <pc_sampling.csv> <kernel_trace.csv> [kernel substring] [top n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()) if rows else None)
kname = {}
if len(sys.argv) > 2 and sys.argv[2]:
    for r in csv.DictReader(open(sys.argv[2])):
        kname[r["Dispatch_Id"]] = r["Kernel_Name"]
want = sys.argv[3] if len(sys.argv) > 3 else "k_em_tile"
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
def short(n):
    n = n.replace("void ", "").replace("oem::", "").replace("(anonymous namespace)::", "")
    return n.split("<")[0].split("(")[0]
per_k = collections.Counter(short(kname.get(r.get("Dispatch_Id", ""), "?")) for r in rows)
print(f"{len(rows)} samples; per kernel:", dict(per_k.most_common(8)))
sel = [r for r in rows if short(kname.get(r.get("Dispatch_Id", ""), "?")) == want]
print(f"\n{want}: {len(sel)} samples")
if not sel:
    sys.exit(0)
cols = sel[0].keys()
for c in ("Wave_Issued_Instruction", "Instruction_Type", "Stall_Reason", "Instruction_Not_Issued_Reason", "Snapshot_Stall_Reason", "Arb_State_Issue", "Arb_State_Stall"):
    if c in cols:
        cnt = collections.Counter(r[c] for r in sel)
        print(f"  {c}: " + ", ".join(f"{k}={v} ({100.0 * v / len(sel):.1f}%)" for k, v in cnt.most_common(12)))
ins = collections.defaultdict(list)
for r in sel:
    ins[(r.get("Instruction_Comment", ""), r.get("Instruction", ""))].append(r)
print(f"\n  top {top} instructions by samples (share; issued share; top stall reasons):")
reason_col = next((c for c in ("Instruction_Not_Issued_Reason", "Stall_Reason", "Snapshot_Stall_Reason") if c in cols), None)
for (cm, i), rs in sorted(ins.items(), key=lambda kv: -len(kv[1]))[:top]:
    issued = sum(1 for r in rs if r.get("Wave_Issued_Instruction", "") in ("1", "true", "True"))
    why = collections.Counter(r[reason_col] for r in rs if reason_col and r.get("Wave_Issued_Instruction", "") not in ("1", "true", "True")).most_common(2) if reason_col else []
    print(f"   {100.0 * len(rs) / len(sel):5.2f}%  issued {100.0 * issued / len(rs):5.1f}%  {i[:60]:60s} {cm[-50:]:50s} {why}")
