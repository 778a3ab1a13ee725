cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/boot_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/boot_prof -o boot -- python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --bootstraps 4 > gpurun_out/boot_prof/bench.json 2>/dev/null
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/boot_prof/boot_kernel_stats.csv")):
    print(r["Name"][:70].replace("\n"," "), r["Calls"], "%.1f us avg" % (float(r["AverageNs"])/1e3), r["Percentage"], "total ms %.1f" % (float(r["TotalDurationNs"])/1e6))
PY
python -c "import json; d=json.load(open('gpurun_out/boot_prof/bench.json')); print(d['bootstraps'])"
