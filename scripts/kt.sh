#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of a command, prints the top kernels.  usage: kt.sh <outdir> <cmd ...>
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- "$@" > $out/cmd.out 2> $out/cmd.err
cat $out/cmd.out | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
python - <<PY
import csv,glob
f=glob.glob("$out/kt/**/kt_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${KT_TOP:-8}]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%10.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
cp $(ls $out/kt/*/kt_kernel_stats.csv $out/kt/kt_kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv 2>/dev/null
