#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the loop bench.py times (scripts/loop_trace.py), summarised per segment.
# usage: loop_trace.sh <outdir> [c3|c2] [steps] [warmup]     (OEM_USE_TESTING_LIB=1 OEM_DEFERRED_RELDIFF=0: the classic loop)
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o lt -- python scripts/loop_trace.py "$@" > $out/cmd.out 2> $out/cmd.err
grep -v amdgpu.ids $out/cmd.out
f=$(find $out/kt -name "lt_kernel_trace.csv" | head -1)
python scripts/loop_trace_summary.py $f > $out/summary.txt
cat $out/summary.txt
rm -f $f
