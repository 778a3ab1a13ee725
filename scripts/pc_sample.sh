#!/bin/bash
# Runs ON THE GPU BOX: stochastic PC sampling (rocprofv3 --pc-sampling-beta-enabled; gfx950 reports issue / stall reasons per
# sample) of a command, summarised per kernel and per instruction by scripts/pc_sample_summary.py.
# usage: pc_sample.sh <outdir> <interval cycles> <cmd ...>
out=$1; interval=$2; shift 2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 60 rocprofv3-avail list --pc-sampling > $out/avail.txt 2>&1 || timeout 60 rocprofv3-avail info --pc-sampling >> $out/avail.txt 2>&1
head -30 $out/avail.txt
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${PCS_METHOD:-stochastic} --pc-sampling-unit ${PCS_UNIT:-cycles} --pc-sampling-interval $interval \
    --kernel-trace --output-format csv -d $out/pcs -o pcs -- "$@" > $out/cmd.out 2> $out/cmd.err
echo "rocprofv3 rc=$?"
grep -v amdgpu.ids $out/cmd.out | tail -3; tail -5 $out/cmd.err
find $out/pcs -type f | head; 
f=$(find $out/pcs -name "*pc_sampling*.csv" | head -1)
[ -n "$f" ] && { head -3 $f; wc -l $f; python scripts/pc_sample_summary.py $f $(find $out/pcs -name "*kernel_trace.csv" | head -1) > $out/summary.txt 2>&1; head -120 $out/summary.txt; }
