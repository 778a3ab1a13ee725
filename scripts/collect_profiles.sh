#!/bin/bash
# Runs ON THE GPU BOX (via gpurun).  Collects, for the bench command:
#   1. rocprofv3 --kernel-trace --stats           -> gpurun_out/profiles/<tag>_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)     -> <tag>_pmc_fetch.csv
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass)     -> <tag>_pmc_write.csv
#   4. FETCH_SIZE calibration on a known byte count in the same access pattern (4 B/lane rows
#      of 64 lanes): scripts/microbench/stream reads 1.5 GiB per launch.
# Each rocprofv3 call is wrapped in its own short timeout.
tag=${1:-r01}
wl=${2:-c3}
out=gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-f32-compare --no-live-traffic --no-side-legs --bootstraps 0 --cells 0"
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o $tag -- $B > $out/${tag}_bench_under_rocprof.json 2> $out/kt.err
cp $out/kt/${tag}_kernel_stats.csv $out/${tag}_kernel_stats.csv 2>/dev/null
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o $tag -- $B > /dev/null 2> $out/pf.err
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o $tag -- $B > /dev/null 2> $out/pw.err
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/cal -o cal -- scripts/microbench/stream > $out/cal_stream.txt 2> $out/cal.err
python scripts/pmc_summary.py $out > $out/${tag}_pmc_summary.txt
cat $out/${tag}_pmc_summary.txt | grep -E "==|k_em_tile|k_remote|k_reldiff|k_stream" | head -40
cat $out/${tag}_kernel_stats.csv | cut -c1-50,180-330 | head -6
