#!/bin/bash
# Runs ON THE GPU BOX: the round-6 profile set that is copied into profiles/.
#   1. E/M pass: kernel stats + FETCH_SIZE / WRITE_SIZE (own passes, calibrated) of the bench command
#   2. batched bootstrap AS SHIPPED (4 slots, one chain): kernel stats + HBM traffic of scripts/boot_passes.py
#      (what bench.py's bootstraps.roofline.kernel_avg_ms times), and the two-chain run of scripts/boot_bench.py
#   3. SQ / TCC counters of both (scripts/collect_pmc_cmd.sh)
#   4. the 625-cell slice
out=gpurun_out/profiles_r06; rm -rf $out gpurun_out/profiles; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r06 c3 > $out/collect_c3.log 2>&1
python scripts/hbm_traffic_json.py gpurun_out/profiles c3 $out/r06_c3_hbm_traffic.json r06 >> $out/collect_c3.log 2>&1
cp gpurun_out/profiles/r06_kernel_stats.csv $out/r06_c3_kernel_stats.csv
cp gpurun_out/profiles/r06_pmc_summary.txt $out/r06_c3_pmc_fetch_write_summary.txt
# the shipped bootstrap pass, one chain: kernel trace, then FETCH / WRITE in their own passes
B="python scripts/boot_passes.py c3 20"
KT_TOP=6 bash scripts/kt.sh $out/boot1 $B > $out/boot1.log 2>&1
cp $out/boot1/kernel_stats.csv $out/r06_c3_boot_passes_kernel_stats.csv
bp=gpurun_out/profiles_boot; rm -rf $bp; mkdir -p $bp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $bp/pf -o r06 -- $B > /dev/null 2> $bp/pf.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $bp/pw -o r06 -- $B > /dev/null 2> $bp/pw.err
mkdir -p $bp/cal; cp -r gpurun_out/profiles/cal/* $bp/cal/ 2>/dev/null
python scripts/hbm_traffic_json.py $bp c3 $out/r06_c3_boot_hbm_traffic.json r06 boot > $out/boot_traffic.log 2>&1
# two chains, as oem_bootstrap runs them
KT_TOP=6 bash scripts/kt.sh $out/boot2 python scripts/boot_bench.py c3 16 > $out/boot2.log 2>&1
cp $out/boot2/kernel_stats.csv $out/r06_c3_boot_two_chains_kernel_stats.csv
# SQ / TCC counters
bash scripts/collect_pmc_cmd.sh $out/pmc_pass "k_em_tile|k_remote_fold|k_reldiff" -- python scripts/pass_time.py c3 > /dev/null 2>&1
cp $out/pmc_pass/summary.txt $out/r06_c3_pmc_sq_tcc_summary.txt
bash scripts/collect_pmc_cmd.sh $out/pmc_boot "k_em_tile_e|k_remote_fold_b|k_reldiff_b" -- $B > /dev/null 2>&1
cp $out/pmc_boot/summary.txt $out/r06_c3_boot_pmc_summary.txt
# per-cell slice
KT_TOP=8 bash scripts/kt.sh $out/cells python scripts/cells_bench.py 625 50000 60000 > $out/cells.log 2>&1
cp $out/cells/kernel_stats.csv $out/r06_c5_cells625_kernel_stats.csv
tail -3 $out/collect_c3.log; cat $out/boot_traffic.log $out/boot1.log $out/boot2.log $out/cells.log
# per-cell loop traffic (FETCH / WRITE in their own passes)
sed 's/r03/r06/g' scripts/collect_cells_traffic.sh > /tmp/collect_cells_traffic_r06.sh; bash /tmp/collect_cells_traffic_r06.sh > $out/cells_traffic.log 2>&1
cp gpurun_out/cells_traffic/r06_c5_cells625_hbm_traffic.json $out/ 2>/dev/null
# SQ / TCC counters of the per-cell loop (averages over the loop's launches: the all-live passes and the thin tail)
bash scripts/collect_pmc_cmd.sh $out/pmc_cells "k_em_tile|k_multi_fold_reldiff" -- python scripts/cells_bench.py 625 50000 60000 > /dev/null 2>&1
cp $out/pmc_cells/summary.txt $out/r06_c5_cells625_pmc_summary.txt
# recurring far alignments (paralog families): kernel stats of the same pass
KT_TOP=4 bash scripts/kt.sh $out/paralog python scripts/pass_time.py c3 paralog > $out/paralog.log 2>&1
cp $out/paralog/kernel_stats.csv $out/r06_c3_paralog_kernel_stats.csv
# configs[1]: kernel stats of the same pass on the 1 M-read store, its loop trace, and the bench line of the C2 workload
KT_TOP=4 bash scripts/kt.sh $out/c2 python scripts/pass_time.py c2 > $out/c2.log 2>&1
cp $out/c2/kernel_stats.csv $out/r06_c2_kernel_stats.csv
bash scripts/loop_trace.sh $out/lt_c2 c2 200 20 > $out/r06_c2_loop_trace.txt 2>&1
python bench.py --workload c2 --steps 1000 --warmup 50 --bootstraps 0 --cells 0 --no-live-traffic --no-cpu-baseline 2> /dev/null | tail -1 > $out/r06_bench_c2.json
# in-kernel phase stamps (test-only library): where a tile's life goes, per phase (the stall-site account: no ATT decoder
# and no PC sampling on this image)
python scripts/tile_probe.py c3 2>/dev/null | grep -v amdgpu > $out/r06_tile_probe.txt
python scripts/tile_e_probe.py c3 2>/dev/null | grep -v amdgpu > $out/r06_tile_e_probe.txt
python scripts/shard_compute_time.py 2>/dev/null | grep "^N=" > $out/r06_shard_compute_time.txt
# gpurun copies back at most 64 MiB: the raw traces and counter tables stay on the box, the summaries travel
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
du -sh gpurun_out | tail -1
