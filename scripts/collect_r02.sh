#!/bin/bash
# Runs ON THE GPU BOX: the round-2 profile set that is copied into profiles/ (kernel stats of the three
# legs of bench.py, HBM traffic of the E/M pass with the FETCH_SIZE calibration).
out=gpurun_out/profiles_r02; mkdir -p $out
bash scripts/collect_profiles.sh r02 c3 > $out/collect_c3.log 2>&1
python scripts/hbm_traffic_json.py gpurun_out/profiles c3 $out/r02_c3_hbm_traffic.json r02 >> $out/collect_c3.log 2>&1
cp gpurun_out/profiles/r02_kernel_stats.csv $out/r02_c3_kernel_stats.csv
cp gpurun_out/profiles/r02_pmc_summary.txt $out/r02_c3_pmc_fetch_write_summary.txt
KT_TOP=6 bash scripts/kt.sh $out/boot python scripts/boot_bench.py c3 16 > $out/boot.log 2>&1
cp $out/boot/kernel_stats.csv $out/r02_c3_boot_kernel_stats.csv
KT_TOP=8 bash scripts/kt.sh $out/cells python scripts/cells_bench.py 625 50000 60000 > $out/cells.log 2>&1
cp $out/cells/kernel_stats.csv $out/r02_c5_cells625_kernel_stats.csv
tail -3 $out/collect_c3.log; cat $out/boot.log $out/cells.log
