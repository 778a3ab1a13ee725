#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic of the per-cell loop (625-cell slice), FETCH_SIZE / WRITE_SIZE in their own passes
# plus the calibration stream; summary -> gpurun_out/cells_traffic/r03_c5_cells625_hbm_traffic.json
out=gpurun_out/cells_traffic; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "from oarfish_amd import build; build.build_microbench()" > /dev/null 2>&1
C="python scripts/cells_bench.py 625 50000 60000"
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/cal -o cal -- scripts/microbench/stream > $out/cal_stream.txt 2> $out/cal.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o r03 -- $C > $out/pf.out 2> $out/pf.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o r03 -- $C > $out/pw.out 2> $out/pw.err
python scripts/hbm_traffic_json.py $out c5 $out/r03_c5_cells625_hbm_traffic.json r03 cells
tail -1 $out/pf.out
