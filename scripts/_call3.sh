for r in 1 2 3; do
  for v in _ab_base ""; do
    echo "== variant '$v'" >> gpurun_out/s3_cells.txt
    OEM_VERBOSE=1 OEM_AB_DIR=$v python scripts/cells_bench.py 625 50000 60000 2>&1 | grep -E "EM loop|batched|ab\]" >> gpurun_out/s3_cells.txt
  done
done
cat gpurun_out/s3_cells.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "cell or c5 or multi" > gpurun_out/s3_tests.log 2>&1; tail -3 gpurun_out/s3_tests.log
for v in _ab_base ""; do OEM_AB_DIR=$v python scripts/pass_time.py c3 2>&1 | tail -3; done
