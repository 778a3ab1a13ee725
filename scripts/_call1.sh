python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
python scripts/overlap_probe.py c3 200 > gpurun_out/s1_overlap.txt 2>&1
python scripts/overlap_probe.py c2 400 >> gpurun_out/s1_overlap.txt 2>&1
for r in 1 2; do
  for v in "" _ab_qsc1 _ab_qsc2 _ab_qsc3; do
    echo "== variant '$v'" >> gpurun_out/s1_qsc.txt
    OEM_AB_DIR=$v python scripts/boot_passes.py c3 40 >> gpurun_out/s1_qsc.txt 2>&1
  done
done
python scripts/boot_size_probe.py "2500000 5000000 10000000" > gpurun_out/s1_bootsize.txt 2>&1
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/s1_tests.log 2>&1
tail -3 gpurun_out/s1_tests.log; cat gpurun_out/s1_overlap.txt gpurun_out/s1_qsc.txt gpurun_out/s1_bootsize.txt
