#!/bin/bash
# usage: scripts/gpu_ablate.sh "<ablate masks>" [variant] [workload]   (timing experiments; results invalid)
wl=${3:-c3}
# needs a profiling build: OEM_TILE_ABLATION=1 python -m oarfish_amd.build --force
OEM_TILE_ABLATION=1 python -m oarfish_amd.build --force > /dev/null 2>&1
for a in $1; do
  OEM_TILE_ABLATE=$a OEM_TILE_VARIANT=${2:-4} timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --bootstraps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('ablate', $a, 'it/s %.0f' % d['value'], 'pass_ms %.4f' % d['roofline']['kernel_avg_ms'])"
done
python -m oarfish_amd.build --force > /dev/null 2>&1  # back to the product build
