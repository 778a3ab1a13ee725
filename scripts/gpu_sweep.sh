#!/bin/bash
# usage: scripts/gpu_sweep.sh "<variants>" [workload]   (run on the GPU box)
wl=${2:-c3}
for v in $1; do
  OEM_TILE_VARIANT=$v timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --bootstraps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('variant', $v, 'it/s %.0f' % d['value'], 'pass_ms %.4f' % d['roofline']['kernel_avg_ms'], 'frac %.3f' % d['roofline']['frac'])"
done
