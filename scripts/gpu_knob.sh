#!/bin/bash
# usage: scripts/gpu_knob.sh ENV_NAME "v1 v2 ..." "workloads"   -- pass time of the product build per knob value
for v in $2; do
  for wl in ${3:-c3}; do
    env $1=$v timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --bootstraps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1=$v', '$wl', 'it/s %.0f' % d['value'], 'pass_ms %.4f' % d['roofline']['kernel_avg_ms'], 'step_ms %.4f' % d['ms_per_step'])"
  done
done
