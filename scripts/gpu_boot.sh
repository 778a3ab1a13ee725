#!/bin/bash
# usage: scripts/gpu_boot.sh "<batch variants>" n_boot   -- batched bootstrap throughput on C3; "u" = unbatched
for v in $1; do
  if [ "$v" = "u" ]; then X="--no-batch-bootstrap"; else X=""; fi
  OEM_BATCH_VARIANT=$v timeout 300 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --bootstraps ${2:-4} $X 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('batch variant', '$v', d['bootstraps'])"
done
