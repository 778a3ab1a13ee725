#!/bin/bash
# timing experiment: what would one-byte (dictionary-coded) weights buy?  results are garbage values
for bw in "" 1; do
  OEM_TILE_ABLATION=1 OEM_ABL_BYTEW=$bw python -m oarfish_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  for wl in c3 c2; do
    OEM_TILE_ABLATE=0 timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --bootstraps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('bytew=$bw', '$wl', 'it/s %.0f' % d['value'], 'pass_ms %.4f' % d['roofline']['kernel_avg_ms'])"
  done
done
python -m oarfish_amd.build --force > /dev/null 2>&1
