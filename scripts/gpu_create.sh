#!/bin/bash
for nt in 32 64 128 256; do
  echo "== OEM_HOST_THREADS=$nt"
  OEM_HOST_THREADS=$nt OEM_VERBOSE=1 timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --bootstraps 0 2>&1 | grep "oem\]" | tr '\n' ';' | sed 's/  */ /g'; echo
done
