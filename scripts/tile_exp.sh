#!/bin/bash
# Runs ON THE GPU BOX: cost attribution of the two tile kernels (test-only library: OEM_TILE_EXP switches parts of
# the kernels off; the results are wrong, the times say what each part costs).  usage: tile_exp.sh [c3]
wl=${1:-c3}
export OEM_USE_TESTING_LIB=1
for m in 0 1 2 4 8 16 3 19 12 31; do
  echo "== OEM_TILE_EXP=$m (1 queue stores, 2 remote denominator atomics, 4 local scatter atomics, 8 local theta reads, 16 remote gathers)"
  OEM_TILE_EXP=$m python scripts/pass_time.py $wl 2>/dev/null | tail -1
  OEM_TILE_EXP=$m python scripts/boot_passes.py $wl 20 2>/dev/null | tail -1
done
