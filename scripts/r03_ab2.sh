#!/bin/bash
# Runs ON THE GPU BOX: occupancy variants of k_em_tile through the test-only library.  usage: r03_ab2.sh <outdir>
out=${1:-gpurun_out/r03_ab2}; mkdir -p $out
export OEM_USE_TESTING_LIB=1
for wl in c3 c2; do
  echo "== $wl: as built (92 VGPRs, 4 copies, 5 workgroups per CU)";                       python scripts/pass_time.py $wl
  echo "== $wl: variant 2: remote products parked in the queue (80 VGPRs), 4 copies, 5/CU"; OEM_TILE_VARIANT=2 python scripts/pass_time.py $wl
  echo "== $wl: variant 3: 3 copies only (24 KiB LDS), 5/CU";                               OEM_TILE_VARIANT=3 python scripts/pass_time.py $wl
  echo "== $wl: variant 1: parked + 3 copies: 6 workgroups per CU";                         OEM_TILE_VARIANT=1 python scripts/pass_time.py $wl
  echo "== $wl: variant 4: parked + 2 copies: 6 workgroups per CU";                         OEM_TILE_VARIANT=4 python scripts/pass_time.py $wl
done 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee $out/ab2.txt
