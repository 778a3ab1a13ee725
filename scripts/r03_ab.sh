#!/bin/bash
# Runs ON THE GPU BOX: A/B timings of round 3 through the test-only library's knobs.  usage: r03_ab.sh <outdir>
out=${1:-gpurun_out/r03_ab}; mkdir -p $out
export OEM_USE_TESTING_LIB=1
for wl in c3 c2; do
  echo "== $wl: direct launches (OEM_GRAPH=0), interleaved count-window copies (OEM_TILE_PLANAR=0)"; OEM_GRAPH=0 OEM_TILE_PLANAR=0 python scripts/pass_time.py $wl
  echo "== $wl: direct launches, planar copies";        OEM_GRAPH=0 python scripts/pass_time.py $wl
  echo "== $wl: graph replay, interleaved copies";      OEM_TILE_PLANAR=0 python scripts/pass_time.py $wl
  echo "== $wl: graph replay, planar copies (default)"; python scripts/pass_time.py $wl
done 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee $out/ab.txt
