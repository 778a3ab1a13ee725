#!/bin/bash
# Runs ON THE GPU BOX: A/B timings of round 3 through the test-only library's knobs.  usage: r03_ab.sh <outdir>
out=${1:-gpurun_out/r03_ab}; mkdir -p $out
export OEM_USE_TESTING_LIB=1
for wl in c3 c2; do
  echo "== $wl: separate fold + rel-diff (OEM_FUSED_FOLD=0)";       OEM_FUSED_FOLD=0 python scripts/pass_time.py $wl
  echo "== $wl: fold finishes the iteration (default)";             python scripts/pass_time.py $wl
  echo "== $wl: + chunks of 16 iterations from a hipGraph";         OEM_GRAPH=1 python scripts/pass_time.py $wl
  echo "== $wl: hipGraph, separate kernels";                        OEM_GRAPH=1 OEM_FUSED_FOLD=0 python scripts/pass_time.py $wl
done 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee $out/ab.txt
