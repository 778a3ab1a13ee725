#!/bin/bash
# Builds the two libraries with extra -D switches into a side directory (for OEM_AB_DIR=<dir>: two builds timed side
# by side in one gpurun call), then restores the default build.  usage: build_variant.sh <dir> "<-DX=1 ...>"
# (OEM_AB_ALL=1: every object is rebuilt with the switches, not only the kernels')
set -e
dir=$1; defs=$2
cd "$(dirname "$0")/.."
stale() {
  if [ -n "$OEM_AB_ALL" ]; then rm -f oarfish_amd/csrc/_obj/*.o
  else rm -f oarfish_amd/csrc/_obj/oem_tile_kernels*.o oarfish_amd/csrc/_obj/oem_batch_kernels*.o oarfish_amd/csrc/_obj/oem_multi_kernels*.o; fi
}
stale
OEM_EXTRA_DEFS="$defs" python -m oarfish_amd.build > /dev/null 2>&1
mkdir -p "$dir"; cp oarfish_amd/liboarfish_em.so oarfish_amd/liboarfish_em_testing.so "$dir"/
stale
python -m oarfish_amd.build > /dev/null 2>&1
echo "built $dir with $defs"
