#!/bin/bash
# register / LDS / occupancy of every kernel of one source file (compiler remarks): scripts/kres.sh oem_batch_kernels.hip [-DOEM_TESTING ...]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -I include -x hip -c oarfish_amd/csrc/$f \
  -Rpass-analysis=kernel-resource-usage "$@" -o /dev/null 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//; s/.*remark: *//' | \
  awk -F': ' '/^Function Name/{n=$2} /^VGPRs:/{v=$2} /^AGPRs/{a=$2} /^ScratchSize/{s=$2} /^Occupancy/{o=$2} /^LDS Size/{print n" vgpr="v" agpr="a" scratch="s" occ="o" lds="$2}' | \
  while read n rest; do echo "$(echo "$n" | c++filt | sed 's/oem::(anonymous namespace):://; s/^void //; s/(oem::TileDesc const\*.*//; s/(unsigned.*//; s/(double.*//' | cut -c1-110) $rest"; done
