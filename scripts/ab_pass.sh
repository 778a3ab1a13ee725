#!/bin/bash
# Interleaved A/B of the C3 E/M pass between the in-tree build and snapshot builds (OEM_AB_DIR), one gpurun call.
# usage: ab_pass.sh <reps> <workload> <kind> <coding> [dir ...]   ("." = the in-tree build)
reps=$1; wl=$2; kind=$3; coding=$4; shift 4
for r in $(seq $reps); do
  for d in "$@"; do
    if [ "$d" = "." ]; then echo -n "[in-tree] "; python scripts/pass_time.py $wl $kind $coding 2>/dev/null | tail -1
    else echo -n "[$d] "; OEM_AB_DIR=$d python scripts/pass_time.py $wl $kind $coding 2>/dev/null | tail -1; fi
  done
done
