#!/bin/bash
# Runs ON THE GPU BOX: k_em_tile cut short behind each of its phases (test-only library, OEM_TILE_EXP 128 / 256 / 512:
# the results are wrong, the time says what the tiles cost up to there at the real occupancy; +16 no theta gathers, +4096 no
# slice requested, +8192 no record requested).  usage: [MASKS="0 128 144 ..."] tile_phase_exp.sh [c3]
wl=${1:-c3}
export OEM_USE_TESTING_LIB=1
for m in ${MASKS:-0 2048 1024 128 256 512 528}; do
  echo "== OEM_TILE_EXP=$m (2048: exit at once; 1024: with the descriptor; 128: exit behind barrier 1 = loads, gathers issued, LDS init; 256: behind barrier 2 = + gathers landed, phase A; 512: behind barrier 3 = + the folds; +16: no gathers)"
  OEM_TILE_EXP=$m KT_TOP=2 bash scripts/kt.sh gpurun_out/phase_$m python scripts/pass_time.py $wl 2>/dev/null | tail -3
done
