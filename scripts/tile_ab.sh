#!/bin/bash
# Runs ON THE GPU BOX: A/B of tile-kernel experiment knobs through the test-only library.
export OEM_USE_TESTING_LIB=1
run() { echo -n "$1: "; env $1 python scripts/pass_time.py ${WL:-c3} 2>/dev/null | tail -1; }
run "OEM_TILE_SORT=0"
run "OEM_TILE_SORT=1"
run "OEM_TILE_UNMASK=1"
run "OEM_TILE_SORT=0 OEM_TILE_UNMASK=1"
