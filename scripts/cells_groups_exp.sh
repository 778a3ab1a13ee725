#!/bin/bash
# Runs ON THE GPU BOX: the 625-cell slice with a head of the cells cut off as a group of its own (test-only library,
# OEM_CELLS_HEAD = cells in the head; 0 = one group), so that the rest uploads under the head's EM loop.
for rep in 1 2 3; do for h in 0 40 80 160 312; do echo -n "head $h: "; OEM_USE_TESTING_LIB=1 OEM_CELLS_HEAD=$h python scripts/cells_bench.py 625 50000 60000 2>&1 | grep "batched" | tail -1; done; done
