#!/bin/bash
# Dev tool: steady-state us/iteration of the tile engine (n = 4096, 2000 iterations) against the sleep in front of the
# first probe of the consumers' / owners' polls (GRAPHLEARN_AMD_TILE_ACQ_SLEEP / _OWN_SLEEP, units of 128 clocks).
# usage: bash tools/sleep_sweep.sh "0 4 8 12 16 20 24" "0 4 8 12"
cd "$(dirname "$0")/.."
for o in ${2:-0}; do
  for a in ${1:-0}; do
    r=$(GRAPHLEARN_AMD_TILE_ACQ_SLEEP=$a GRAPHLEARN_AMD_TILE_OWN_SLEEP=$o python tools/iter_bench.py 4096 2000 | sed -e 's/.*: gpu //' -e 's/ wall.*//')
    echo "acq_sleep $a own_sleep $o: $r"
  done
done
