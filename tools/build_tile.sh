#!/bin/bash
# Dev tool: compile gl_tile.hip alone with the register / scratch report (into build/report/: the library's own objects are not touched).  usage: tools/build_tile.sh [-DGL_TILE_...=..]
cd "$(dirname "$0")/../graph_learning_amd/csrc" && mkdir -p build/report &&
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include -I. -Wall \
  -Wno-unused-function "$@" -c gl_tile.hip -o build/report/gl_tile.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2> build/report/gl_tile.usage.txt
grep -E "error|warning" build/report/gl_tile.usage.txt | head -20
python ../../tools/usage_report.py build/report/gl_tile.usage.txt | grep k_tile
