#!/bin/bash
# Dev tool: a variant of the library with other compile-time switches of gl_graphdict.hip, beside the shipped one
# (e.g. -DGD_UPD_EXP=1..5: k_gd_update2 without its coefficient part / without its weight part / without the fp32
# mirror's stores / without the gradient slabs' loads / without any store of the weights -- timing only, wrong results).
#   tools/build_variant_gd.sh NAME -DGD_UPD_EXP=1      ->  graph_learning_amd/csrc/build/var/NAME.so
# Use with GRAPHLEARN_AMD_LIB=<that path>.
set -e
name=$1; shift
cd "$(dirname "$0")/../graph_learning_amd/csrc"
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include -I. -Wall \
  -Wno-unused-function "$@" -c gl_graphdict.hip -o build/var/$name.gl_graphdict.o
objs=$(ls build/*.o | grep -v "build/gl_graphdict.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,libgraphlearn_amd.so -ldl $objs build/var/$name.gl_graphdict.o -o build/var/$name.so
echo build/var/$name.so
