#!/bin/bash
# Dev tool: a variant of the library with other compile-time switches of gl_lgmrf.hip (-DLG_...=...), beside the shipped one.
#   tools/build_variant_lgmrf.sh NAME -DLG_HIST_LDS=0   ->  graph_learning_amd/csrc/build/var/NAME.so   (use with GRAPHLEARN_AMD_LIB)
set -e
name=$1; shift
cd "$(dirname "$0")/../graph_learning_amd/csrc"
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include -I. -Wall \
  -Wno-unused-function "$@" -c gl_lgmrf.hip -o build/var/$name.gl_lgmrf.o
objs=$(ls build/*.o | grep -v gl_lgmrf.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,libgraphlearn_amd.so -ldl $objs build/var/$name.gl_lgmrf.o -o build/var/$name.so
echo build/var/$name.so
