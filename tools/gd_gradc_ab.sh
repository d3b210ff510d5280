#!/bin/bash
# Dev tool: k_gd_grad_c's columns of x_j in flight (GRAPHLEARN_AMD_GD_XD = 2 | 4) side by side on one GPU, fp32 and fp64 form
# of BASELINE configs[4]: step time and the two gradient kernels between HIP events.  Further arguments: names of variant
# libraries (tools/build_variant_gd.sh NAME -DGD_GC_EXP=1..3: timing-only ablations of the kernel's loads).
V=graph_learning_amd/csrc/build/var
for bits in 32 64; do
  for xd in 2 4 2 4; do
    echo "== bits $bits xd $xd"
    GD_BITS=$bits GD_STEPS=20 GRAPHLEARN_AMD_GD_XD=$xd python tools/graphdict_bench.py > /tmp/gd_one.log 2>&1 || { tail -3 /tmp/gd_one.log; exit 1; }
    grep "^fp" /tmp/gd_one.log | cut -c1-110
  done
  for lib in "$@"; do
    echo "== bits $bits $lib"
    GD_BITS=$bits GD_STEPS=20 GRAPHLEARN_AMD_LIB=$V/$lib.so python tools/graphdict_bench.py > /tmp/gd_one.log 2>&1 || { tail -3 /tmp/gd_one.log; exit 1; }
    grep "^fp" /tmp/gd_one.log | cut -c1-110
  done
done
