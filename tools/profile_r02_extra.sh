#!/bin/bash
# rocprofv3 passes for the kernels beside the headline (run on the GPU box through gpurun), outputs under
# gpurun_out/prof_r02x/: the tile engine at n = 1024, 2048 and 8192 (kernel trace + stats; FETCH_SIZE / WRITE_SIZE at
# 8192 in passes of their own), the LGMRF node updates at n = 2048, GraphDictLog K=8 n=1024 T=4096 in fp64 and fp32.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r02x
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P="rocprofv3 --output-format csv"
for N in 1024 2048 8192; do
  $P --kernel-trace --stats -d $O/tile$N -- python3 $R/tools/iter_bench.py $N 500 > $O/tile$N.log 2>&1 && echo tile$N done || exit 1
done
$P --kernel-trace --pmc FETCH_SIZE -d $O/fetch8192 -- python3 $R/tools/iter_bench.py 8192 500 > $O/fetch8192.log 2>&1 && echo fetch done &&
$P --kernel-trace --pmc WRITE_SIZE -d $O/write8192 -- python3 $R/tools/iter_bench.py 8192 500 > $O/write8192.log 2>&1 && echo write done &&
GRAPHLEARN_AMD_LGMRF_MAX_NODES=256 $P --kernel-trace --stats -d $O/lgmrf -- python3 $R/tools/lgmrf_bench.py 2048 512 1 > $O/lgmrf.log 2>&1 && echo lgmrf done &&
GD_BITS=64 $P --kernel-trace --stats -d $O/gd64 -- python3 $R/tools/graphdict_bench.py > $O/gd64.log 2>&1 && echo gd64 done &&
GD_BITS=32 $P --kernel-trace --stats -d $O/gd32 -- python3 $R/tools/graphdict_bench.py > $O/gd32.log 2>&1 && echo gd32 done
du -sh $O
