#!/bin/bash
# rocprofv3 passes of bench.py for profiles/r02_* (run on the GPU box through gpurun):
#   kernel trace + stats of the driver's command (--steps 20 --warmup 5) and of the default run (2000 steps),
#   then FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, section HBM; PMC passes carry no
#   trace domain other than the kernel trace).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d $O/trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace20.json 2> $O/trace20.err && echo trace20 done &&
$P --kernel-trace --stats -d $O/trace2000 -- python3 $R/bench.py --no-cpu-baseline > $O/trace2000.json 2> $O/trace2000.err && echo trace2000 done &&
$P --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --no-cpu-baseline --no-converge > $O/fetch.json 2> $O/fetch.err && echo fetch done &&
$P --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --no-cpu-baseline --no-converge > $O/write.json 2> $O/write.err && echo write done
find $O -name "*.csv" | head -40
du -sh $O
