#!/bin/bash
# rocprofv3 passes of bench.py for profiles/r04_* (run on the GPU box through gpurun):
#   kernel trace + stats of the driver's command (--steps 20 --warmup 5) and of the default run (2000 steps), then
#   FETCH_SIZE and WRITE_SIZE of BOTH commands in passes of their own (MI355X_MICROARCH.md, section HBM; PMC passes carry
#   no trace domain other than the kernel trace), the tile engine's phase stamps of a 20-iteration call and the chip-wide timeline of an iteration (tools/stamps_raw.py).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d $O/trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace20.json 2> $O/trace20.err && echo trace20 done &&
$P --kernel-trace --stats -d $O/trace2000 -- python3 $R/bench.py --no-cpu-baseline > $O/trace2000.json 2> $O/trace2000.err && echo trace2000 done &&
$P --kernel-trace --pmc FETCH_SIZE -d $O/fetch20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-converge --no-other-configs > $O/fetch20.json 2> $O/fetch20.err && echo fetch20 done &&
$P --kernel-trace --pmc WRITE_SIZE -d $O/write20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-converge --no-other-configs > $O/write20.json 2> $O/write20.err && echo write20 done &&
$P --kernel-trace --pmc FETCH_SIZE -d $O/fetch2000 -- python3 $R/bench.py --no-cpu-baseline --no-converge --no-other-configs > $O/fetch2000.json 2> $O/fetch2000.err && echo fetch2000 done &&
$P --kernel-trace --pmc WRITE_SIZE -d $O/write2000 -- python3 $R/bench.py --no-cpu-baseline --no-converge --no-other-configs > $O/write2000.json 2> $O/write2000.err && echo write2000 done &&
python3 $R/tools/call_stamps.py 4096 20 > $O/call_stamps_20.txt 2>&1 && echo stamps done &&
python3 $R/tools/stamps_raw.py 4096 40 > $O/stamps_timeline.txt 2>&1 && echo timeline done
find $O -name "*.csv" | wc -l
du -sh $O
