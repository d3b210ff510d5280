#!/bin/bash
# Run on the GPU box (through gpurun): six rocprofv3 passes of bench.py, outputs under gpurun_out/prof/.
# PMC passes are separate from each other and carry no trace domain other than the kernel trace.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-converge"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/res_trace -- $B > $O/res_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/res_fetch -- $B > $O/res_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/res_write -- $B > $O/res_write.log 2>&1
export GRAPHLEARN_AMD_ENGINE=stream GRAPHLEARN_AMD_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/str_trace -- $B > $O/str_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/str_fetch -- $B > $O/str_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/str_write -- $B > $O/str_write.log 2>&1
find $O -name "*.csv" | head -40
du -sh $O
