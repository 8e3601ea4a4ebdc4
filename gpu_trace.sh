#!/bin/bash
# kernel trace of the bench loop -> gpurun_out/prof/, timeline of one step printed
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-host-data --min-seconds 0 "$@" > $R/gpurun_out/prof.log 2>&1
rc=$?; tail -3 $R/gpurun_out/prof.log; [ $rc -ne 0 ] && exit $rc
cd $R && python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv 60 | tee gpurun_out/timeline.txt
