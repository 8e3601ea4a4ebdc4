#!/bin/bash
# kernel-trace profile of the benchmark loop -> gpurun_out/prof/*.csv
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $R/gpurun_out/prof.log 2>&1
rc=$?; tail -3 $R/gpurun_out/prof.log; ls -R $R/gpurun_out/prof | head; exit $rc
