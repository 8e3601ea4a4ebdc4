#!/bin/bash
# kernel-trace profile of the forced data-parallel benchmark loop (one rank) -> gpurun_out/profdp/*.csv
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profdp
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profdp -o bench -- python3 $R/bench.py --force-dp --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $R/gpurun_out/profdp.log 2>&1
rc=$?; tail -3 $R/gpurun_out/profdp.log; exit $rc
