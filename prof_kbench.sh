#!/bin/bash
# usage: prof_kbench.sh <layer> <op>   kernel-trace stats of one kbench selection -> gpurun_out/kprof/*.csv
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/kprof
cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kprof -o k -- python3 $R/tools/kbench.py "$1" "$2" > $R/gpurun_out/kprof.log 2>&1
rc=$?; cut -d, -f1-4 $R/gpurun_out/kprof/k_kernel_stats.csv | cut -c1-110; exit $rc
