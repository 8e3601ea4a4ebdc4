#!/bin/bash
# usage: pmc_kbench.sh <layer> <op> "<counters>"   -> gpurun_out/pmc/*.csv
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o pmc -- python3 $R/tools/kbench.py "$1" "$2" > $R/gpurun_out/pmc.log 2>&1
rc=$?; tail -3 $R/gpurun_out/pmc.log; ls $R/gpurun_out/pmc; exit $rc
