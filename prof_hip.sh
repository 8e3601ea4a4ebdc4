#!/bin/bash
# HIP API trace + kernel trace of the benchmark loop -> gpurun_out/hipprof/*.csv (host calls beside the kernels they launch)
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hipprof
cd /tmp && timeout -k 10 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $R/gpurun_out/hipprof -o bench -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-profile > $R/gpurun_out/hipprof.log 2>&1
rc=$?; tail -3 $R/gpurun_out/hipprof.log; ls -la $R/gpurun_out/hipprof | head; exit $rc
