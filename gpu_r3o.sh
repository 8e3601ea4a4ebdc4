#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
./gpu_trace.sh --workload deepid2plus > /dev/null 2>&1
python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv median
