#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "filter_acts" > gpurun_out/t_q.log 2>&1 || { tail -30 gpurun_out/t_q.log; exit 1; }
tail -3 gpurun_out/t_q.log
timeout -k 10 120 python tools/kbench.py local filterActs || exit 1
timeout -k 10 120 python tools/kbench.py conv filterActs || exit 1
