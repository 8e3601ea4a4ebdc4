#!/bin/bash
# round 3, first pass: new weight-sharing tests + failed-step test, baseline bench
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_weight_sharing_gpu.py tests/test_net_gpu.py -k "shared or failed_step" -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_r3a.log 2>&1
rc=$?; tail -25 gpurun_out/pytest_r3a.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err
rc=$?; tail -c 1500 gpurun_out/bench_r3a.json; tail -5 gpurun_out/bench_r3a.err; exit $rc
