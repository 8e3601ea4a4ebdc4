#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -8 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -3 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --force-dp --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_forced_dp.json 2> gpurun_out/bench_forced_dp.err
rc=$?; tail -c 900 gpurun_out/bench_forced_dp.json; tail -3 gpurun_out/bench_forced_dp.err; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
rc=$?; tail -c 600 gpurun_out/bench_plain.json; exit $rc
