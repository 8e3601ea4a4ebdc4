#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-host-data > gpurun_out/bench_r3b.json 2> gpurun_out/bench_r3b.err
rc=$?; tail -c 600 gpurun_out/bench_r3b.json; tail -5 gpurun_out/bench_r3b.err; [ $rc -ne 0 ] && exit $rc
MCN_NO_GRAD_STREAM=1 timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-host-data > gpurun_out/bench_r3b_1s.json 2> gpurun_out/bench_r3b_1s.err
rc=$?; tail -c 600 gpurun_out/bench_r3b_1s.json; [ $rc -ne 0 ] && exit $rc
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-host-data --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
rc=$?; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof.log; exit $rc
