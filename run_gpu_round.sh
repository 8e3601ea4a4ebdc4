#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, kernel-trace profile.  Every step has its own time limit.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -5 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; tail -c 3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err; [ $rc -ne 0 ] && exit $rc
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
rc=$?; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof.log; ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20
exit $rc
