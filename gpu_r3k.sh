#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_net_gpu.py tests/test_dp_two_ranks_gpu.py tests/test_gates_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_r3k.log 2>&1
rc=$?; tail -8 gpurun_out/pytest_r3k.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2; do
MCN_DEBUG_TIMES=1 timeout -k 10 300 python -u bench.py --force-dp --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3k_dp_$i.json 2> gpurun_out/bench_r3k_dp.err || { tail -5 gpurun_out/bench_r3k_dp.err; exit 1; }
MCN_NO_COMM_GATES=1 timeout -k 10 300 python -u bench.py --force-dp --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3k_dpnogate_$i.json 2>> gpurun_out/bench_r3k_dp.err || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3k_plain_$i.json 2>> gpurun_out/bench_r3k_dp.err || exit 1
done
grep "\[mcn\]" gpurun_out/bench_r3k_dp.err | head -5
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r3k_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['host_enqueue_ms_per_step'], d.get('rccl'))
PY
./gpu_trace.sh --force-dp > /dev/null 2>&1
python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv median
