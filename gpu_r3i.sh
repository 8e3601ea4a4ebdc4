#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gates_gpu.py tests/test_net_layers_gpu.py tests/test_net_gpu.py tests/test_weight_sharing_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_r3i.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_r3i.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data > gpurun_out/bench_r3i_$i.json 2> gpurun_out/bench_r3i.err || exit 1
MCN_NO_FORK_GATES=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3i_nofg_$i.json 2>> gpurun_out/bench_r3i.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r3i_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['host_enqueue_ms_per_step'], d.get('host_breakdown_us_per_step'))
PY
./gpu_trace.sh > /dev/null 2>&1
python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv median
