#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_net_layers_gpu.py tests/test_net_gpu.py tests/test_weight_sharing_gpu.py -k "partials or fused or net_layers or train_3 or ragged or failed or shared or checkpoint or data_parallel" -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_r3g.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_r3g.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3g_$i.json 2> gpurun_out/bench_r3g.err || exit 1
MCN_NO_BIAS_IN_WGRAD=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3g_nobias_$i.json 2>> gpurun_out/bench_r3g.err || exit 1
MCN_SIDE_COST=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3g_sidecost_$i.json 2>> gpurun_out/bench_r3g.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r3g_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['host_enqueue_ms_per_step'])
PY
./gpu_trace.sh > /dev/null 2>&1
python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv median | tail -16
