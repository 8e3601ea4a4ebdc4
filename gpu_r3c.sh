#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_net_layers_gpu.py tests/test_net_gpu.py -k "fused or train_3_steps or sgd or ragged or add_sum or two_streams or graph_replay" -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_r3c.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_r3c.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3c_$i.json 2> gpurun_out/bench_r3c.err || exit 1
MCN_NO_FUSED_TAIL=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3c_nofuse_$i.json 2>> gpurun_out/bench_r3c.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r3c_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])
PY
