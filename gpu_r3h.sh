#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3h_$i.json 2> gpurun_out/bench_r3h.err || exit 1
MCN_TICK_KERNEL=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-host-data --no-profile > gpurun_out/bench_r3h_tickkernel_$i.json 2>> gpurun_out/bench_r3h.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r3h_*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['host_enqueue_ms_per_step'])
PY
./gpu_trace.sh > /dev/null 2>&1
python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv median
