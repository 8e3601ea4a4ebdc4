#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py --workload deepid2plus --no-cpu-baseline --no-host-data > gpurun_out/bench_r3l_plus.json 2> gpurun_out/bench_r3l.err || { tail -5 gpurun_out/bench_r3l.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r3l_plus.json'))
print({k:v for k,v in d.items() if k not in ('kernels','config')})
for e in d['kernels']: print('%-28s %6.2f min %6.2f  %s %.3f'%(e['op'],e['avg_us'],e['min_us'],e['bound'],e['frac']))
PY
./gpu_trace.sh --workload deepid2plus > /dev/null 2>&1
python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv median
