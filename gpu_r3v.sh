#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_r3v.json 2> gpurun_out/bench_r3v.err || { tail -5 gpurun_out/bench_r3v.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r3v.json'))
print({k:v for k,v in d.items() if k not in ('kernels','config','host_breakdown_us_per_step')})
for e in d['kernels']: print('%-28s %6.2f min %6.2f  %s %.3f'%(e['op'],e['avg_us'],e['min_us'],e['bound'],e['frac']))
PY
timeout -k 10 120 python tools/kbench.py "" "" | grep -v "+" 
