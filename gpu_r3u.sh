#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_net_gpu.py tests/test_net_layers_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_u.log 2>&1 || { tail -30 gpurun_out/t_u.log; exit 1; }
tail -2 gpurun_out/t_u.log
for w in deepid2plus deepid2; do for e in 1 0; do
echo "== $w early=$e"
if [ $e = 0 ]; then export MCN_NO_EARLY_SGD=1; else unset MCN_NO_EARLY_SGD; fi
timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done; done
