#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MCN_FWD_PIPE=1 timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "filter_acts" > gpurun_out/t_y.log 2>&1 || { tail -30 gpurun_out/t_y.log; exit 1; }
tail -2 gpurun_out/t_y.log
for p in 0 1; do
echo "== MCN_FWD_PIPE=$p"
MCN_FWD_PIPE=$p timeout -k 10 120 python tools/kbench.py conv filterActs || exit 1
done
for w in deepid2 deepid2plus; do for p in 0 1; do
echo "== step $w PIPE=$p"
MCN_FWD_PIPE=$p timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done; done
