#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "weight_acts or filter_acts_shapes or fused_update" > gpurun_out/t_p.log 2>&1 || { tail -30 gpurun_out/t_p.log; exit 1; }
tail -3 gpurun_out/t_p.log
echo "== old"; MCN_WA_WIDE=0 timeout -k 10 120 python tools/kbench.py + weightActs || exit 1
for b in 192 288 384 576 864; do
echo "== wide, blocks $b"; MCN_WA_WIDE_BLOCKS=$b timeout -k 10 120 python tools/kbench.py + weightActs || exit 1
done
for cfg in "0 384" "1 192" "1 288" "1 384" "1 576"; do set -- $cfg
echo "== step, WIDE=$1 BLOCKS=$2"
MCN_WA_WIDE=$1 MCN_WA_WIDE_BLOCKS=$2 timeout -k 10 200 python -u bench.py --workload deepid2plus --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done
