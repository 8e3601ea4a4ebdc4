#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "img_acts" > gpurun_out/t_m.log 2>&1 || { tail -30 gpurun_out/t_m.log; exit 1; }
tail -3 gpurun_out/t_m.log
for cfg in "0 1" "4 1" "2 1" "4 2" "2 2"; do set -- $cfg
echo "== narrow form, MCN_IA_CHAIN=$1 KS=$2"
MCN_IA_WIDE=0 MCN_IA_CHAIN=$1 MCN_IA_CHAIN_KS=$2 timeout -k 10 120 python tools/kbench.py + imgActs || exit 1
done
echo "== wide form"; timeout -k 10 120 python tools/kbench.py + imgActs || exit 1
for cfg in "0 1" "4 1" "2 1" "4 2" "2 2"; do set -- $cfg
echo "== step, MCN_IA_CHAIN=$1 KS=$2"
MCN_IA_CHAIN=$1 MCN_IA_CHAIN_KS=$2 timeout -k 10 200 python -u bench.py --workload deepid2plus --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done
