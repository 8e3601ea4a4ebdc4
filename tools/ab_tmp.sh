#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gates_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
timeout -k 10 300 python -u -m pytest tests/test_net_layers_gpu.py -x -q --timeout 120 --timeout-method thread -k "l2sn_gradient_added" 2>&1 | tail -15 || exit 1
for w in deepid2 deepid2plus; do
  timeout -k 10 120 python bench.py --workload $w --steps 300 --warmup 30 --no-profile --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' || exit 1
  MCN_NO_L2SN_FOLD=1 timeout -k 10 120 python bench.py --workload $w --steps 300 --warmup 30 --no-profile --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' || exit 1
done
