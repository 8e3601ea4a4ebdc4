#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_dp_two_ranks_gpu.py tests/test_net_gpu.py tests/test_weight_sharing_gpu.py tests/test_gates_gpu.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -8 || exit 1
timeout -k 10 300 python -u bench.py --force-dp --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced dp', d['ms_per_step'], d['rccl'])"
