#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_net_layers_gpu.py tests/test_net_gpu.py tests/test_weight_sharing_gpu.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -12
