#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_net_layers_gpu.py -x -q --timeout 300 --timeout-method thread -k "early_update" 2>&1 | tail -8
