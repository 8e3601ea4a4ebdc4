#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_all.log 2>&1; rc=$?; tail -5 gpurun_out/t_all.log; exit $rc
