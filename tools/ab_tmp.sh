#!/bin/bash
set -o pipefail
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
