#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for w in -1 2 4; do
echo "== step MCN_IA_WIDE=$w"
if [ $w -lt 0 ]; then unset MCN_IA_WIDE; else export MCN_IA_WIDE=$w; fi
timeout -k 10 200 python -u bench.py --workload deepid2plus --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done
unset MCN_IA_WIDE
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepid2', d['ms_per_step'], d['ms_per_step_min'])" || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_all.log 2>&1; rc=$?; tail -5 gpurun_out/t_all.log; exit $rc
