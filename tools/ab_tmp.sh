#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "img_acts" 2>&1 | tail -2 || exit 1
for e in 0 1 3; do
echo "== MCN_IA_XCD_PAIR=$e"
MCN_IA_XCD_PAIR=$e timeout -k 10 120 python tools/kbench.py "" "imgActs+maxUndo" | grep -v "+ " || exit 1
done
for w in deepid2 deepid2plus; do for e in 0 1 3 0 1 3; do
echo "== step $w IA_XCD_PAIR=$e"
MCN_IA_XCD_PAIR=$e timeout -k 10 200 python -u bench.py --workload $w --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done; done
