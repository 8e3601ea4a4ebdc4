#!/bin/bash
# A/B of MCN_SIDE_UNITS (deferred gradient pieces collected before they go to the side stream) in one box
for u in 1 2 3 4 6; do
echo "== MCN_SIDE_UNITS=$u"
MCN_SIDE_UNITS=$u timeout -k 10 200 python -u bench.py --workload ${1:-deepid2} --no-cpu-baseline --no-host-data --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'])" || exit 1
done
