#!/bin/bash
# A/B of ONE environment switch with extra bench.py arguments, inside one box: tools/ab_env.sh VAR=1 [bench.py arguments]
set -o pipefail
sw=$1; shift
run() { env "$1" timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 300 --warmup 30 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('%.4f ms/step  host %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for round in 1 2 3; do
  echo -n "base : "; run X=1 "$@" || exit 1
  echo -n "$sw : "; run "$sw" "$@" || exit 1
done
