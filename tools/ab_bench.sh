#!/bin/bash
# A/B of environment switches inside ONE box (boxes differ by several per cent): tools/ab_bench.sh "VAR=1" ["VAR2=1" ...]
# runs bench.py (no CPU baseline, no profile pass) alternately without and with each switch, 3 rounds.
set -o pipefail
run() { env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('%.4f ms/step  host %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for round in 1 2 3; do
  echo -n "base        : "; run X=1 || exit 1
  for sw in "$@"; do echo -n "$sw : "; run $sw || exit 1; done
done
