#!/bin/bash
# forced data-parallel step (one rank) under environment switches, one box: ab_dp.sh "A=1" ...
run() {
  env $1 timeout -k 10 200 python bench.py --force-dp --no-cpu-baseline --no-profile 2>/dev/null > /tmp/bd.json || exit 1
  python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/bd.json') if l.startswith('{')][-1])
print('%-40s %.4f ms/step  %.0f images/s' % (sys.argv[1], d['ms_per_step'], d['value']))
PY
}
for r in 1 2 3; do
  run "MCN_X=0"
  for e in "$@"; do run "$e"; done
done
