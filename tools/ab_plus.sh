#!/bin/bash
# DeepID2+ step time under environment switches (one box): ab_plus.sh "A=1" "B=2 C=3" ...
run() {
  env $1 timeout -k 10 200 python bench.py --workload deepid2plus --no-cpu-baseline --no-profile --steps 100 2>/dev/null > /tmp/bp.json || exit 1
  python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/bp.json') if l.startswith('{')][-1])
print('%-40s %.4f ms/step  %.0f images/s' % (sys.argv[1], d['ms_per_step'], d['value']))
PY
}
for r in 1 2; do
  run "MCN_X=0"
  for e in "$@"; do run "$e"; done
done
