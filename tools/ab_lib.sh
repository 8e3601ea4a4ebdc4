#!/bin/bash
# A/B of two builds of libmiconvnet.so on the whole step inside ONE box: tools/ab_lib.sh <other.so> [bench.py arguments]
set -o pipefail
L=cuda-convnet_amd/libmiconvnet.so
other=$1; shift
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('%.4f ms/step  host %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
cp $L /tmp/cur.so
for round in 1 2 3; do
  echo -n "current : "; run "$@" || exit 1
  cp "$other" $L
  echo -n "other   : "; run "$@"; rc=$?
  cp /tmp/cur.so $L
  [ $rc -ne 0 ] && exit $rc
done
