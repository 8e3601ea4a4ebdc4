#!/bin/bash
# kbench selection for two builds of libmiconvnet.so in one box: ab_kbench.sh <other.so> <layer> <op>
L=cuda-convnet_amd/libmiconvnet.so
for r in 1 2; do
  echo "-- current"; timeout -k 10 200 python tools/kbench.py "$2" "$3" || exit 1
  cp $L /tmp/cur.so && cp "$1" $L
  echo "-- $1"; timeout -k 10 200 python tools/kbench.py "$2" "$3"; rc=$?
  cp /tmp/cur.so $L
  [ $rc -ne 0 ] && exit $rc
done
