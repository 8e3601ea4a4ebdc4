#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tr
grep -m1 "model name" /proc/cpuinfo
for i in 1 2 3; do
rm -rf /tmp/trp; mkdir -p /tmp/trp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-host-data --min-seconds 0 "$@" > $R/gpurun_out/tr/prof$i.log 2>&1) || { tail -5 $R/gpurun_out/tr/prof$i.log; exit 1; }
T=$(find /tmp/trp -name 'bench_kernel_trace.csv' | head -1); S=$(find /tmp/trp -name 'bench_kernel_stats.csv' | head -1)
cp "$T" $R/gpurun_out/tr/trace$i.csv; cp "$S" $R/gpurun_out/tr/stats$i.csv
python3 $R/tools/step_timeline.py $R/gpurun_out/tr/trace$i.csv median > $R/gpurun_out/tr/timeline$i.txt
grep "step length\|both queues\|share the" $R/gpurun_out/tr/timeline$i.txt | cut -c1-200
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/tr/prof$i.log
done
