#!/bin/bash
# The evidence of a round in one GPU-box pass -> gpurun_out/final/: bench line (default run), rocprofv3 kernel stats + trace of
# the same command, step timeline, phases, per-kernel table (kbench), DeepID2+ / forced data-parallel / host-data bench lines.
# Every step has its own time limit; the script stops at the first failure.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
if [ "$1" != "rest" ]; then
timeout -k 10 400 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-host-data --min-seconds 0 > $O/prof.log 2>&1) || { tail -5 $O/prof.log; exit 1; }
echo "rocprof done"
T=$(find $O/prof -name 'bench_kernel_trace.csv' | head -1)
S=$(find $O/prof -name 'bench_kernel_stats.csv' | head -1)
cp "$S" $O/kernel_stats.csv && python tools/step_timeline.py "$T" median > $O/step_timeline.txt || exit 1
rm -rf $O/prof
timeout -k 10 200 python tools/phases.py > $O/phases.txt 2>&1 || { tail -5 $O/phases.txt; exit 1; }
echo "phases done"
fi
timeout -k 10 400 python tools/kbench.py > $O/kbench.txt 2>&1 || { tail -5 $O/kbench.txt; exit 1; }
echo "kbench done"
timeout -k 10 300 python -u bench.py --workload deepid2plus --no-cpu-baseline > $O/bench_deepid2plus.json 2>> $O/bench.err || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profp -o bench -- python3 $R/bench.py --workload deepid2plus --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-host-data --min-seconds 0 > $O/profp.log 2>&1) || { tail -5 $O/profp.log; exit 1; }
T=$(find $O/profp -name 'bench_kernel_trace.csv' | head -1)
S=$(find $O/profp -name 'bench_kernel_stats.csv' | head -1)
cp "$S" $O/kernel_stats_deepid2plus.csv && python tools/step_timeline.py "$T" median > $O/step_timeline_deepid2plus.txt || exit 1
rm -rf $O/profp
timeout -k 10 300 python -u bench.py --force-dp --no-cpu-baseline > $O/bench_forced_dp.json 2>> $O/bench.err || exit 1
echo "all done"
