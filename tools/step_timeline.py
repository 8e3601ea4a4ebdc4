#!/usr/bin/env python3
"""One training step's kernels per hardware queue from a rocprofv3 kernel trace (gpurun_out/prof/bench_kernel_trace.csv):
start / end / duration in microseconds relative to the end of the previous step's weight update."""
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof/bench_kernel_trace.csv'
step = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_sgd_planned' in r['Kernel_Name']]
a, b = idx[step], idx[step + 1]
t0 = int(rows[a]['End_Timestamp'])
print('step length %.1f us' % ((int(rows[b]['End_Timestamp']) - t0) / 1e3))
busy = {}
for r in rows[a + 1:b + 1]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    busy[r['Queue_Id']] = busy.get(r['Queue_Id'], 0) + e - s
    print('%7.1f %7.1f %6.1f  q%s  %s' % (s, e, e - s, r['Queue_Id'], r['Kernel_Name'][:70]))
print('busy per queue (us):', {k: round(v, 1) for k, v in busy.items()})
