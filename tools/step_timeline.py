#!/usr/bin/env python3
"""One training step's kernels per hardware queue from a rocprofv3 kernel trace (gpurun_out/prof/bench_kernel_trace.csv):
start / end / duration in microseconds relative to the end of the previous step's weight update.

    step_timeline.py trace.csv 60        one step (the 60th)
    step_timeline.py trace.csv median    every kernel's MEDIAN start / end over all steps with the usual launch sequence
                                         (a single step can hold a host hiccup; the median is what the bench times),
                                         plus the intervals in which BOTH queues are idle"""
import csv
import statistics
import sys

path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof/bench_kernel_trace.csv'
mode = sys.argv[2] if len(sys.argv) > 2 else '60'
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
# a step ends with the update launch on the MAIN queue (the one that runs the forward pass; the early update of the other
# weight layers runs on the side queue and may end after it)
queues = {}
for r in rows:
    queues[r['Queue_Id']] = queues.get(r['Queue_Id'], 0) + (0 if 'k_stream_' in r['Kernel_Name'] else 1)
main_q = max(queues, key=queues.get)
idx = [i for i, r in enumerate(rows) if 'k_sgd_planned' in r['Kernel_Name'] and r['Queue_Id'] == main_q]


def step_rows(k):
    a, b = idx[k], idx[k + 1]
    t0 = int(rows[a]['End_Timestamp'])
    return [((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r['Queue_Id'], r['Kernel_Name']) for r in rows[a + 1:b + 1]]


def show(items, length):
    print('step length %.1f us' % length)
    busy = {}
    for s, e, q, name in items:
        busy[q] = busy.get(q, 0) + e - s
        print('%7.1f %7.1f %6.1f  q%s  %s' % (s, e, e - s, q, name[:70]))
    print('busy per queue (us):', {k: round(v, 1) for k, v in busy.items()})
    # both-queues-idle intervals
    ev = sorted((s, e) for s, e, _, _ in items)
    cur, gaps = 0.0, []
    for s, e in ev:
        if s > cur + 0.05:
            gaps.append((cur, s))
        cur = max(cur, e)
    print('both queues idle:', ', '.join('%.1f-%.1f (%.1f us)' % (a, b, b - a) for a, b in gaps if b - a >= 1.0) or 'never for 1 us or more')


if mode != 'median':
    it = step_rows(int(mode))
    show(it, it[-1][1])
else:
    steps = [step_rows(k) for k in range(5, len(idx) - 1)]
    # keep the steps with the most common sequence of (queue-agnostic) kernel names in per-queue order
    def sig(it):
        return tuple(sorted((q, tuple(n for _, _, qq, n in it if qq == q)) for q in {x[2] for x in it}))
    sigs = {}
    for it in steps:
        sigs.setdefault(sig(it), []).append(it)
    common = max(sigs.values(), key=len)
    print('%d of %d steps share the launch sequence' % (len(common), len(steps)))
    # order kernels per queue, then merge by median start
    per_q = {}
    for it in common:
        cnt = {}
        for s, e, q, n in it:
            k = (q, cnt.get(q, 0))
            cnt[q] = cnt.get(q, 0) + 1
            per_q.setdefault(k, []).append((s, e, n))
    items = []
    for (q, i), v in per_q.items():
        items.append((statistics.median(x[0] for x in v), statistics.median(x[1] for x in v), q, v[0][2]))
    items.sort()
    show(items, statistics.median(it[-1][1] for it in common))
