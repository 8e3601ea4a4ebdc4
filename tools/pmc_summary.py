#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 counter_collection CSVs: tools/pmc_summary.py file.csv ..."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print('    %-28s %14.1f  (%d dispatches)' % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
