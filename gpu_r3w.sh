#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
./gpu_trace.sh --force-dp > /dev/null 2>&1
python3 - <<'PY'
import csv,statistics
rows=sorted(csv.DictReader(open('gpurun_out/prof/bench_kernel_trace.csv')),key=lambda r:int(r['Start_Timestamp']))
# step delimiter: k_sgd_planned (any queue), take last in each cluster
idx=[i for i,r in enumerate(rows) if 'k_sgd' in r['Kernel_Name']]
print(len(idx),'sgd launches; names:',set(r['Kernel_Name'][:40] for r in rows if 'sgd' in r['Kernel_Name']))
a,b=idx[60],idx[61]
t0=int(rows[a]['End_Timestamp'])
for r in rows[a+1:b+1]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print('%7.1f %7.1f %6.1f q%s %s'%(s,e,e-s,r['Queue_Id'],r['Kernel_Name'][:80]))
PY
