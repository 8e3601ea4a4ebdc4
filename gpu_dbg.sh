#!/bin/bash
export TMPDIR=/tmp
python - <<'PY'
import os, sys
sys.path.insert(0,'tests')
import numpy as np
import test_dp_two_ranks_gpu as T
import pathlib, tempfile
def maxrel(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b))) / max(float(np.max(np.abs(b))), 1e-30)
keys = None
for steps in (1, 2):
    runs = []
    for rep in range(3):
        tmp = pathlib.Path(tempfile.mkdtemp())
        runs.append(T.run_ranks('deepid2', steps, tmp, 4)[0])
    keys = [k for k in runs[0].files if '.inc' in k]
    print('steps', steps, 'DP run-to-run max rel diff:', {k: max(maxrel(runs[0][k], r[k]) for r in runs[1:]) for k in keys})
    print('  costs', [dict((k, r[k].tolist()) for k in r.files if k.startswith('cost.')) for r in runs])
    Ls = [T.single_gpu('deepid2', steps, 4) for _ in range(2)]
    print('  single costs', [c for _, c, _ in Ls])
    for l0, l1 in zip(Ls[0][0], Ls[1][0]):
        if 'weights' in l0:
            print('  single run-to-run', l0['name'], maxrel(l0['weightsInc'][0], l1['weightsInc'][0]))
    for l in Ls[0][0]:
        if 'weights' in l:
            print('  DP vs single', l['name'], maxrel(runs[0]['%s.inc0' % l['name']], l['weightsInc'][0]))
PY
