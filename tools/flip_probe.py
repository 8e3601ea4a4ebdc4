#!/usr/bin/env python3
"""Is a difference between two forward tilings a rounding difference or a relu decision that flipped?  One DeepID2+ training step
with MCN_FA_BIG_LOCAL=1 and =0 (child processes), local3's weight increments compared entry by entry: where do the entries
beyond 1e-4 x max sit?  (local3 weights: rows = (module, k), columns = filters.)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

if len(sys.argv) > 1:
    import ccn_loader
    ccn_loader.load()
    from test_net_gpu import Native
    from test_net_layers_gpu import preset
    dp, L = preset('deepid2plus', 128)
    data = dp.get_next_batch()[2]
    with Native(L) as lib:
        lib.startBatch(data, 0)
        lib.finishBatch()
        lib.syncWithHost()
    l3 = [l for l in L if l['name'] == 'local3'][0]
    np.save(sys.argv[1], l3['weightsInc'][0])
    sys.exit(0)

out = []
for v in ('1', '0'):
    path = '/tmp/flip_%s.npy' % v
    env = dict(os.environ, MCN_FA_BIG_LOCAL=v)
    subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env=env, timeout=300)
    out.append(np.load(path))
a, b = out
d = np.abs(a.astype(np.float64) - b)
mx = np.abs(b).max()
bad = np.argwhere(d > 1e-4 * mx)
print('max |inc| %.3e, max diff %.3e, entries beyond 1e-4 max: %d of %d' % (mx, d.max(), len(bad), d.size))
K = a.shape[0] // 100
mods = sorted(set((int(r) // K, int(c)) for r, c in bad))
print('they sit in %d (module, filter) columns: %s' % (len(mods), mods[:10]))
print('median diff elsewhere: %.3e' % np.median(d))
