#!/usr/bin/env python3
"""HIP-event time of the small cost / reduction kernels at the DeepID2 shapes (N = 128)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from kbench import timeit, capi, call, dev, dev_empty, ws_bytes  # noqa: E402
from kutil import dev_double  # noqa: E402
N, D, K = 128, 160, 10575
rng = np.random.default_rng(0)
f = dev((rng.standard_normal((D, N)) * 0.3).astype(np.float32))
ids = rng.integers(0, K, (1, N)); ids[0, 64:96] = ids[0, :32]
ids = dev(ids.astype(np.float32))
out, info, stats, cost = dev_empty((1, 64), 0), dev_empty(4, 0), dev_empty(128, 0), dev_double(2)
tgt = dev_empty((D, N), 0)
def show(name, fn): print('%-34s %7.2f us' % (name, timeit(fn)), flush=True)
show('l2sn pair stats', lambda: call('mcn_l2sn_pair_stats', ids, f, stats, D, N, None))
show('l2sn cost (m = -1)', lambda: call('mcn_l2sn_cost', ids, f, out, D, N, -1.0, info, stats, cost, None, None, 0, None))
show('l2sn cost (m = 3)', lambda: call('mcn_l2sn_cost', ids, f, out, D, N, 3.0, info, stats, cost, None, None, 0, None))
show('l2sn grad (recompute)', lambda: call('mcn_l2sn_grad', ids, f, tgt, D, N, -1.0, 0, 0.0025, None, None, 0, None))
show('l2sn grad from stats', lambda: call('mcn_l2sn_grad_from_stats', f, tgt, D, N, stats, info, 0, 0.0025, None))
x = dev(rng.standard_normal((K, N)).astype(np.float32)); p = dev_empty((K, N), 0)
nb = max(ws_bytes('mcn_softmax_workspace_bytes', K, N), ws_bytes('mcn_logreg_cost_workspace_bytes', K, N))
ws = capi.workspace(nb)
show('softmax 10575 x 128', lambda: call('mcn_softmax', x, p, K, N, ws, nb, None))

# ---- the multi-input FC layer (fc5): all inputs in one launch per direction, DeepID2 and DeepID2+ shapes
import ctypes as C  # noqa: E402


def ptr_array(bufs):
    return (C.c_void_p * len(bufs))(*[b.ptr if b is not None else None for b in bufs])


for tag, O_, Is in (('fc5', 160, [1024, 1200]), ('fc5+', 512, [2048, 3200])):
    Ws = [dev((rng.standard_normal((O_, I)) * 0.02).astype(np.float32)) for I in Is]
    Xs = [dev(np.maximum(rng.standard_normal((I, N)), 0).astype(np.float32)) for I in Is]
    dXs = [dev_empty((I, N), 0) for I in Is]
    dWs = [dev_empty((O_, I), 0) for I in Is]
    b, y, dy = dev(np.zeros(O_, np.float32)), dev_empty((O_, N), 0), dev(rng.standard_normal((O_, N)).astype(np.float32))
    wa, xa, ta_, ga, ka = ptr_array(Ws), ptr_array(Xs), ptr_array(dXs), ptr_array(dWs), (C.c_int * len(Is))(*Is)
    zf = (C.c_float * len(Is))(*([0.0] * len(Is)))
    one = (C.c_float * len(Is))(*([0.001] * len(Is)))
    mom = (C.c_float * len(Is))(*([0.9] * len(Is)))
    na, nt = ptr_array(Xs), (C.c_int * len(Is))(*([1] * len(Is)))
    flops = 2.0 * N * O_ * sum(Is)
    for name, fn in (('fwd (all inputs)', lambda: call('mcn_fc_fprop_multi', wa, xa, ka, len(Is), b, y, O_, N, 1, 0.0, 0.0, None)),
                     ('bwd-data + relu grad', lambda: call('mcn_fc_bprop_acts_multi', wa, ta_, ka, zf, na, nt, zf, zf, len(Is), dy, O_, N, None)),
                     ('bwd-weights', lambda: call('mcn_fc_bprop_weights_multi', dy, xa, ga, ka, one, mom, len(Is), O_, N, None))):
        us = timeit(fn)
        print('%-34s %7.2f us  %6.2f TFLOP/s' % (tag + ' ' + name, us, flops / us / 1e6), flush=True)

# ---- the bias gradients: row sums (mcn_add_sum, the code k_row_sums_multi runs per matrix)
for tag, rows, cols in (('conv2 bias', 32, 576 * N), ('local3 bias', 4800, N), ('local4 bias', 1024, N), ('fc6 bias', K, N),
                        ('conv2+ bias', 128, 576 * N), ('local3+ bias', 12800, N), ('local4+ bias', 2048, N)):
    m = dev(rng.standard_normal((rows, cols)).astype(np.float32))
    t = dev_empty((rows,), 0)
    wsb = ws_bytes('mcn_add_sum_workspace_bytes', rows, cols, 1)
    w = capi.workspace(max(wsb, 16))
    us = timeit(lambda: call('mcn_add_sum', t, m, rows, cols, 1, 0.0, 1.0, w, wsb, None))
    print('%-34s %7.2f us  %7.1f GB/s' % (tag + ' row sum', us, rows * cols * 4 / us / 1e3), flush=True)
