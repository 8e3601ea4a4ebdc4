#!/usr/bin/env python3
"""Generates the committed golden fixtures (inputs + expected outputs) from the CPU oracle.

The reference holds no golden vectors for this path and cannot be run here (CUDA-only; SURVEY.md
section 8c), so these vectors pin the ORACLE's answers at the time of generation: tests/test_golden.py
checks that the oracle still reproduces them (CPU) and that the HIP path matches them (GPU).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

from oracle import oracle as O  # noqa: E402

# name: (imgY, imgX, C, F, fs, mY, mX, pad, stride, groups, conv)
GEOMS = {
    'conv_c3': (9, 8, 3, 16, 3, 9, 8, -1, 1, 1, True),
    'conv_c16_s2': (9, 7, 16, 32, 3, 4, 3, 0, 2, 1, True),
    'local_c8': (6, 5, 8, 16, 2, 5, 4, 0, 1, 1, False),
    'local_g2': (5, 5, 8, 32, 3, 5, 5, -1, 1, 2, False),
}


def mk(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def kernel_cases():
    out = {}
    N = 12
    for gi, (name, g) in enumerate(GEOMS.items()):
        imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
        kw = dict(imgY=imgY, imgX=imgX, C_=C_, F=F, fs=fs, mY=mY, mX=mX, pad=pad, stride=stride, groups=groups)
        rng = np.random.default_rng(100 + gi)
        M = mY * mX
        rows = (C_ // groups) * fs * fs * (1 if conv else M)
        img, W, hid = mk(rng, C_ * imgY * imgX, N), mk(rng, rows, F, scale=0.1), mk(rng, F * M, N)
        out[name + '.geom'] = np.array([imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, int(conv)], np.int32)
        out[name + '.img'], out[name + '.W'], out[name + '.hid'] = img, W, hid
        out[name + '.filter_acts'] = O.filter_acts(img, W, N, conv=conv, **kw)
        out[name + '.img_acts'] = O.img_acts(hid, W, N, conv=conv, **kw)
        out[name + '.weight_acts'] = O.weight_acts(img, hid, N, conv=conv, partial_sum=0 if conv else 1, sO=0.01, **kw)
    # pooling 3/2 over an odd, clipped image
    rng = np.random.default_rng(200)
    F, imgY, imgX, outY, outX = 16, 7, 6, 4, 3
    x = mk(rng, F * imgY * imgX, N)
    x[:, ::3] = np.maximum(x[:, ::3], 0)  # exact ties at 0, as after a relu
    g = mk(rng, F * outY * outX, N)
    out['pool.x'], out['pool.g'] = x, g
    for mode in ('max', 'avg'):
        out['pool.' + mode] = O.local_pool(x, N, F, imgY, imgX, 3, 0, 2, outY, outX, mode)
    out['pool.max_undo'] = O.local_max_undo(x, g, out['pool.max'], N, F, imgY, imgX, 3, 0, 2, outY, outX)
    out['pool.avg_undo'] = O.local_avg_undo(g, N, F, imgY, imgX, 3, 0, 2, outY, outX)
    # response normalisation
    rng = np.random.default_rng(300)
    F, imgY, imgX = 16, 5, 4
    x, gr = mk(rng, F * imgY * imgX, N), mk(rng, F * imgY * imgX, N)
    out['norm.x'], out['norm.g'] = x, gr
    y, den = O.response_norm(x, N, F, imgY, imgX, 3, 0.001, 0.75)
    out['norm.rnorm'], out['norm.rnorm_denoms'] = y, den
    out['norm.rnorm_undo'] = O.response_norm_undo(gr, den, x, y.copy(), N, F, imgY, imgX, 3, 0.001, 0.75)
    y, den = O.response_norm_cross_map(x, N, F, imgY * imgX, 5, 0.001, 0.75)
    out['norm.cmrnorm'], out['norm.cmrnorm_denoms'] = y, den
    out['norm.cmrnorm_undo'] = O.response_norm_cross_map_undo(gr, den, x, y.copy(), N, F, imgY * imgX, 5, 0.001, 0.75)
    # softmax + logreg (ties in the maximum included)
    rng = np.random.default_rng(400)
    K = 37
    z = mk(rng, K, N, scale=3.0)
    z[5, 3] = z[:, 3].max()
    z[9, 3] = z[5, 3]
    labels = rng.integers(0, K, (1, N)).astype(np.float32)
    labels[0, 3] = 5
    p = O.softmax(z, K, N)
    lp, cp = O.logreg_cost(labels, p, K, N)
    out['logreg.z'], out['logreg.labels'], out['logreg.probs'], out['logreg.logprob'], out['logreg.correct'] = z, labels, p, lp, cp
    out['logreg.grad'] = O.logreg_softmax_grad(labels, p, K, N, 1.0)
    # L2-SN (data-dependent margin) and L2-reg
    rng = np.random.default_rng(500)
    D, N2 = 24, 32
    ids = rng.integers(0, 50, N2)
    same = rng.random(N2 // 2) < 0.5
    ids[N2 // 2:][same] = ids[:N2 // 2][same]
    ids = ids.astype(np.float32).reshape(1, N2)
    f = mk(rng, D, N2)
    cost, m, midx, dists, plab = O.l2sn_cost(ids, f, D, N2, -1.0)
    out['l2sn.ids'], out['l2sn.data'], out['l2sn.cost'] = ids, f, cost
    out['l2sn.m'], out['l2sn.dists'], out['l2sn.pair_labels'] = np.array([m, midx], np.float64), dists, plab
    out['l2sn.grad'] = O.l2sn_grad(ids, f, D, N2, 0.0025, -1.0)
    gt = mk(rng, D, N2)
    out['l2reg.gt'] = gt
    out['l2reg.cost'] = O.l2reg_cost(gt, f, D, N2)
    out['l2reg.grad'] = O.l2reg_grad(gt, f, D, N2, 0.01)
    # momentum / weight-decay update, both flavours
    rng = np.random.default_rng(600)
    W, inc, gr = mk(rng, 1000), mk(rng, 1000, scale=0.01), mk(rng, 1000, scale=0.001)
    out['sgd.W'], out['sgd.inc'], out['sgd.grad'] = W, inc, gr
    for ug in (1, 0):
        w2, i2 = W.copy(), inc.copy()
        O.sgd_update(w2, i2, gr, 0.9, 5e-4, 5e-4, ug)
        out['sgd.W_%d' % ug], out['sgd.inc_%d' % ug] = w2, i2
    return out


def net_case():
    """Three training steps of the small all-layer-types net + a feature pass."""
    import ccn_loader
    ccn_loader.load()
    from cuda_convnet_amd import layer
    from cuda_convnet_amd.data import SyntheticDataProvider
    from oracle.net import OracleNet
    cfg = os.path.join(ROOT, 'tests', 'cfg')
    dp = SyntheticDataProvider('dummy', 48, img=(12, 12, 3), minibatch=16, num_classes=10)
    L = layer.parse_layers(os.path.join(cfg, 'gc_small.cfg'), os.path.join(cfg, 'gc_small_params.cfg'), {0: 432, 1: 1}, 10, seed=3)
    data = dp.get_next_batch()[2]
    out = {'data': data[0], 'labels': data[1]}
    L0 = copy.deepcopy(L)
    net = OracleNet(L, 16)
    costs, n = net.run_batch(data)
    net.sync_with_host()
    for k, v in costs.items():
        out['cost.' + k] = np.array(v, np.float64)
    for l0, l in zip(L0, L):
        if 'weights' in l:
            for i in range(len(l['weights'])):
                out['init.%s.w%d' % (l['name'], i)] = np.ascontiguousarray(l0['weights'][i])
                out['final.%s.w%d' % (l['name'], i)] = np.ascontiguousarray(l['weights'][i])
                out['final.%s.winc%d' % (l['name'], i)] = np.ascontiguousarray(l['weightsInc'][i])
            out['final.%s.b' % l['name']] = l['biases']
    idx = [l['name'] for l in L].index('fc3')
    out['features.fc3'] = net.write_features(data, idx)
    return out


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'kernels.npz'), **kernel_cases())
    np.savez_compressed(os.path.join(HERE, 'net_gc_small.npz'), **net_case())
    for f in ('kernels.npz', 'net_gc_small.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')
