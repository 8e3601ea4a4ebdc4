"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU
oracle): the oracle must keep reproducing them (CPU test, guards the checker itself against drift) and
the HIP path must match them (GPU tests, through the C ABI and through `_ConvNet`).  The reference has no
golden vectors of its own for this path."""
import copy
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG  # noqa: E402

K = np.load(os.path.join(HERE, 'golden', 'kernels.npz'))
NET = np.load(os.path.join(HERE, 'golden', 'net_gc_small.npz'))


def assert_close(got, want, rtol, atol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.max(np.abs(got - want)) if got.size else 0.0
    bound = rtol * np.max(np.abs(want)) + atol
    assert err <= bound, '%s: max abs err %.3e > %.3e' % (what, err, bound)


def test_oracle_reproduces_the_kernel_goldens():
    now = MG.kernel_cases()
    assert sorted(now) == sorted(K.files)
    for k in K.files:
        assert_close(now[k], K[k], 1e-6, 1e-9, k)  # same code, same seeds: only compiler/libm noise allowed
    # decisions are exact
    assert np.array_equal(now['logreg.correct'], K['logreg.correct'])
    assert np.array_equal(now['l2sn.m'], K['l2sn.m'])
    assert np.array_equal(now['pool.max'], K['pool.max'])


def test_oracle_reproduces_the_net_golden():
    now = MG.net_case()
    assert sorted(now) == sorted(NET.files)
    for k in NET.files:
        assert_close(now[k], NET[k], 1e-5, 1e-9, k)


def test_golden_known_answers():
    """Hand-checkable facts about the fixture: the tie case of kLogregCost (layer_kernels.cu:82-90) and the
    first-strict-maximum margin rule (layer_kernels.cu:482-498)."""
    # case 3 has its label (5) tied with class 9 and the original maximum: correct = 1 / #argmax = 1/3
    assert int(np.sum(K['logreg.z'][:, 3] == K['logreg.z'][:, 3].max())) == 3
    assert K['logreg.correct'][0, 3] == np.float32(1.0) / np.float32(3.0)
    d, s = K['l2sn.dists'], K['l2sn.pair_labels']
    best, m = -1, d[0]
    for cand in d:  # getOptimalThr restated in python
        score = np.sum((s > 0) & (d < cand)) + np.sum((s < 0) & (d > cand))
        if score > best:
            best, m = score, cand
    assert K['l2sn.m'][0] == m
    pos = s > 0
    want = np.where(pos, 0.5 * d * d, np.where(d < m, 0.5 * (m - d) ** 2, 0))
    assert_close(K['l2sn.cost'][0], want, 1e-6, 1e-9, 'l2sn cost formula')


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('name', list(MG.GEOMS))
def test_hip_conv_trio_matches_golden(name):
    from kutil import call, close, dev, dev_empty
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = [int(v) for v in K[name + '.geom']]
    img, W, hid = K[name + '.img'], K[name + '.W'], K[name + '.hid']
    N = img.shape[1]
    pre = 'mcn_conv_' if conv else 'mcn_local_'
    out = dev_empty(K[name + '.filter_acts'].shape, np.nan)
    call(pre + 'filter_acts', dev(img), dev(W), out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 0.0, 1.0, None)
    close(out.numpy(), K[name + '.filter_acts'], what=name + ' filterActs')
    out = dev_empty(K[name + '.img_acts'].shape, np.nan)
    call(pre + 'img_acts', dev(hid), dev(W), out, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 0.0, 1.0, None)
    close(out.numpy(), K[name + '.img_acts'], what=name + ' imgActs')
    out = dev_empty(K[name + '.weight_acts'].shape, np.nan)
    if conv:
        call('mcn_conv_weight_acts', dev(img), dev(hid), out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 0, 0.0, 0.01, None)
    else:
        call('mcn_local_weight_acts', dev(img), dev(hid), out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 0.0, 0.01, None)
    close(out.numpy(), K[name + '.weight_acts'], what=name + ' weightActs')


@pytest.mark.gpu
def test_hip_pool_and_costs_match_golden():
    from kutil import call, close, dev, dev_empty, dev_double, ws_bytes, capi
    x, g = K['pool.x'], K['pool.g']
    N, F, imgY, imgX, outY, outX = x.shape[1], 16, 7, 6, 4, 3
    for mode, mid in (('max', 0), ('avg', 1)):
        out = dev_empty(K['pool.' + mode].shape, np.nan)
        call('mcn_local_pool', dev(x), out, N, F, imgY, imgX, 3, 0, 2, outY, outX, mid, None)
        got = out.numpy()
        if mode == 'max':
            assert np.array_equal(got, K['pool.max'])  # comparisons only: exact
        else:
            close(got, K['pool.avg'], what='avg pool')
    out = dev_empty(x.shape, np.nan)
    call('mcn_local_max_undo', dev(x), dev(g), dev(K['pool.max']), out, N, F, imgY, imgX, 3, 0, 2, outY, outX, 0.0, 1.0, None)
    close(out.numpy(), K['pool.max_undo'], what='max undo')
    # softmax + logreg: probabilities to tolerance, tie-aware correctness exactly
    z, labels = K['logreg.z'], K['logreg.labels']
    Kc, N = z.shape
    ws = capi.workspace(max(ws_bytes('mcn_softmax_workspace_bytes', Kc, N), ws_bytes('mcn_logreg_cost_workspace_bytes', Kc, N)))
    p = dev_empty(z.shape, np.nan)
    call('mcn_softmax', dev(z), p, Kc, N, ws, ws.nbytes, None)
    close(p.numpy(), K['logreg.probs'], what='softmax')
    lp, cp, cost = dev_empty((1, N), np.nan), dev_empty((1, N), np.nan), dev_double(2)
    call('mcn_logreg_cost', dev(labels), dev(K['logreg.probs']), lp, cp, Kc, N, cost, ws, ws.nbytes, None)
    assert np.array_equal(cp.numpy(), K['logreg.correct'])
    close(lp.numpy(), K['logreg.logprob'], what='logprob')
    # L2-SN: margin index and decisions exact
    ids, f = K['l2sn.ids'], K['l2sn.data']
    D, N2 = f.shape
    out, info, stats = dev_empty((1, N2 // 2), np.nan), dev_empty((4,), np.nan), dev_empty((N2,), np.nan)
    call('mcn_l2sn_cost', dev(ids), dev(f), out, D, N2, -1.0, info, stats, None, None, None, 0, None)
    inf = info.numpy()
    assert inf[0] == np.float32(K['l2sn.m'][0]) and int(inf[1]) == int(K['l2sn.m'][1])
    assert np.array_equal(stats.numpy()[:N2 // 2], K['l2sn.dists'])  # same summation order: bit-exact distances
    close(out.numpy(), K['l2sn.cost'], what='l2sn cost')
    tgt = dev_empty(f.shape, np.nan)
    call('mcn_l2sn_grad', dev(ids), dev(f), tgt, D, N2, -1.0, 0, 0.0025, None, None, 0, None)
    got = tgt.numpy()
    assert np.array_equal(got == 0, K['l2sn.grad'] == 0)  # which pairs are inside the margin
    close(got, K['l2sn.grad'], what='l2sn grad')
    for ug in (1, 0):
        W, inc = dev(K['sgd.W']), dev(K['sgd.inc'])
        call('mcn_sgd_update', W, inc, dev(K['sgd.grad']), 1000, 0.9, 5e-4, 5e-4, ug, None)
        close(W.numpy(), K['sgd.W_%d' % ug], rtol=1e-6, what='sgd W')
        close(inc.numpy(), K['sgd.inc_%d' % ug], rtol=1e-6, what='sgd inc')


@pytest.mark.gpu
def test_hip_net_matches_golden():
    import ccn_loader
    ccn_loader.load()
    from cuda_convnet_amd import layer
    from cuda_convnet_amd.convnet import import_native
    cfg = os.path.join(HERE, 'cfg')
    L = layer.parse_layers(os.path.join(cfg, 'gc_small.cfg'), os.path.join(cfg, 'gc_small_params.cfg'), {0: 432, 1: 1}, 10, seed=3)
    for l in L:
        if 'weights' in l:
            for i, w in enumerate(l['weights']):
                assert np.array_equal(w, NET['init.%s.w%d' % (l['name'], i)])  # the fixture's own initial weights
    data = [NET['data'], NET['labels']]
    lib = import_native()
    lib.initModel(L, 16, 0)
    try:
        lib.startBatch(data, 0)
        costs, n = lib.finishBatch()
        lib.syncWithHost()
        idx = [l['name'] for l in L].index('fc3')
        ftrs = np.zeros((48, 24), np.float32)
        lib.startFeatureWriter(data + [ftrs], idx)
        lib.finishBatch()
    finally:
        lib.destroyModel()
    assert n == 48
    for k, v in costs.items():
        assert_close(v[:1], NET['cost.' + k][:1], 2e-4, 1e-5, 'cost ' + k)
        assert v[1:] == list(NET['cost.' + k][1:])  # error count: exact
    for l in L:
        if 'weights' in l:
            for i in range(len(l['weights'])):
                assert_close(l['weightsInc'][i], NET['final.%s.winc%d' % (l['name'], i)], 2e-3, 1e-9, l['name'] + ' inc')
                assert_close(l['weights'][i], NET['final.%s.w%d' % (l['name'], i)], 1e-5, 1e-8, l['name'] + ' W')
            assert_close(l['biases'], NET['final.%s.b' % l['name']], 2e-3, 1e-9, l['name'] + ' b')
    assert_close(ftrs, NET['features.fc3'], 1e-4, 1e-6, 'features')
