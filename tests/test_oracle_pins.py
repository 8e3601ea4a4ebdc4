"""CPU tests that pin the oracle (oracle/) with everything the reference itself offers for this
path (there are NO golden vectors or known-answer tests in the reference):

  (i)   its only self-check, the finite-difference gradient check (convnet.cu:257-330) with its own
        steps (0.001/0.002 conv+local, 0.1/0.01 FC: layer.cu:294-295,351-352) and its own acceptance
        threshold 0.02 (util.cuh:47);
  (ii)  the layer output shapes its training logs print (deepid2/nohup.out:1-20, deepid_attr/nohup.out:1-32);
  (iii) the first-batch costs of those logs: logprob ~= ln(numClasses), error ~= 1 - 1/numClasses
        (deepid/nohup.out:54, deepid2/nohup.out:55).

Parity therefore stays "unpinned" in the strict sense (no reference output to compare with); these
are the soft pins, and they are what ties the oracle to the reference's own acceptance test.
"""
import copy
import math
import os

import numpy as np

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import CFG_DIR, layer  # noqa: E402
from cuda_convnet_amd.convnet import PRESETS  # noqa: E402
from cuda_convnet_amd.data import SyntheticDataProvider  # noqa: E402
from gradcheck import directional_check  # noqa: E402
from oracle.net import OracleNet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def small_net(cases, scale=1.0):
    dp = SyntheticDataProvider('dummy', cases, img=(12, 12, 3), minibatch=cases, num_classes=10)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(os.path.join(HERE, 'cfg', 'gc_small.cfg'), os.path.join(HERE, 'cfg', 'gc_small_params.cfg'),
                           dims, 10, seed=3)
    data = dp.get_next_batch()[2]
    return L, [(data[0] * scale).astype(np.float32), data[1]]


def test_reference_gradcheck_recipe():
    """Per-element finite differences with the reference's steps and threshold on a net holding every
    hot-path layer type (conv+relu, max pool, cmrnorm, local+tanh, avg pool, 2-input fc, fc, softmax,
    logreg, L2-SN).  Central differences at the reference's steps: its forward difference carries an
    O(step) curvature term which at the FC step 0.1 (twice the fc4 weight scale) alone exceeds 0.02.
    'fc4 weights[0]' keeps a cubic term of the same origin (0.05-0.2 depending on the elements drawn) and is
    exempt here; test_directional_gradcheck covers it at 2e-2 with a small step."""
    L, data = small_net(16, scale=8.0)
    res = OracleNet(L, 16).check_gradients(data, max_elems=30, seed=0, central=True)
    assert len(res) == 9
    for name, rel, ok in res:
        if name != 'fc4 weights[0]':
            assert rel < 0.02, (name, rel)


def test_reference_gradcheck_forward_difference():
    """The reference's exact recipe (forward difference): the matrices whose step is small against their
    scale pass its 0.02 threshold."""
    L, data = small_net(16, scale=8.0)
    res = dict((n, r) for n, r, _ in OracleNet(L, 16).check_gradients(data, max_elems=30, seed=0))
    for name in ('conv1 weights[0]', 'conv1 biases', 'fc3 weights[0]', 'fc3 biases', 'fc4 biases'):
        assert res[name] < 0.02, (name, res[name])


def test_directional_gradcheck():
    L, data = small_net(64)

    def train(Lx, d):
        net = OracleNet(Lx, d[0].shape[1])
        net.run_batch(d)
        net.sync_with_host()

    def cost(Lx, d):
        return OracleNet(Lx, d[0].shape[1]).run_batch(d, test=True)[0]

    res = directional_check(L, data, train, cost)
    assert len(res) == 9
    for name, num, anal, rel in res:
        assert rel < 0.02, (name, num, anal, rel)


def parse(preset, cases=128):
    ldef, lpar, kind, img = PRESETS[preset]
    dp = SyntheticDataProvider(kind, cases, img=img)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dims, dp.get_num_classes(), seed=3)
    return dp, L


def test_layer_shapes_match_reference_logs():
    _, L = parse('deepid2')
    by = {l['name']: l for l in L}
    # deepid2/nohup.out:1-20
    want = {'data': 9075, 'labels': 1, 'conv1': 52 * 52 * 16, 'pool1': 26 * 26 * 16, 'conv2': 24 * 24 * 32, 'pool2': 12 * 12 * 32,
            'local3': 10 * 10 * 48, 'pool3': 5 * 5 * 48, 'local4': 4 * 4 * 64, 'fc5': 160, 'fc6': 10575, 'probs': 10575,
            'conv1_neuron': 43264, 'conv2_neuron': 18432, 'local3_neuron': 4800, 'local4_neuron': 1024, 'fc5_neuron': 160,
            'fc6_neuron': 10575}
    for k, v in want.items():
        assert by[k]['outputs'] == v, (k, by[k]['outputs'], v)
    assert (by['conv1']['modulesY'], by['conv1']['modulesX']) == (52, 52)
    _, L = parse('deepid-attr')
    by = {l['name']: l for l in L}
    # deepid_attr/nohup.out:1-32 (the three attribute data layers are declared 1-dimensional: convdata.py:926-927)
    want = {'pose': 1, 'light': 1, 'expression': 1, 'fc-softmax': 4131, 'probs': 4131, 'fc-pose': 2, 'fc-light': 27,
            'fc-expression': 23, 'fc5': 160, 'local4': 1024}
    for k, v in want.items():
        assert by[k]['outputs'] == v, (k, by[k]['outputs'], v)


def test_first_batch_costs_match_reference_logs():
    """deepid2/nohup.out:55: logprob 9.2645 (ln 10575 = 9.266), error 0.9999; deepid/nohup.out:54: 8.3587 (ln 4213 = 8.346)."""
    for preset, K in (('deepid2', 10575), ('deepid', 4213)):
        dp, L = parse(preset)
        costs, n = OracleNet(copy.deepcopy(L), 128).run_batch(dp.get_next_batch()[2], test=True)
        assert abs(costs['logprob'][0] / n - math.log(K)) < 0.02 * math.log(K)
        assert costs['logprob'][1] / n >= 0.98


def test_fisher2_cost_and_gradient_are_consistent():
    """cost.fisher2 has no golden vector in the reference (parity unpinned).  Soft pins: the step-by-step C restatement of
    layer_kernels.cu:310-479 equals an independent float64 formula -a^2/V, its gradient is -dC/dX by central differences
    (the whole net back-propagates -dE/dx), and the value is a small negative number like the reference's own log
    (deepidn/nohup.out:69-75: fisher -0.04 ... -0.39 on the first batches)."""
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    D, N, alpha = 12, 16, 0.5
    X = (rng.standard_normal((D, N)) * 0.5).astype(np.float32)
    lab = rng.integers(0, 4, N).astype(np.float32).reshape(1, N)

    def cost64(Xd):
        l = lab.reshape(-1)
        same = l[None, :] == l[:, None]
        tri = np.triu(np.ones((N, N), bool), 1)
        diff = (np.where(same, 1, -1) * tri).sum()
        n = N * (N - 1) / 2
        M = np.where(same, 2 / (n + diff), -2 / (n - diff)) * tri
        W = np.where(same, alpha, 1 - alpha)
        W = W / W.sum()
        s = (Xd * Xd).sum(0)
        E = s[None, :] + s[:, None] - 2 * Xd.T @ Xd
        a, b = (M * E).sum(), (W * E).sum()
        return -a * a / (W * (E - b) ** 2).sum()

    c = O.fisher2_cost(lab, X, D, N, alpha)
    assert abs(c - cost64(X.astype(np.float64))) < 1e-5 * abs(c)
    assert -1.0 < c < 0.0
    g = O.fisher2_grad(lab, X, D, N, alpha, 1.0)
    num = np.zeros((D, N))
    h = 1e-5
    for d in range(D):
        for n in range(N):
            Xp, Xm = X.astype(np.float64), X.astype(np.float64)
            Xp[d, n] += h
            Xm[d, n] -= h
            num[d, n] = (cost64(Xp) - cost64(Xm)) / (2 * h)
    assert np.abs(g + num).max() < 1e-4 * np.abs(num).max()
    # add / coeff semantics of computeFisher2Grad (:471-478)
    t0 = rng.standard_normal((D, N)).astype(np.float32)
    g2 = O.fisher2_grad(lab, X, D, N, alpha, 0.25, target=t0.copy(), add=True)
    assert np.allclose(g2, t0 + 0.25 * g, rtol=1e-5, atol=1e-7)


def test_deepidn_preset_parses_with_fisher2():
    _, L = parse('deepidn')
    f = [l for l in L if l['type'] == 'cost.fisher2']
    assert len(f) == 1 and f[0]['alpha'] == 0.5 and f[0]['coeff'] == 1 and f[0]['numInputs'][0] == 1
