"""Weight sharing (`weightSource`) through the `_ConvNet` boundary against the whole-net CPU oracle: a siamese net whose
conv, local and two-input fc layers are tied (tests/cfg/siamese.cfg, the shape of the reference's archival
layers/deepid_HailinShi cfg) and the two tied fc layers of tests/cfg/shared.cfg.  Tolerances as in test_net_gpu.py
(three steps: costs 2e-4, increments 2e-3, error counts exact).  Reference: weights.cuh:59-68, :150-181; layer.cu:277-289,
:378, :481, :554."""
import copy
import os

import numpy as np
import pytest

import ccn_loader

ccn_loader.load()
from test_net_gpu import Native, build, check_costs, check_weights, oracle_train  # noqa: E402
from test_weight_sharing import unshare  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

NETS = {'siamese': ('siamese.cfg', 'siamese_params.cfg', (16, 16, 3)), 'shared': ('shared.cfg', 'shared_params.cfg', (4, 4, 3))}


def net(name, cases=96, minibatch=32):
    ldef, lpar, img = NETS[name]
    return build(os.path.join(HERE, 'cfg', ldef), os.path.join(HERE, 'cfg', lpar), 'dummy', img, cases, minibatch=minibatch, num_classes=10)


@pytest.mark.parametrize('name', ['siamese', 'shared'])
def test_shared_weights_train_3_steps_matches_oracle(name):
    dp, Lg = net(name)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    _, want = oracle_train(Lc, data, minibatch=32)
    with Native(Lg, 32) as lib:
        lib.startBatch(data, 0)
        got = lib.finishBatch()
        lib.syncWithHost()
    check_costs(got, want)
    check_weights(Lg, Lc)
    for lg in Lg:  # the sharer's dict still shows the owner's (updated) arrays
        for i, li in enumerate(lg.get('weightSourceLayerIndices', [])):
            if li >= 0:
                assert lg['weights'][i] is Lg[li]['weights'][lg['weightSourceMatrixIndices'][i]]


def run3(name, env, dp_rank0=False, strip=None):
    dp, L = net(name)
    if strip:
        L = strip(L)
    data = dp.get_next_batch()[2]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with Native(L, 32) as lib:
            if dp_rank0:
                lib.initDataParallel(0, 1, lib.commUniqueId(), 1)
            lib.startBatch(data, 0)
            costs = lib.finishBatch()
            lib.syncWithHost()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return L, costs


def same_weights(La, Lb, what):
    for la, lb in zip(La, Lb):
        if 'weights' not in la:
            continue
        for key in ('weights', 'weightsInc'):
            for i, (a, b) in enumerate(zip(la[key], lb[key])):
                assert np.array_equal(a, b), '%s %s[%d] differs %s' % (la['name'], key, i, what)
        for key in ('biases', 'biasesInc'):
            assert np.array_equal(la[key], lb[key]), '%s %s differs %s' % (la['name'], key, what)


@pytest.mark.parametrize('name', ['siamese', 'shared'])
def test_shared_weights_two_streams_equal_one_stream_bitwise(name):
    """Two layers accumulate into one matrix: on two streams that is only safe because every gradient piece of a sharing
    net goes to the side stream in pass order.  Any difference from the one-stream schedule is a race."""
    La, ca = run3(name, {})
    Lb, cb = run3(name, {'MCN_NO_GRAD_STREAM': '1'})
    assert ca == cb
    same_weights(La, Lb, 'between the two-stream and the one-stream schedule')
    Lc, cc = run3(name, {'MCN_NO_FUSION': '1'})  # every layer as its own kernels: same sums up to the fused epilogues' rounding
    for k in ca[0]:
        assert abs(ca[0][k][0] - cc[0][k][0]) <= 1e-5 * abs(cc[0][k][0])


def conv_only(L):
    """siamese.cfg with the local / fc ties cut (private copies): only the conv matrix is still shared"""
    keep = {'conv1b'}
    U = unshare([l for l in L])
    for lu, lo in zip(U, L):
        if lo['name'] in keep:
            lu['weightSourceLayerIndices'] = list(lo['weightSourceLayerIndices'])
            lu['weightSourceMatrixIndices'] = list(lo['weightSourceMatrixIndices'])
            for i, li in enumerate(lu['weightSourceLayerIndices']):
                lu['weights'][i] = U[li]['weights'][lu['weightSourceMatrixIndices'][i]]
                lu['weightsInc'][i] = U[li]['weightsInc'][lu['weightSourceMatrixIndices'][i]]
    return U


def test_shared_conv_weights_on_the_replica_exchange_path():
    """Data-parallel mode keeps raw gradients in the arena: the owner's segment is complete only when the sharer has added
    its part (ConvNet::gradReady counts contributors).  One-rank communicator == plain step, at the 3-step tolerance."""
    La, ca = run3('siamese', {}, strip=conv_only)
    Lb, cb = run3('siamese', {}, dp_rank0=True, strip=conv_only)
    check_costs(cb, ca)
    check_weights(Lb, La)
    dp, Lc = net('siamese')
    Lc = conv_only(Lc)
    Lo = copy.deepcopy(Lc)
    _, want = oracle_train(Lo, dp.get_next_batch()[2], minibatch=32)
    check_costs(ca, want)
    check_weights(La, Lo)


def test_shared_fc_weights_are_refused_for_replicas():
    dp, L = net('shared')
    with Native(L, 32) as lib:
        with pytest.raises(RuntimeError, match='not defined for data-parallel'):
            lib.initDataParallel(0, 1, lib.commUniqueId(), 1)


def test_shared_weights_step_graph_replay_equals_eager_bitwise():
    outs = []
    for env in ({'MCN_STEP_GRAPH': '1'}, {}):
        dp, L = net('siamese')
        data = dp.get_next_batch()[2]
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            with Native(L, 32) as lib:
                lib.loadBatch(data)
                lib.startSteps(8, 0, 0)
                lib.finishBatch()
                lib.syncWithHost()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        outs.append(L)
    same_weights(outs[0], outs[1], 'between graph replay and eager submission')
