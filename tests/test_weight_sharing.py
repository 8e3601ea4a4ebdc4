"""Weight sharing (`weightSource`: weights.cuh:59-68, :150-181; layer.cu:277-289, :378, :481, :554) in the CPU oracle,
checked against an independent construction: the same net WITHOUT sharing, whose towers start from equal copies of the
matrices.  After one step from zero momentum

  * a shared conv matrix (gradient buffer, layer.cu:481: scaleTargets = numUpdates > 0) holds the SUM of the towers'
    increments -- each scaled by its own epsW, weight decay from the owner only;
  * a shared fc / local matrix (written straight into `inc` with scaleInc = (numUpdates == 0) * mom, layer.cu:378, :554)
    holds only the LAST contribution of the backward pass -- the reference's own rule, restated as it is.

CPU only (the product's side of it: tests/test_weight_sharing_gpu.py)."""
import copy
import os

import numpy as np

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import layer  # noqa: E402
from cuda_convnet_amd.data import SyntheticDataProvider  # noqa: E402
from oracle.net import OracleNet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def siamese(cases=32, minibatch=32):
    dp = SyntheticDataProvider('dummy', cases, img=(16, 16, 3), minibatch=minibatch, num_classes=10)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(os.path.join(HERE, 'cfg', 'siamese.cfg'), os.path.join(HERE, 'cfg', 'siamese_params.cfg'), dims, 10, seed=1)
    return L, dp.get_next_batch()[2]


def unshare(L):
    """the same layers with every shared matrix replaced by a private copy"""
    U = copy.deepcopy(L)
    for l in U:
        if 'weightSourceLayerIndices' in l:
            for i, li in enumerate(l['weightSourceLayerIndices']):
                if li >= 0:
                    l['weights'][i] = l['weights'][i].copy(order='K')
                    l['weightsInc'][i] = l['weightsInc'][i].copy(order='K')
            l['weightSourceLayerIndices'] = [-1] * len(l['weightSourceLayerIndices'])
            l['weightSourceMatrixIndices'] = [-1] * len(l['weightSourceMatrixIndices'])
    return U


def by_name(L, name):
    return L[layer.layer_index(L, name)]


def test_parser_hands_out_the_owner_s_arrays():
    L, _ = siamese()
    for a, b in (('conv1', 'conv1b'), ('local2', 'local2b'), ('fc3', 'fc3b')):
        for w, v in zip(by_name(L, a)['weights'], by_name(L, b)['weights']):
            assert w is v
        assert by_name(L, b)['weightSourceLayerIndices'] == [layer.layer_index(L, a)] * len(by_name(L, b)['weights'])
    assert by_name(L, 'fc3b')['weightSourceMatrixIndices'] == [0, 1]


def test_shared_matrices_after_one_step_against_the_unshared_towers():
    L, data = siamese()
    U = unshare(L)
    ns, nu = OracleNet(L, 32), OracleNet(U, 32)
    cs, _ = ns.run_batch(data)
    cu, _ = nu.run_batch(data)
    assert cs == cu  # forward passes are the same computation
    ns.sync_with_host()
    nu.sync_with_host()
    # conv: both towers' gradients, each with its own epsW (0.001 / 0.0005), the owner's weight decay
    got = by_name(L, 'conv1')['weightsInc'][0]
    want = by_name(U, 'conv1')['weightsInc'][0] + by_name(U, 'conv1b')['weightsInc'][0]
    assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))
    assert np.max(np.abs(by_name(U, 'conv1b')['weightsInc'][0])) > 0.05 * np.max(np.abs(want))  # (the sharer's part is not negligible)
    # local / fc: the backward pass reaches tower A first (eltsum walks its inputs in order, layer.cu:150-156), so tower B's
    # contribution is the one that stays; the owner's update then subtracts ITS weight decay (sharers have wc = 0)
    for a, b, eps_a, wc_a in (('local2', 'local2b', 0.001, 0.004), ('fc3', 'fc3b', 0.001, 0.004)):
        for i, (got, last) in enumerate(zip(by_name(L, a)['weightsInc'], by_name(U, b)['weightsInc'])):
            w0 = by_name(U, b)['weights'][i] - last  # the common starting matrix (U's sharer has no weight decay: W1 = W0 + inc)
            want = last - np.float32(wc_a * eps_a) * w0
            assert np.max(np.abs(got - want)) <= 2e-6 * np.max(np.abs(want)), (a, i)
    # ... and the weights moved by exactly that increment, in both dicts of the sharing net (one array)
    assert by_name(L, 'fc3')['weights'][1] is by_name(L, 'fc3b')['weights'][1]


def test_update_counter_lives_in_the_owner():
    L, data = siamese()
    net = OracleNet(L, 32)
    net.fprop(data)
    net.bprop()
    own, shr = net.layer('conv1').weights[0], net.layer('conv1b').weights[0]
    assert shr.src is own and own.num_updates == 2 and shr.num_updates == 2  # weights.cuh:165-181
    assert shr.wc == 0.0 and shr.mom == own.mom and shr.eps == 0.0005
    net.update_weights()
    assert own.num_updates == 0 and shr.num_updates == 0  # weights.cuh:161
