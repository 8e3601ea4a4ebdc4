"""GPU parity tests of the whole training path: the `_ConvNet` extension (C++ layer graph over the
C ABI, HIP kernels) against the whole-net CPU oracle (oracle/net.py) on identical seeded inputs,
for every BASELINE config.

Stated fp32 tolerance after 3 consecutive SGD steps (momentum and weight decay engaged):
  * cost sums: |a-b| <= 2e-4*|b| + 1e-5;  the top-1 error COUNT (logreg cost[1]) must agree exactly;
  * weight increments (what a step changes): per tensor max|a-b| <= 2e-3*max|b| + 1e-9;
  * weights: per tensor max|a-b| <= 1e-5*max|b| + 1e-8 (biases start at zero and are held to the increment bound).
The wider increment bound covers fp32 summation-order differences compounding through three
forward/backward passes; single-kernel parity at 1e-4 is in test_kernels_gpu.py.
"""
import copy
import os

import numpy as np
import pytest

import ccn_loader

ccn = ccn_loader.load()
from cuda_convnet_amd import CFG_DIR, layer  # noqa: E402
from cuda_convnet_amd.convnet import PRESETS, import_native  # noqa: E402
from cuda_convnet_amd.data import SyntheticDataProvider  # noqa: E402
from oracle.net import OracleNet  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def build(ldef, lpar, kind, img, cases, minibatch=128, seed=3, num_classes=None):
    dp = SyntheticDataProvider(kind, cases, img=img, minibatch=minibatch, num_classes=num_classes)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(ldef, lpar, dims, dp.get_num_classes(), seed=seed)
    return dp, L


def preset(name, cases, minibatch=128):
    ldef, lpar, kind, img = PRESETS[name]
    return build(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), kind, img, cases, minibatch)


class Native:
    """One model per process (pyconvnet.cu:45); always destroyed on exit."""

    def __init__(self, layers, minibatch=128, hooks=False):
        self.lib = import_native(hooks=hooks)
        self.lib.initModel(layers, minibatch, 0)

    def __enter__(self):
        return self.lib

    def __exit__(self, *a):
        self.lib.destroyModel()


def check_costs(got, want):
    gc, gn = got
    wc, wn = want
    assert gn == wn
    assert list(gc) == list(wc), 'cost layers must be reported in cfg order'
    for k in wc:
        assert len(gc[k]) == len(wc[k])
        assert abs(gc[k][0] - wc[k][0]) <= 2e-4 * abs(wc[k][0]) + 1e-5, (k, gc[k], wc[k])
        if len(wc[k]) > 1:  # top-1 error count: exact
            assert gc[k][1] == wc[k][1], (k, gc[k], wc[k])


def check_weights(Lg, Lc):
    for lg, lc in zip(Lg, Lc):
        if 'weights' not in lg:
            continue
        pairs = [('weights[%d]' % i, a, b) for i, (a, b) in enumerate(zip(lg['weights'], lc['weights']))]
        pairs += [('weightsInc[%d]' % i, a, b) for i, (a, b) in enumerate(zip(lg['weightsInc'], lc['weightsInc']))]
        pairs += [('biases', lg['biases'], lc['biases']), ('biasesInc', lg['biasesInc'], lc['biasesInc'])]
        for nm, a, b in pairs:
            assert np.all(np.isfinite(a)), (lg['name'], nm)
            rt, at = (2e-3, 1e-9) if ('Inc' in nm or nm == 'biases') else (1e-5, 1e-8)  # biases start at 0: they ARE increments
            err = float(np.max(np.abs(a.astype(np.float64) - b)))
            bound = rt * float(np.max(np.abs(b))) + at
            assert err <= bound, '%s %s: max abs err %.3e > %.3e' % (lg['name'], nm, err, bound)
            if 'Inc' in nm:
                assert float(np.max(np.abs(b))) > 0, '%s %s: the oracle never updated it' % (lg['name'], nm)


def oracle_train(Lc, data, minibatch=128, test=False):
    net = OracleNet(Lc, minibatch)
    out = net.run_batch(data, test=test)
    net.sync_with_host()  # write the stepped weights back into Lc's arrays, like syncWithHost does
    return net, out


@pytest.mark.parametrize('name', ['deepid-39x31', 'deepid', 'deepid2', 'deepid-attr', 'cifar10', 'deepid2plus', 'deepidn', 'cifar10-cnorm', 'cifar10-imgops'])
def test_train_3_steps_matches_oracle(name):
    """C1..C5: three consecutive minibatches of 128 through fprop/bprop/update, then syncWithHost."""
    dp, Lg = preset(name, 384)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    _, want = oracle_train(Lc, data)
    with Native(Lg) as lib:
        lib.startBatch(data, 0)
        got = lib.finishBatch()
        lib.syncWithHost()
    check_costs(got, want)
    check_weights(Lg, Lc)


def expected_buckets(segments, bucket_floats):
    """The plan of DESIGN.md section 6: segments in arena order (= reverse layer order = completion order of backward) are
    merged until a bucket holds at least bucket_floats; what is left when backward ends goes in one last exchange.  The
    last weight layer of a pass is finished by the main thread and always travels with that last exchange."""
    out, start = [], 0
    for i, (name, b, e) in enumerate(segments):
        last = i == len(segments) - 1
        if not last and e - start >= bucket_floats:
            out.append((start, e - start))
            start = e
    total = segments[-1][2]
    if total > start:
        out.append((start, total - start))
    return out


@pytest.mark.parametrize('name', ['deepid2', 'deepid2plus', 'deepid-attr'])
def test_data_parallel_plan_is_the_documented_one_and_repeats(name):
    """Every rank must issue the same collectives in the same order or RCCL hangs.  With a one-rank communicator the
    exchange thread still runs the whole plan; its log must (1) cover the gradient arena exactly once per step with
    buckets that end on layer boundaries, (2) be the greedy plan above, (3) put the L2-SN all-gather pair before the
    all-reduces, (4) repeat identically every step and across two models."""
    logs = []
    for run in range(2):
        dp, Lg = preset(name, 384)
        data = dp.get_next_batch()[2]
        with Native(Lg) as lib:
            lib.initDataParallel(0, 1, lib.commUniqueId(), 1)
            segments, bucket = lib.getGradSegments()
            lib.getCommLog()
            lib.startBatch(data, 0)
            lib.finishBatch()
            logs.append(lib.getCommLog())
    assert logs[0] == logs[1], 'the exchange sequence differs between two identical runs'
    log = logs[0]
    assert len(log) % 3 == 0
    per = len(log) // 3
    steps = [log[i * per:(i + 1) * per] for i in range(3)]
    assert steps[0] == steps[1] == steps[2], 'the exchange sequence differs from step to step'
    gathers = [r for r in steps[0] if r[0] == 'allgather']
    reduces = [(o, c) for k, o, c in steps[0] if k == 'allreduce']
    assert len(gathers) == 2 and all(c == 64 for _, _, c in gathers) and steps[0][:2] == gathers  # pair distances, pair labels first
    assert reduces == expected_buckets(segments, bucket), (reduces, segments, bucket)
    ends = {e for _, _, e in segments}
    assert all(o + c in ends for o, c in reduces)
    if name == 'deepid2':  # fc6 | fc5 + local4 + local3 | conv2 + conv1
        names = [[n for n, b, e in segments if o <= b and e <= o + c] for o, c in reduces]
        assert names == [['fc6'], ['fc5', 'local4', 'local3'], ['conv2', 'conv1']], names


@pytest.mark.parametrize('name', ['deepid2', 'cifar10', 'deepid2plus'])
def test_data_parallel_path_single_rank(name):
    """The replica-exchange path (raw gradients into the arena, bucketed RCCL all-reduce on the side stream,
    momentum folded after the exchange, L2-SN margin over all-gathered pairs) with a one-rank communicator
    must reproduce the single-GPU result; the N > 1 algorithm is covered on CPU in test_dp_gloo.py."""
    dp, Lg = preset(name, 384)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    _, want = oracle_train(Lc, data)
    with Native(Lg) as lib:
        lib.initDataParallel(0, 1, lib.commUniqueId(), 1)
        lib.startBatch(data, 0)
        got = lib.finishBatch()
        lib.syncWithHost()
    check_costs(got, want)
    check_weights(Lg, Lc)


@pytest.mark.parametrize('name,cases', [('deepid-39x31', 1), ('deepid-39x31', 33), ('deepid2', 130), ('deepid2', 200)])
def test_ragged_batches(name, cases):
    """A data batch that is not a multiple of the minibatch ends with a short minibatch
    (DataProvider::getNumMinibatches data.cu:100-103).  Odd sizes only without L2-SN: the reference's
    gallery/probe halves (layer_kernels.cu:517-527) need an even minibatch."""
    dp, Lg = preset(name, cases)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    _, want = oracle_train(Lc, data)
    with Native(Lg) as lib:
        lib.startBatch(data, 0)
        got = lib.finishBatch()
        lib.syncWithHost()
    check_costs(got, want)
    check_weights(Lg, Lc)


@pytest.mark.parametrize('name', ['deepid2', 'deepid-attr', 'deepidn'])
def test_test_pass_with_a_ragged_last_minibatch(name):
    """Cost kernels are deferred past the forward pass that issued them; each must keep ITS minibatch's case count (a test
    pass over 3 x 128 + 72 cases once counted the third minibatch with the fourth's 72 cases)."""
    dp, Lg = preset(name, 456)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    _, want = oracle_train(Lc, data, test=True)
    with Native(Lg) as lib:
        lib.startBatch(data, 1)
        got = lib.finishBatch()
    check_costs(got, want)


def test_test_pass_leaves_weights_alone_and_resident_batch():
    dp, Lg = preset('deepid2', 256)
    Lc = copy.deepcopy(Lg)
    before = [w.copy() for l in Lg if 'weights' in l for w in l['weights']]
    data = dp.get_next_batch()[2]
    _, want = oracle_train(Lc, data, test=True)
    with Native(Lg) as lib:
        lib.startBatch(data, 1)
        got = lib.finishBatch()
        lib.loadBatch(data)          # keep the batch in HBM, then run on it without passing host data
        lib.startBatch([], 1)
        got2 = lib.finishBatch()
        lib.syncWithHost()
    check_costs(got, want)
    check_costs(got2, want)
    after = [w for l in Lg if 'weights' in l for w in l['weights']]
    for a, b in zip(before, after):
        assert np.array_equal(a, b)


def test_write_features_matches_oracle():
    """C4: startFeatureWriter on fc5 and fc-expression (shownet.py:238-259, worker.cu:191-214)."""
    dp, Lg = preset('deepid-attr', 200)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    onet = OracleNet(Lc, 128)
    with Native(Lg) as lib:
        for lname in ('fc5', 'fc-expression'):
            idx = layer.layer_index(Lg, lname)
            ftrs = np.full((200, Lg[idx]['outputs']), np.nan, np.float32)
            lib.startFeatureWriter(list(data) + [ftrs], idx)
            lib.finishBatch()
            want = onet.write_features(data, idx)
            err = np.max(np.abs(ftrs - want))
            assert err <= 1e-4 * np.max(np.abs(want)) + 1e-6, (lname, err)


@pytest.mark.parametrize('name,real,views', [('cifar10', 200, 3), ('deepid-39x31', 128, 2), ('deepid2', 96, 2)])
def test_multiview_test_matches_oracle(name, real, views):
    """startMultiviewTest (pyconvnet.cu:112-127, MultiviewTestWorker worker.cu:142-174): probabilities averaged over the
    views of each case, logreg cost of the average; the other costs report their last view."""
    dp, Lg = preset(name, real * views)
    Lc = copy.deepcopy(Lg)
    data = dp.get_next_batch()[2]
    # the views of a case share its label (view-major layout: case c of view v is column v*real + c)
    data[1] = np.ascontiguousarray(np.tile(data[1][:, :real], (1, views)))
    idx = layer.layer_index(Lg, 'logprob')
    want = OracleNet(Lc, 128).run_multiview(data, views, idx)
    with Native(Lg) as lib:
        lib.startMultiviewTest(data, views, idx)
        got = lib.finishBatch()
    assert got[1] == want[1] == real
    check_costs(got, want)


def test_checkpoint_save_load_resume_is_bit_identical(tmp_path):
    """gpumodel.py:254-281 / convnet.py:55-60: syncWithHost fills the weight AND momentum arrays, save_state pickles
    {'model_state', 'op'}, a new process-model built from the checkpoint's layers (hyper-parameters refreshed from the
    params file) continues exactly where the first stopped: the step after resume equals the uninterrupted run's."""
    from cuda_convnet_amd.convnet import ConvNet, make_preset
    ldef, lpar, kind, img = PRESETS['deepid-attr']
    kw = dict(num_cases=256, num_batches=3)

    def weights(net):
        return [np.array(a) for l in net.layers if 'weights' in l for a in l['weights'] + l['weightsInc'] + [l['biases'], l['biasesInc']]]

    straight = make_preset('deepid-attr', **kw)
    try:
        straight.train(1, quiet=True)  # one epoch = three data batches of two minibatches
        straight.sync_with_host()
        want = weights(straight)
    finally:
        straight.close()

    first = make_preset('deepid-attr', **kw)
    try:
        dp = first.train_data_provider
        for _ in range(2):
            first.start_batch(dp.get_next_batch())
            first.finish_batch()
        first.epoch, first.batchnum = 1, 2
        first.sync_with_host()
        ck = str(tmp_path / 'ConvNet__test' / '1.2')
        first.save_state(ck)
    finally:
        first.close()

    st = ConvNet.load_state(ck)
    assert set(st) == {'model_state', 'op'} and st['model_state']['batchnum'] == 2
    dp2 = SyntheticDataProvider(kind, 256, num_batches=3, minibatch=128, img=img)
    dp2.curr_batchnum = 3  # resume after batch 2
    resumed = ConvNet(st['op']['layer_def'], st['op']['layer_params'], dp2, minibatch_size=st['op']['minibatch_size'],
                      layers=st['model_state']['layers'])
    try:
        resumed.start_batch(dp2.get_next_batch())
        resumed.finish_batch()
        resumed.sync_with_host()
        got = weights(resumed)
    finally:
        resumed.close()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b), 'resumed training differs from the uninterrupted run'


def test_write_features_mat_files(tmp_path):
    """shownet.py:238-259: `data_batch_<n>.mat` with key 'feats', one per test batch."""
    import scipy.io
    from cuda_convnet_amd.convnet import make_preset
    net = make_preset('deepid-39x31', num_cases=200, num_batches=2)
    try:
        paths = net.write_features_mat('fc5', str(tmp_path / 'ftrs'))
        assert [os.path.basename(p) for p in paths] == ['data_batch_1.mat', 'data_batch_2.mat']
        data = net.test_data_provider.make_batch(2)
        want = net.write_features('fc5', data)
        got = scipy.io.loadmat(paths[1])['feats']
        assert got.dtype == np.float32 and np.array_equal(got, want) and got.shape == (200, 160)
    finally:
        net.close()


def final_weights(L):
    return [np.array(a) for l in L if 'weights' in l for a in l['weights'] + l['weightsInc'] + [l['biases'], l['biasesInc']]]


@pytest.mark.parametrize('pinned', [False, True])
def test_prefetched_batches_equal_plain_batches(pinned):
    """f1: batch k+1 goes to the spare device slot over the copy stream (own thread; page-locked arrays by DMA, pageable ones
    through two staging chunks) while batch k trains; startBatch([]) then trains on it.  Same weights, bit for bit, as
    handing every batch to startBatch."""
    dp, Lp = preset('deepid2', 256)
    Lq = copy.deepcopy(Lp)
    batches = []
    for b in range(3):
        dpb = SyntheticDataProvider('deepid2', 256, img=(55, 55, 3), minibatch=128, seed=11 + b)
        batches.append(dpb.get_next_batch()[2])
    with Native(Lp) as lib:
        plain_costs = []
        for data in batches:
            lib.startBatch(data, 0)
            plain_costs.append(lib.finishBatch())
        lib.syncWithHost()
    with Native(Lq) as lib:
        if pinned:
            staged = []
            for data in batches:
                arrs = [lib.pinnedArray(*d.shape) for d in data]
                for a, d in zip(arrs, data):
                    a[...] = d
                staged.append(arrs)
            batches = staged
        costs = []
        lib.startBatch(batches[0], 0)
        for nxt in batches[1:]:
            lib.prefetchBatch(nxt)
            costs.append(lib.finishBatch())
            lib.startBatch([], 0)
        costs.append(lib.finishBatch())
        lib.startBatch([], 1)  # the adopted batch stays resident: a test pass on it
        lib.finishBatch()
        lib.syncWithHost()
    for a, b in zip(final_weights(Lq), final_weights(Lp)):
        assert np.array_equal(a, b)
    for (ga, na), (gb, nb_) in zip(costs, plain_costs):
        assert na == nb_
        for k in gb:
            assert all(abs(x - y) <= 1e-9 * abs(y) for x, y in zip(ga[k], gb[k])), (k, ga[k], gb[k])


@pytest.mark.parametrize('name,cases', [('deepid2', 456), ('deepid-attr', 256)])
def test_batch_streamed_from_the_host_equals_resident_batch(name, cases, monkeypatch):
    """MCN_MAX_DATA_ON_GPU_MB=0 (the reference's path for batches above 200 MB, util.cuh:53): the batch stays on the host and
    minibatch slices arrive over the copy stream two steps ahead.  Same costs and weights as the resident batch."""
    dp, La = preset(name, cases)
    Lb = copy.deepcopy(La)
    data = dp.get_next_batch()[2]
    with Native(La) as lib:
        lib.startBatch(data, 0)
        ca = lib.finishBatch()
        lib.startBatch(data, 1)
        ta = lib.finishBatch()
        lib.syncWithHost()
    monkeypatch.setenv('MCN_MAX_DATA_ON_GPU_MB', '0')
    with Native(Lb) as lib:
        lib.startBatch(data, 0)
        cb = lib.finishBatch()
        lib.startBatch(data, 1)
        tb = lib.finishBatch()
        lib.syncWithHost()
    for a, b in zip(final_weights(Lb), final_weights(La)):
        assert np.array_equal(a, b)
    for got, want in ((cb, ca), (tb, ta)):
        assert got[1] == want[1]
        for k in want[0]:
            assert all(abs(x - y) <= 1e-9 * abs(y) + 1e-12 for x, y in zip(got[0][k], want[0][k])), (k, got[0][k], want[0][k])


def small_net(cases, scale=1.0):
    dp, L = build(os.path.join(HERE, 'cfg', 'gc_small.cfg'), os.path.join(HERE, 'cfg', 'gc_small_params.cfg'),
                  'dummy', (12, 12, 3), cases, minibatch=cases, num_classes=10)
    data = dp.get_next_batch()[2]
    return L, [(data[0] * scale).astype(np.float32), data[1]]


def test_gradient_check_small_net():
    """The reference's only self-check (convnet.cu:257-330) run by the native checkGradients: its steps,
    its 0.02 threshold, central differences; same net, data and exemption as tests/test_oracle_pins.py."""
    L, data = small_net(16, scale=8.0)
    with Native(L, 16) as lib:
        res = lib.checkGradients(data, 30, 1)
    assert len(res) == 9  # conv1, local2, fc3 x2, fc4 weights + 4 biases
    for name, an, nn, rel, fail in res:
        if name != 'fc4 weights[0]':
            assert rel < 0.02, (name, an, nn, rel)


def test_training_after_a_gradient_check_is_unaffected():
    """checkGradients leaves the model usable (the reference exits, convnet.cu:255): the PASS_GC backward writes raw gradients
    into the momentum matrices, so they are set aside and restored -- a training step after a check must leave exactly the
    weights the same step leaves without one."""
    outs = []
    for check in (False, True):
        L, data = small_net(16, scale=8.0)
        with Native(L, 16) as lib:
            lib.startBatch(data, 0)   # one step first, so that the momentum matrices are not zero
            lib.finishBatch()
            if check:
                lib.checkGradients(data, 10, 0)
            lib.startBatch(data, 0)
            cost = lib.finishBatch()
            lib.syncWithHost()
        outs.append((L, cost))
    (La, ca), (Lb, cb) = outs
    assert ca == cb
    for la, lb in zip(La, Lb):
        if 'weights' not in la:
            continue
        for key in ('weights', 'weightsInc'):
            for a, b in zip(la[key], lb[key]):
                assert np.array_equal(a, b), (la['name'], key)
        for key in ('biases', 'biasesInc'):
            assert np.array_equal(la[key], lb[key]), (la['name'], key)


def test_directional_gradcheck_through_the_boundary():
    from gradcheck import directional_check
    L, data = small_net(64)
    lib = import_native()

    def train(Lx, d):
        with Native(Lx, d[0].shape[1]) as m:
            m.startBatch(d, 0)
            m.finishBatch()
            m.syncWithHost()

    def cost(Lx, d):
        with Native(Lx, d[0].shape[1]) as m:
            m.startBatch(d, 1)
            return m.finishBatch()[0]

    res = directional_check(L, data, train, cost)
    assert len(res) == 9
    for name, num, anal, rel in res:
        assert rel < 0.02, (name, num, anal, rel)


@pytest.mark.parametrize('name,layer', [('deepid2', 'conv2'), ('deepid2', 'fc5'), ('cifar10', 'conv1')])
def test_a_failed_step_leaves_a_usable_model(name, layer):
    """A step that throws in the middle of its backward pass (layers above have queued work on both streams and threads)
    is reported as an exception, and nothing of it leaks into the next step: the model then behaves exactly like a fresh
    model initialised with the state it holds -- same costs, same weights, bit for bit."""
    dp, L = preset(name, 256)
    data = dp.get_next_batch()[2]
    with Native(L, hooks=True) as lib:  # failNextBackwardAt exists in the test build only
        for _ in range(6 if name == 'cifar10' else 1):  # (cifar10: until some cases are classified correctly, see below)
            lib.startBatch(data, 0)
            lib.finishBatch()
        lib.failNextBackwardAt(layer)
        lib.startBatch(data, 0)
        with pytest.raises(RuntimeError, match='injected failure'):
            lib.finishBatch()
        lib.syncWithHost()              # the state after the failure, as the host sees it
        L2 = copy.deepcopy(L)
        lib.startBatch(data, 0)         # ... and on from there
        cost_a = lib.finishBatch()
        lib.startBatch(data, 1)
        test_a = lib.finishBatch()
        lib.syncWithHost()
    with Native(L2) as lib:
        lib.startBatch(data, 0)
        cost_b = lib.finishBatch()
        lib.startBatch(data, 1)
        test_b = lib.finishBatch()
        lib.syncWithHost()
    assert cost_a == cost_b and test_a == test_b
    assert all(np.isfinite(v).all() for v in cost_a[0].values())
    # the failed pass's forward left per-column tie counts behind (its finishing launch never ran): a step after the
    # recovery must still count whole errors, and on cifar10 some cases must be hits for that to mean anything
    errs = cost_a[0]['logprob'][1]
    assert errs == int(errs), cost_a
    if name == 'cifar10':
        assert errs < 256, 'no correctly classified case: the tie counters are not exercised'

    for la, lb in zip(L, L2):
        if 'weights' not in la:
            continue
        for key in ('weights', 'weightsInc'):
            for a, b in zip(la[key], lb[key]):
                assert np.array_equal(a, b), (la['name'], key)
        for key in ('biases', 'biasesInc'):
            assert np.array_equal(la[key], lb[key]), (la['name'], key)


def test_product_binaries_hold_no_test_hooks():
    """The shared-memory rehearsal communicator and failNextBackwardAt live in libmiconvnet_hooks.so / _ConvNetHooks only."""
    lib = import_native()
    assert not hasattr(lib, 'failNextBackwardAt')
    assert hasattr(import_native(hooks=True), 'failNextBackwardAt')
    from cuda_convnet_amd import capi
    blob = open(capi.LIB_PATH, 'rb').read()
    assert b'MCNSHM' not in blob and b'shm_open' not in blob
    os.environ['MCN_COMM'] = 'shm'
    try:
        with pytest.raises(RuntimeError, match='only in the test build'):
            lib.commUniqueId()
    finally:
        del os.environ['MCN_COMM']


def test_errors_are_exceptions_not_exits():
    lib = import_native()
    with pytest.raises(RuntimeError):
        lib.finishBatch()  # no model
    dp, Lg = preset('cifar10', 128)
    bad = copy.deepcopy(Lg)
    bad[2]['type'] = 'rscale'   # parsed by the reference's layer.py, unknown to its initModel (convnet.cu:86-134)
    with pytest.raises(RuntimeError, match='Unknown layer type'):
        lib.initModel(bad, 128, 0)
    with Native(Lg) as lib:
        with pytest.raises(RuntimeError):
            lib.startBatch([np.zeros((3072, 128), np.float32)], 0)  # labels missing
            lib.finishBatch()
