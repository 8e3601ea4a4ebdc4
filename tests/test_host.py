"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/mi_convnet.h
declares, the `_ConvNet` extension exposes the reference's seven methods (pyconvnet.cu:47-55), the
.cfg parser produces the reference's dict schema (layer.py), and the data-parallel sharding keeps
L2-SN pairs together.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import ccn_loader

ccn = ccn_loader.load()
from cuda_convnet_amd import CFG_DIR, capi, layer  # noqa: E402
from cuda_convnet_amd.convnet import PRESETS  # noqa: E402
from cuda_convnet_amd.data import SyntheticDataProvider, paired_labels, shard_for_rank  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def parse(preset, cases=128, **kw):
    ldef, lpar, kind, img = PRESETS[preset]
    dp = SyntheticDataProvider(kind, cases, img=img)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    return layer.parse_layers(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dims, dp.get_num_classes(), seed=3, **kw)


# ------------------------------------------------------------------------------------ the boundary
def test_c_abi_exports_every_declared_symbol():
    protos = capi.parse_header()
    assert len(protos) >= 60
    dll = ctypes.CDLL(capi.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), 'libmiconvnet.so does not export %s' % name
    # one entry per public function of the reference kernel headers (cudaconv2.cuh:36-78, conv_util.cuh, layer_kernels.cuh)
    for name in ('mcn_conv_filter_acts', 'mcn_local_filter_acts', 'mcn_conv_img_acts', 'mcn_local_img_acts',
                 'mcn_conv_weight_acts', 'mcn_local_weight_acts', 'mcn_local_pool', 'mcn_local_max_undo', 'mcn_local_avg_undo',
                 'mcn_response_norm', 'mcn_response_norm_undo', 'mcn_response_norm_cross_map', 'mcn_response_norm_cross_map_undo',
                 'mcn_logreg_cost', 'mcn_logreg_grad', 'mcn_softmax_grad', 'mcn_logreg_softmax_grad', 'mcn_eltwise_max_grad',
                 'mcn_l2sn_cost', 'mcn_l2sn_grad', 'mcn_l2reg_cost', 'mcn_l2reg_grad', 'mcn_gemm', 'mcn_sgd_update',
                 'mcn_allreduce_sum'):
        assert name in protos


def test_header_has_no_torch_or_cpp_types():
    src = open(capi.HEADER).read()
    assert 'extern "C"' in src
    body = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    for bad in ('torch', 'at::', 'std::', 'hipStream_t ', 'template'):
        assert bad not in body, bad


def test_binding_rejects_bad_arguments_without_a_gpu():
    # argument validation happens before any HIP call, so it can be exercised here
    L = capi.lib()
    rc = L.dll.mcn_conv_filter_acts(None, None, None, 128, 128, 8, 8, 3, 16, 3, 6, 6, 1, 1, 1, ctypes.c_float(0), ctypes.c_float(1), None)
    assert rc == -1  # MCN_EINVAL, with a message instead of the reference's assert()
    assert b'null pointer' in L.dll.mcn_last_error().lower()
    rc = L.dll.mcn_sgd_update(None, None, None, 16, ctypes.c_float(0.9), ctypes.c_float(0), ctypes.c_float(1), 1, None)
    assert rc == -1


def test_extension_exposes_the_reference_methods():
    from cuda_convnet_amd import _ConvNet
    for m in ('initModel', 'startBatch', 'finishBatch', 'checkGradients', 'startMultiviewTest', 'startFeatureWriter', 'syncWithHost'):
        assert callable(getattr(_ConvNet, m)), m
    with pytest.raises(RuntimeError):
        _ConvNet.finishBatch()  # no model yet: an exception, not exit(1)
    with pytest.raises(RuntimeError):
        _ConvNet.startBatch([], 0)


def test_weight_sharing_is_accepted_by_the_graph_builder():
    """`weightSource` (weights.cuh:59-68): the layer graph is built before any device call, so on a machine without a
    GPU initModel gets past the graph and fails at the device (or succeeds on a GPU box) -- never on the sharing itself.
    What a shared net computes: tests/test_weight_sharing.py (oracle), tests/test_weight_sharing_gpu.py (product)."""
    from cuda_convnet_amd import _ConvNet
    from cuda_convnet_amd.data import SyntheticDataProvider
    here = os.path.dirname(os.path.abspath(__file__))
    dp = SyntheticDataProvider('dummy', 32, img=(4, 4, 3), minibatch=32, num_classes=10)
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(os.path.join(here, 'cfg', 'shared.cfg'), os.path.join(here, 'cfg', 'shared_params.cfg'), dims, 10, seed=1)
    fc2 = L[layer.layer_index(L, 'fc2')]
    assert fc2['weightSourceLayerIndices'] == [layer.layer_index(L, 'fc1')] and fc2['weightSourceMatrixIndices'] == [0]
    try:
        _ConvNet.initModel(L, 32, 0)
    except RuntimeError as e:
        assert 'weight' not in str(e).lower(), e
    else:
        _ConvNet.destroyModel()
    bad = [dict(l) for l in L]
    bad[layer.layer_index(L, 'fc2')]['weightSourceMatrixIndices'] = [3]  # fc1 has one matrix
    with pytest.raises(RuntimeError, match='weight source matrix index'):
        _ConvNet.initModel(bad, 32, 0)


def test_checkpoint_loader_admits_arrays_and_nothing_else(tmp_path):
    import pickle
    from cuda_convnet_amd.convnet import ConvNet
    good = {'model_state': {'layers': [{'name': 'fc', 'weights': [np.asfortranarray(np.arange(6, dtype=np.float32).reshape(3, 2))],
                                        'epsW': [0.1], 'neuron': {'type': 'relu', 'params': {}}}], 'epoch': 2, 'batchnum': 7}, 'op': {'seed': 3}}
    p = tmp_path / 'ok'
    p.write_bytes(pickle.dumps(good, protocol=pickle.HIGHEST_PROTOCOL))
    back = ConvNet.load_state(str(p))
    w = back['model_state']['layers'][0]['weights'][0]
    assert w.flags['F_CONTIGUOUS'] and np.array_equal(w, good['model_state']['layers'][0]['weights'][0]) and back['op'] == {'seed': 3}

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))

    q = tmp_path / 'bad'
    q.write_bytes(pickle.dumps({'model_state': Evil()}))
    with pytest.raises(pickle.UnpicklingError):
        ConvNet.load_state(str(q))


# ------------------------------------------------------------------------------------ cfg parser
def test_deepid2_dicts_follow_the_reference_schema():
    L = parse('deepid2')
    by = {l['name']: l for l in L}
    names = [l['name'] for l in L]
    assert names.index('conv1_neuron') == names.index('conv1') + 1  # implicit neuron right after its layer (layer.py:238-244)
    c1 = by['conv1']
    for k in ('name', 'type', 'inputs', 'gradConsumer', 'actsTarget', 'actsGradTarget', 'conserveMem', 'weights', 'weightsInc',
              'biases', 'biasesInc', 'momW', 'momB', 'epsW', 'epsB', 'wc', 'weightSourceLayerIndices', 'weightSourceMatrixIndices',
              'padding', 'stride', 'filterSize', 'channels', 'imgSize', 'filters', 'groups', 'filterChannels', 'randSparse',
              'overSample', 'filterPixels', 'imgPixels', 'modulesX', 'modules', 'filterConns', 'partialSum', 'sharedBiases'):
        assert k in c1, k  # keys read natively: layer.cu:46-54, :261-275, :393-418, :439-440
    assert c1['weights'][0].shape == (48, 16) and c1['weights'][0].flags['C_CONTIGUOUS']
    assert c1['biases'].shape == (16, 1)
    assert c1['padding'] == [0] and c1['modules'] == 2704 and c1['partialSum'] == 1
    l3 = by['local3']
    assert l3['weights'][0].shape == (100 * 288, 48) and l3['biases'].shape == (4800, 1)  # layer.py:958-959
    f5 = by['fc5']
    assert [w.shape for w in f5['weights']] == [(1024, 160), (1200, 160)]
    assert all(w.flags['F_CONTIGUOUS'] and not w.flags['C_CONTIGUOUS'] for w in f5['weights'])  # layer.py:788
    assert by['conv1_neuron']['actsTarget'] == 0 and by['conv1_neuron']['actsGradTarget'] == 0  # in place on the conv acts
    assert by['pool1']['actsTarget'] == -1
    assert by['L2-SN']['m'] == -1 and by['L2-SN']['coeff'] == 0.0025
    assert by['fc6']['epsW'] == [0.0005] and by['fc6']['momW'] == [0.9] and by['fc6']['wc'] == [0.0005]
    assert not by['data']['gradConsumer'] and by['conv1']['gradConsumer']


def test_rectangular_geometry_39x31():
    by = {l['name']: l for l in parse('deepid-39x31')}
    assert (by['conv1']['imgSizeY'], by['conv1']['imgSizeX']) == ([39], [31])
    assert (by['conv1']['modulesY'], by['conv1']['modulesX']) == (36, 28)
    assert (by['pool1']['outputsY'], by['pool1']['outputsX']) == (18, 14)
    assert (by['conv2']['modulesY'], by['conv2']['modulesX']) == (16, 12)
    assert (by['local3']['modulesY'], by['local3']['modulesX']) == (6, 4)
    assert (by['local4']['modulesY'], by['local4']['modulesX']) == (2, 1)
    assert by['fc5']['numInputs'] == [64 * 2 * 1, 48 * 3 * 2]


def test_norm_scale_is_divided_by_window():
    by = {l['name']: l for l in parse('cifar10')}
    r = [l for l in by.values() if l['type'] == 'rnorm'][0]
    c = [l for l in by.values() if l['type'] == 'cmrnorm'][0]
    raw = layer.Cfg(os.path.join(CFG_DIR, PRESETS['cifar10'][1]))
    assert r['scale'] == pytest.approx(raw.get_float(r['name'], 'scale') / r['size'] ** 2)  # layer.py:1033-1037
    assert c['scale'] == pytest.approx(raw.get_float(c['name'], 'scale') / c['size'])


def test_checkpoint_layers_only_get_hyperparameters_refreshed():
    L = parse('deepid2')
    w = L[2]['weights'][0]
    w += 1.0
    L2 = parse('deepid2', layers=L)  # layer.py:276-284
    assert L2[2]['weights'][0] is w


@pytest.mark.parametrize('edit,msg', [
    (('inputs=data', 'inputs=nosuch'), 'not defined'),
    (('neuron=relu', 'neuron=frobnicate'), 'unable to parse neuron'),
    (('type=conv', 'type=rscale'), 'outside the scope'),
    (('filters=16', 'filters=10'), 'divisible by 16'),
    (('partialSum=4', 'partialSum=7'), 'does not divide'),
])
def test_parser_errors(tmp_path, edit, msg):
    src = open(os.path.join(HERE, 'cfg', 'gc_small.cfg')).read()
    assert edit[0] in src
    p = tmp_path / 'bad.cfg'
    p.write_text(src.replace(edit[0], edit[1], 1))
    with pytest.raises(layer.LayerParsingError, match=msg):
        layer.parse_layers(str(p), os.path.join(HERE, 'cfg', 'gc_small_params.cfg'), {0: 432, 1: 1}, 10)


def test_image_op_layers_and_random_sparse_connectivity_parse():
    """The rarely-used layer types: dict schema of layer.py:367-530, :830-853, :1161-1205."""
    by = {l['name']: l for l in parse('cifar10-imgops')}
    assert (by['small']['tgtSize'], by['small']['outputs']) == (24, 3 * 24 * 24)                      # floor(32 / 1.3333)
    assert by['lab']['center'] is True and by['lab']['channels'] == 3 and by['yuv']['outputs'] == 3072
    f = by['blur1']['filter']
    assert f.shape == (1, 5) and f.dtype == np.float32 and abs(float(f.sum()) - 1) < 1e-6 and np.allclose(f, f[:, ::-1])
    n = by['nails']
    assert (n['outputsX'], n['start'], n['outputs']) == (12, 1, 32 * 144)                              # layer.py:385-387
    sh = by['shifted']
    assert sh['outputsX'] == 30 and sh['outputs'] == 30 * 30 * 3 * 4 and sh['filter'].shape == (900, 4)
    assert sh['filter'][31 * 1 + 0, 3] == (1 + 1) * 32 + (1 + 1)                                       # x=1, y=1, dx=dy=1 -> pixel (2, 2)... row x*out+y
    c2 = by['conv2']
    assert c2['randSparse'] == [True] and c2['filterChannels'] == [16] and c2['overSample'] == [2] and c2['filters'] == 64
    conns = c2['filterConns'][0]
    assert len(conns) == 64 and sorted(conns[:32]) == list(range(32)) and sorted(conns[32:]) == list(range(32))
    assert c2['weights'][0].shape == (16 * 9, 64)
    l3 = by['local3']
    assert l3['overSample'] == [1] and sorted(l3['filterConns'][0]) == list(range(64)) and l3['weights'][0].shape == (16 * 32 * 9, 32)
    for name in ('small', 'lab', 'yuv', 'shifted'):   # nothing with parameters below them (layer.py:363-365)
        assert not by[name]['gradConsumer']


@pytest.mark.parametrize('edit,msg', [
    (('inputs=yuv\nchannels=3', 'inputs=conv1\nchannels=32'), 'cannot propagate gradient'),
    (('groups=4\nfilterChannels=16', 'groups=1\nfilterChannels=16'), 'greater than 1 when using random sparse'),
    (('groups=4\nfilterChannels=16', 'groups=4\nfilterChannels=6'), 'divisible'),
    (('filterSize=5\nstdev=1.2', 'filterSize=4\nstdev=1.2'), 'must be one of 3, 5, 7, 9'),
    (('scale=1.3333', 'scale=3'), 'range 0.5-2'),
])
def test_image_op_parser_errors(tmp_path, edit, msg):
    src = open(os.path.join(CFG_DIR, PRESETS['cifar10-imgops'][0])).read()
    assert edit[0] in src
    p = tmp_path / 'bad.cfg'
    p.write_text(src.replace(edit[0], edit[1], 1))
    with pytest.raises(layer.LayerParsingError, match=msg):
        layer.parse_layers(str(p), os.path.join(CFG_DIR, PRESETS['cifar10-imgops'][1]), {0: 3072, 1: 1}, 10)


# ------------------------------------------------------------------------------------ data / sharding
def test_paired_labels_have_both_kinds_of_pairs():
    ids = paired_labels(4, 10575, 512, 128)[0]
    for s in range(0, 512, 128):
        same = ids[s:s + 64] == ids[s + 64:s + 128]
        assert 10 < same.sum() < 54


@pytest.mark.parametrize('world', [2, 4, 8])
def test_shards_keep_l2sn_pairs_together(world):
    mb = 128
    n = mb * world * 2
    data = [np.arange(n, dtype=np.float32).reshape(1, n)]
    seen = []
    for r in range(world):
        (cols,) = shard_for_rank(data, r, world, mb)
        cols = cols[0].astype(int)
        assert cols.size == 2 * mb
        for s in range(0, cols.size, mb):
            g, p = cols[s:s + mb // 2], cols[s + mb // 2:s + mb]
            assert np.all(p - g == mb * world // 2)  # partner of global column j is j + N/2 (layer_kernels.cu:524-527)
        seen.extend(cols)
    assert sorted(seen) == list(range(n))
