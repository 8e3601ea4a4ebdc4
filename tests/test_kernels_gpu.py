"""GPU parity tests: every C-ABI entry point of include/mi_convnet.h against the CPU
oracle on the same seeded inputs, at the shapes of the BASELINE configs
 plus ragged batch sizes (the reference special-cases
N % 128 / 64 / 32, filter_acts.cu:604-608).

Tolerance (fp32, stated per SURVEY 8d): max|a-b| <= 1e-4*max|b| + 1e-6 per tensor;
exact agreement for argmax / correctProbs, max-pool routing and L2-SN decisions.
"""
import ctypes as C

import numpy as np
import pytest

from kutil import call, close, dev, dev_double, dev_empty, ws_bytes, capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu

RAGGED = [1, 33, 64, 96, 128, 130]

# name: (imgY, imgX, C, F, fs, mY, mX, pad, stride, groups, conv)
GEOMS = {
    'conv1_55': (55, 55, 3, 16, 4, 52, 52, 0, 1, 1, True),
    'conv1_39x31': (39, 31, 3, 16, 4, 36, 28, 0, 1, 1, True),
    'conv1_plus': (55, 55, 3, 32, 4, 52, 52, 0, 1, 1, True),
    'conv2': (26, 26, 16, 32, 3, 24, 24, 0, 1, 1, True),
    'conv2_plus': (26, 26, 32, 128, 3, 24, 24, 0, 1, 1, True),
    'local3_plus': (12, 12, 128, 128, 3, 10, 10, 0, 1, 1, False),
    'local4_plus': (5, 5, 128, 128, 2, 4, 4, 0, 1, 1, False),
    'conv_f48': (20, 20, 8, 48, 3, 18, 18, 0, 1, 1, True),
    'conv_f64_pad': (24, 24, 4, 64, 5, 24, 24, -2, 1, 1, True),
    'conv2_39x31': (18, 14, 16, 32, 3, 16, 12, 0, 1, 1, True),
    'local3': (12, 12, 32, 48, 3, 10, 10, 0, 1, 1, False),
    'local3_39x31': (8, 6, 32, 48, 3, 6, 4, 0, 1, 1, False),
    'local4': (5, 5, 48, 64, 2, 4, 4, 0, 1, 1, False),
    'local4_39x31': (3, 2, 48, 64, 2, 2, 1, 0, 1, 1, False),
    'cifar_conv1': (32, 32, 3, 32, 5, 32, 32, -2, 1, 1, True),
    'cifar_conv2': (16, 16, 32, 32, 5, 16, 16, -2, 1, 1, True),
    'cifar_conv3': (8, 8, 32, 64, 5, 8, 8, -2, 1, 1, True),
    'stride2_pad': (13, 11, 8, 16, 3, 7, 6, -1, 2, 1, True),
    'groups2': (10, 10, 16, 32, 3, 8, 8, 0, 1, 2, True),
    'local_groups2_pad': (7, 7, 8, 32, 3, 7, 7, -1, 1, 2, False),
    'odd_k': (9, 9, 1, 16, 3, 7, 7, 0, 1, 1, True),
    'wide_pad_stride': (11, 9, 32, 128, 3, 6, 5, -1, 2, 1, True),   # k_weight_acts_wide with padding checks and stride 2
    'wide_local_pad': (6, 6, 32, 256, 2, 7, 7, -1, 1, 1, False),    # ... KV = 4 (K = 128), two filter blocks, unshared
    'conv_odd_quads': (20, 20, 4, 16, 3, 18, 18, 0, 1, 1, True),    # 81 pool windows: the last strip / workgroup of the pool-fused forward is partial
    'conv_odd_quads_f64': (20, 20, 4, 64, 3, 18, 18, 0, 1, 1, True),
}


def geo_kw(g):
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    return dict(imgY=imgY, imgX=imgX, C_=C_, F=F, fs=fs, mY=mY, mX=mX, pad=pad, stride=stride, groups=groups)


def mk(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def filt_shape(g):
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rows = (C_ // groups) * fs * fs * (1 if conv else mY * mX)
    return rows, F


# ----------------------------------------------------------------------------------- filterActs
@pytest.mark.parametrize('name', list(GEOMS))
def test_filter_acts_shapes(name):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 128
    rng = np.random.default_rng(1)
    img = mk(rng, C_ * imgY * imgX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    want = O.filter_acts(img, W, N, conv=conv, **geo_kw(g))
    d_img, d_W, d_out = dev(img), dev(W), dev_empty(want.shape, np.nan)
    call('mcn_conv_filter_acts' if conv else 'mcn_local_filter_acts', d_img, d_W, d_out, N, N, imgY, imgX, C_, F, fs,
         mY, mX, pad, stride, groups, 0.0, 1.0, None)
    close(d_out.numpy(), want, what=name)


@pytest.mark.parametrize('N', RAGGED)
@pytest.mark.parametrize('name', ['conv1_39x31', 'conv2_39x31', 'local3', 'cifar_conv3'])
def test_filter_acts_ragged(name, N):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(2)
    img = mk(rng, C_ * imgY * imgX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    tgt0 = mk(rng, F * mY * mX, N)
    want = O.filter_acts(img, W, N, conv=conv, targets=tgt0.copy(), sT=0.5, sO=2.0, **geo_kw(g))
    d_out = dev(tgt0)
    call('mcn_conv_filter_acts' if conv else 'mcn_local_filter_acts', dev(img), dev(W), d_out, N, N, imgY, imgX, C_, F,
         fs, mY, mX, pad, stride, groups, 0.5, 2.0, None)
    close(d_out.numpy(), want, what='%s N=%d' % (name, N))


def test_filter_acts_strided_minibatch():
    """images as a column slice of a wider data matrix (data.cu:72-98 without the host copy)."""
    g = GEOMS['conv1_39x31']
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(3)
    big = mk(rng, C_ * imgY * imgX, 512)
    W = mk(rng, *filt_shape(g), scale=0.04)
    d_big = dev(big)
    for start, N in ((128, 128), (384, 128), (256, 100)):
        sl = np.ascontiguousarray(big[:, start:start + N])
        want = O.filter_acts(sl, W, N, conv=True, **geo_kw(g))
        d_out = dev_empty(want.shape, np.nan)
        call('mcn_conv_filter_acts', d_big.offset(4 * start), dev(W), d_out, N, 512, imgY, imgX, C_, F, fs, mY, mX,
             pad, stride, groups, 0.0, 1.0, None)
        close(d_out.numpy(), want, what='slice %d' % start)


@pytest.mark.parametrize('name,per_module', [('conv1_39x31', 0), ('conv2', 0), ('local3', 1), ('cifar_conv1', 1)])
def test_filter_acts_bias_neuron(name, per_module):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N, M = 128, mY * mX
    rng = np.random.default_rng(4)
    img = mk(rng, C_ * imgY * imgX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    b = mk(rng, F * M if per_module else F, scale=0.1)
    want = O.filter_acts(img, W, N, conv=conv, **geo_kw(g))
    if per_module:
        O.add_vector(want, b, F * M, N, True)
    else:
        O.add_vector(want, b, F, M * N, True)
    want = O.neuron_fwd(want, 'relu')
    d_out = dev_empty(want.shape, np.nan)
    call('mcn_filter_acts_bias_neuron', dev(img), dev(W), dev(b), d_out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride,
         groups, int(conv), per_module, 1, 0.0, 0.0, None)
    close(d_out.numpy(), want, what=name)


# the forward launch that also writes the 2 x 2 max pool of its output: strips of 1, 2, 4 and 8 modules (quads shared by 4 / 2 / 1
# waves), filter vectors of 1-4 tiles, ragged image tiles
@pytest.mark.parametrize('name,N', [('conv2', 128), ('conv1_55', 128), ('conv1_55', 256), ('conv1_55', 512), ('conv1_plus', 128),
                                    ('conv2_plus', 128), ('conv_f48', 448), ('conv2_39x31', 100), ('conv1_39x31', 4), ('conv2', 96),
                                    ('conv_odd_quads_f64', 1024), ('conv_odd_quads', 2048), ('conv_odd_quads', 4096), ('groups2', 128)])
@pytest.mark.parametrize('neuron', [1, 0])
def test_filter_acts_with_the_max_pool_in_the_same_launch(name, N, neuron):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(41)
    img = mk(rng, C_ * imgY * imgX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    b = mk(rng, F, scale=0.1)
    d_img, d_W, d_b = dev(img), dev(W), dev(b)
    shape_args = (N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 1, 0, neuron)
    d_ref = dev_empty((F * mY * mX, N), np.nan)
    call('mcn_filter_acts_bias_neuron', d_img, d_W, d_b, d_ref, *shape_args, 0.0, 0.0, None)
    d_pref = dev_empty((F * (mY // 2) * (mX // 2), N), np.nan)
    call('mcn_local_pool', d_ref, d_pref, N, F, mY, mX, 2, 0, 2, mY // 2, mX // 2, 0, None)
    assert capi.lib().raw('mcn_filter_acts_pool_supported')(d_img.ptr, d_W.ptr, d_b.ptr, d_ref.ptr, *shape_args) == 1
    d_out, d_pool = dev_empty((F * mY * mX, N), np.nan), dev_empty((F * (mY // 2) * (mX // 2), N), np.nan)
    call('mcn_filter_acts_bias_neuron_pool', d_img, d_W, d_b, d_out, d_pool, *shape_args, 0.0, 0.0, None)
    ref, pref = d_ref.numpy(), d_pref.numpy()
    assert np.array_equal(d_out.numpy(), ref), name      # the same per-module arithmetic
    assert np.array_equal(d_pool.numpy(), pref), name    # pure selection
    if N <= 128:
        want = O.local_pool(ref, N, F, mY, mX, 2, 0, 2, mY // 2, mX // 2, 'max')
        assert np.array_equal(pref, want), name


@pytest.mark.parametrize('name', ['odd_k', 'cifar_conv1', 'local3', 'stride2_pad'])
def test_filter_acts_with_pool_says_which_shapes_it_takes(name):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 64
    rng = np.random.default_rng(42)
    d_img, d_W = dev(mk(rng, C_ * imgY * imgX, N)), dev(mk(rng, *filt_shape(g)))
    d_b = dev(mk(rng, F * (1 if conv else mY * mX)))
    d_out, d_pool = dev_empty((F * mY * mX, N), 0), dev_empty((F * max(1, mY // 2) * max(1, mX // 2), N), 0)
    shape_args = (N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, int(conv), 0 if conv else 1, 1)
    L = capi.lib()
    assert L.raw('mcn_filter_acts_pool_supported')(d_img.ptr, d_W.ptr, d_b.ptr, d_out.ptr, *shape_args) == 0
    rc = L.raw('mcn_filter_acts_bias_neuron_pool')(d_img.ptr, d_W.ptr, d_b.ptr, d_out.ptr, d_pool.ptr, *shape_args, 0.0, 0.0, None)
    assert rc == -1 and b'pool' in L.raw('mcn_last_error')()


def test_filter_acts_rejects_bad_args():
    L = capi.lib()
    d = dev(np.zeros(16, np.float32))
    rc = L.raw('mcn_conv_filter_acts')(d.ptr, d.ptr, d.ptr, 4, 4, 5, 5, 3, 15, 3, 3, 3, 0, 1, 1, C.c_float(0), C.c_float(1), None)
    assert rc == -1 and b'numFilters' in L.raw('mcn_last_error')()
    rc = L.raw('mcn_conv_filter_acts')(d.ptr, d.ptr, d.ptr, 4, 4, 5, 5, 3, 16, 3, 3, 3, 1, 1, 1, C.c_float(0), C.c_float(1), None)
    assert rc == -1 and b'paddingStart' in L.raw('mcn_last_error')()


# ----------------------------------------------------------------------------------- imgActs
@pytest.mark.parametrize('name', ['conv2', 'conv2_39x31', 'local3', 'local3_39x31', 'local4', 'local4_39x31', 'conv1_39x31',
                                  'cifar_conv2', 'cifar_conv3', 'stride2_pad', 'groups2', 'local_groups2_pad', 'odd_k',
                                  'conv1_plus', 'conv2_plus', 'local3_plus', 'local4_plus'])
def test_img_acts_shapes(name):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 128
    rng = np.random.default_rng(5)
    hid = mk(rng, F * mY * mX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    want = O.img_acts(hid, W, N, conv=conv, **geo_kw(g))
    d_out = dev_empty(want.shape, np.nan)
    call('mcn_conv_img_acts' if conv else 'mcn_local_img_acts', dev(hid), dev(W), d_out, N, imgY, imgX, C_, F, fs, mY, mX,
         pad, stride, groups, 0.0, 1.0, None)
    close(d_out.numpy(), want, what=name)


@pytest.mark.parametrize('N', [128, 96, 4])
@pytest.mark.parametrize('sT', [0.0, 1.0])
@pytest.mark.parametrize('name', ['conv2_plus', 'local3_plus'])
def test_img_acts_narrow_form_of_the_layer_graph(name, sT, N):
    """The layer graph asks for one wave per (pixel, 32 channels, 64 images) on wide layers (`img_acts_wide` = 0); with several
    channel tiles a block is then the channel tiles of ONE pixel (BwdDataParams::waveCt).  Against the oracle at the kernel
    tolerance, and the fused (max-pool undo + relu) epilogue == the unfused chain bit for bit."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(55)
    hid = mk(rng, F * mY * mX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    t0 = mk(rng, C_ * imgY * imgX, N)
    want = O.img_acts(hid, W, N, conv=conv, targets=t0.copy(), sT=sT, sO=1.0, **geo_kw(g))
    cY, cX = 2 * imgY, 2 * imgX
    acts = np.maximum(mk(rng, C_ * cY * cX, N), 0)
    pooled = acts.reshape(C_, imgY, 2, imgX, 2, N).max(axis=(2, 4)).reshape(C_ * imgY * imgX, N)
    call('mcn_set_thread_option', b'img_acts_wide', 0)
    try:
        d_out = dev(t0)
        call('mcn_conv_img_acts' if conv else 'mcn_local_img_acts', dev(hid), dev(W), d_out, N, imgY, imgX, C_, F, fs, mY, mX,
             pad, stride, groups, sT, 1.0, None)
        close(d_out.numpy(), want, what='%s N=%d' % (name, N))
        d_want = dev_empty(acts.shape, np.nan)
        call('mcn_local_max_undo_neuron', dev(acts), d_out, dev(pooled), d_want, N, C_, cY, cX, 2, 0, 2, imgY, imgX, 1, 0.0, 0.0, None)
        d_got = dev_empty(acts.shape, np.nan)
        call('mcn_img_acts_max_undo', dev(hid), dev(W), dev(t0) if sT else None, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups,
             1 if conv else 0, sT, 1.0, dev(acts), dev(pooled), d_got, 1, 0.0, 0.0, None)
        assert np.array_equal(d_got.numpy(), d_want.numpy())
    finally:
        call('mcn_set_thread_option', b'img_acts_wide', 4)


@pytest.mark.parametrize('N', RAGGED)
@pytest.mark.parametrize('name', ['conv2_39x31', 'local4'])
def test_img_acts_ragged_accumulate(name, N):
    """scaleTargets=1: pool3 also receives gradient from fc5 (layer.cu:138)."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(6)
    hid = mk(rng, F * mY * mX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    t0 = mk(rng, C_ * imgY * imgX, N)
    want = O.img_acts(hid, W, N, conv=conv, targets=t0.copy(), sT=1.0, sO=1.0, **geo_kw(g))
    d_out = dev(t0)
    call('mcn_conv_img_acts' if conv else 'mcn_local_img_acts', dev(hid), dev(W), d_out, N, imgY, imgX, C_, F, fs, mY, mX,
         pad, stride, groups, 1.0, 1.0, None)
    close(d_out.numpy(), want, what='%s N=%d' % (name, N))


# ----------------------------------------------------------------------------------- weightActs
@pytest.mark.parametrize('name,ps', [('conv1_55', 1), ('conv1_39x31', 1), ('conv1_39x31', 0), ('conv2', 1), ('conv2_39x31', 4),
                                     ('cifar_conv1', 32), ('cifar_conv3', 0), ('stride2_pad', 1), ('groups2', 8), ('odd_k', 1),
                                     ('local3', 1), ('local3_39x31', 1), ('local4', 1), ('local4_39x31', 1),
                                     ('local_groups2_pad', 1),
                                     ('conv1_plus', 1), ('conv2_plus', 1), ('local3_plus', 1), ('local4_plus', 1)])
def test_weight_acts_shapes(name, ps):
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 128
    rng = np.random.default_rng(7)
    img = mk(rng, C_ * imgY * imgX, N)
    hid = mk(rng, F * mY * mX, N)
    want = O.weight_acts(img, hid, N, partial_sum=ps, conv=conv, sO=0.01, **geo_kw(g))
    d_out = dev_empty(want.shape, np.nan)
    if conv:
        call('mcn_conv_weight_acts', dev(img), dev(hid), d_out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, ps,
             0.0, 0.01, None)
    else:
        call('mcn_local_weight_acts', dev(img), dev(hid), d_out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups,
             0.0, 0.01, None)
    close(d_out.numpy(), want, what=name)


@pytest.mark.parametrize('N', [4, 20, 96, 128, 132])
@pytest.mark.parametrize('name,ps', [('conv2_plus', 8), ('local3_plus', 1), ('local4_plus', 1), ('wide_pad_stride', 3), ('wide_local_pad', 1)])
def test_weight_acts_wide_form_ragged_momentum(name, ps, N):
    """k_weight_acts_wide (>= 128 filters, K a multiple of 96 / 128): one wave walks (module, 16 images) steps of a whole module
    chunk; image counts that are not multiples of 16, chunks of several modules, padding, stride 2, momentum on the target."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(81)
    img = mk(rng, C_ * imgY * imgX, N)
    hid = mk(rng, F * mY * mX, N)
    rows = C_ * fs * fs * (mY * mX // ps if conv else mY * mX)
    t0 = mk(rng, rows, F, scale=0.01)
    want = O.weight_acts(img, hid, N, partial_sum=ps, conv=conv, targets=t0.copy(), sT=0.9, sO=5e-4 / N, **geo_kw(g))
    d_out = dev(t0)
    args = [dev(img), dev(hid), d_out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups]
    if conv:
        args.append(ps)
    call('mcn_conv_weight_acts' if conv else 'mcn_local_weight_acts', *args, 0.9, 5e-4 / N, None)
    close(d_out.numpy(), want, what='%s N=%d' % (name, N))


@pytest.mark.parametrize('N', RAGGED)
@pytest.mark.parametrize('name', ['conv2_39x31', 'local3_39x31'])
def test_weight_acts_ragged_momentum(name, N):
    """local layers write inc = mom*inc + (eps/N)*g directly (layer.cu:551-564)."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(8)
    img = mk(rng, C_ * imgY * imgX, N)
    hid = mk(rng, F * mY * mX, N)
    rows = (C_ // groups) * fs * fs * mY * mX
    t0 = mk(rng, rows, F, scale=0.01)
    want = O.weight_acts(img, hid, N, partial_sum=1, conv=conv, targets=t0.copy(), sT=0.9, sO=5e-4 / N, **geo_kw(g))
    d_out = dev(t0)
    fn = 'mcn_conv_weight_acts' if conv else 'mcn_local_weight_acts'
    args = [dev(img), dev(hid), d_out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups]
    if conv:
        args.append(1)
    call(fn, *args, 0.9, 5e-4 / N, None)
    close(d_out.numpy(), want, what='%s N=%d' % (name, N))


@pytest.mark.parametrize('name', ['conv1_55', 'conv1_39x31', 'conv2', 'conv2_39x31', 'cifar_conv1', 'cifar_conv3',
                                  'conv1_plus', 'conv2_plus'])
@pytest.mark.parametrize('sT', [0.0, 1.0])
def test_conv_weight_acts_reduced(name, sT):
    """ConvLayer::bpropWeights in one call == weightActs(partialSum) + addSum over module chunks."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 128
    rng = np.random.default_rng(9)
    img = mk(rng, C_ * imgY * imgX, N)
    hid = mk(rng, F * mY * mX, N)
    K = (C_ // groups) * fs * fs
    g0 = mk(rng, K, F, scale=0.01)
    tmp = O.weight_acts(img, hid, N, partial_sum=1, conv=True, sO=5e-4 / N, **geo_kw(g))
    want = g0.copy().reshape(-1)
    O.add_sum(want, tmp, mY * mX, K * F, 0, sT, 1.0)
    nbytes = ws_bytes('mcn_conv_weight_acts_reduced_workspace_bytes', C_, F, fs, mY, mX, groups)
    wsb = capi.workspace(nbytes)
    d_out = dev(g0)
    call('mcn_conv_weight_acts_reduced', dev(img), dev(hid), d_out, wsb, nbytes, N, N, imgY, imgX, C_, F, fs, mY, mX, pad,
         stride, groups, sT, 5e-4 / N, None)
    close(d_out.numpy(), want.reshape(K, F), what=name)


# ----------------------------------------------------------------------------------- pooling
POOLS = {  # F, imgY, imgX, size, start, stride, outY, outX
    'pool1': (16, 52, 52, 2, 0, 2, 26, 26),
    'pool1_39x31': (16, 36, 28, 2, 0, 2, 18, 14),
    'pool2': (32, 24, 24, 2, 0, 2, 12, 12),
    'pool3': (48, 10, 10, 2, 0, 2, 5, 5),
    'pool3_39x31': (48, 6, 4, 2, 0, 2, 3, 2),
    'cifar_3_2': (32, 32, 32, 3, 0, 2, 16, 16),
    'odd_clip': (16, 7, 9, 3, -1, 2, 4, 5),
}


@pytest.mark.parametrize('mode', ['max', 'avg'])
@pytest.mark.parametrize('name', list(POOLS))
@pytest.mark.parametrize('N', [128, 33])
def test_local_pool(name, mode, N):
    F, imgY, imgX, size, start, stride, outY, outX = POOLS[name]
    rng = np.random.default_rng(10)
    img = mk(rng, F * imgY * imgX, N)
    want = O.local_pool(img, N, F, imgY, imgX, size, start, stride, outY, outX, mode)
    d_out = dev_empty(want.shape, np.nan)
    call('mcn_local_pool', dev(img), d_out, N, F, imgY, imgX, size, start, stride, outY, outX, 0 if mode == 'max' else 1, None)
    got = d_out.numpy()
    if mode == 'max':
        assert np.array_equal(got, want), name  # pure selection: bit exact
    else:
        close(got, want, what=name)


@pytest.mark.parametrize('name', list(POOLS))
@pytest.mark.parametrize('N,sT', [(128, 0.0), (130, 1.0)])
def test_local_max_undo(name, N, sT):
    F, imgY, imgX, size, start, stride, outY, outX = POOLS[name]
    rng = np.random.default_rng(11)
    # relu'd inputs: plenty of ties at 0 -> exercises "every pixel equal to the max gets the gradient"
    img = np.maximum(mk(rng, F * imgY * imgX, N), 0)
    acts = O.local_pool(img, N, F, imgY, imgX, size, start, stride, outY, outX, 'max')
    grads = mk(rng, *acts.shape)
    t0 = mk(rng, *img.shape)
    want = O.local_max_undo(img, grads, acts, N, F, imgY, imgX, size, start, stride, outY, outX, target=t0.copy(), sT=sT, sO=1.0)
    d_out = dev(t0)
    call('mcn_local_max_undo', dev(img), dev(grads), dev(acts), d_out, N, F, imgY, imgX, size, start, stride, outY, outX,
         sT, 1.0, None)
    got = d_out.numpy()
    assert np.array_equal(got != 0, want != 0), 'max-pool gradient routing differs'
    close(got, want, what=name)


@pytest.mark.parametrize('name', ['pool1_39x31', 'pool3', 'cifar_3_2'])
def test_local_max_undo_neuron(name):
    F, imgY, imgX, size, start, stride, outY, outX = POOLS[name]
    N = 128
    rng = np.random.default_rng(12)
    img = np.maximum(mk(rng, F * imgY * imgX, N), 0)
    acts = O.local_pool(img, N, F, imgY, imgX, size, start, stride, outY, outX, 'max')
    grads = mk(rng, *acts.shape)
    undo = O.local_max_undo(img, grads, acts, N, F, imgY, imgX, size, start, stride, outY, outX)
    want = O.neuron_bwd(undo, img, img, 'relu')
    d_out = dev_empty(img.shape, np.nan)
    call('mcn_local_max_undo_neuron', dev(img), dev(grads), dev(acts), d_out, N, F, imgY, imgX, size, start, stride, outY,
         outX, 1, 0.0, 0.0, None)
    close(d_out.numpy(), want, what=name)


@pytest.mark.parametrize('name', ['pool2', 'cifar_3_2', 'odd_clip'])
@pytest.mark.parametrize('N,sT', [(128, 0.0), (33, 1.0)])
def test_local_avg_undo(name, N, sT):
    F, imgY, imgX, size, start, stride, outY, outX = POOLS[name]
    rng = np.random.default_rng(13)
    grads = mk(rng, F * outY * outX, N)
    t0 = mk(rng, F * imgY * imgX, N)
    want = O.local_avg_undo(grads, N, F, imgY, imgX, size, start, stride, outY, outX, target=t0.copy(), sT=sT, sO=1.0)
    d_out = dev(t0)
    call('mcn_local_avg_undo', dev(grads), d_out, N, F, imgY, imgX, size, start, stride, outY, outX, sT, 1.0, None)
    close(d_out.numpy(), want, what=name)


# ----------------------------------------------------------------------------------- response normalisation
@pytest.mark.parametrize('F,imgY,imgX,size', [(32, 16, 16, 3), (16, 9, 7, 5), (32, 8, 8, 9)])
@pytest.mark.parametrize('N', [128, 33])
def test_response_norm_fwd_bwd(F, imgY, imgX, size, N):
    rng = np.random.default_rng(14)
    x = mk(rng, F * imgY * imgX, N)
    scale, pw = 0.001 / size ** 2 * 50, 0.75
    want, wden = O.response_norm(x, N, F, imgY, imgX, size, scale, pw)
    d_x, d_den, d_y = dev(x), dev_empty(x.shape, np.nan), dev_empty(x.shape, np.nan)
    call('mcn_response_norm', d_x, d_den, d_y, N, F, imgY, imgX, size, scale, pw, None)
    close(d_den.numpy(), wden, what='denoms')
    close(d_y.numpy(), want, what='rnorm')
    og = mk(rng, *x.shape)
    t0 = mk(rng, *x.shape)
    acts = want.copy()
    wt = O.response_norm_undo(og, wden, x, acts, N, F, imgY, imgX, size, scale, pw, target=t0.copy(), sT=1.0, sO=1.0)
    d_acts, d_t = dev(want), dev(t0)
    call('mcn_response_norm_undo', dev(og), dev(wden), d_x, d_acts, d_t, N, F, imgY, imgX, size, scale, pw, 1.0, 1.0, None)
    close(d_acts.numpy(), acts, what='rnorm undo: overwritten acts')
    close(d_t.numpy(), wt, what='rnorm undo')


@pytest.mark.parametrize('F,px,size,blocked', [(32, 64, 9, 0), (64, 16, 5, 0), (32, 30, 4, 1), (16, 25, 16, 0)])
@pytest.mark.parametrize('N', [128, 33])
def test_response_norm_cross_map_fwd_bwd(F, px, size, blocked, N):
    rng = np.random.default_rng(15)
    x = mk(rng, F * px, N)
    scale, pw = 0.001 / size * 50, 0.75
    want, wden = O.response_norm_cross_map(x, N, F, px, size, scale, pw, blocked)
    d_x, d_den, d_y = dev(x), dev_empty(x.shape, np.nan), dev_empty(x.shape, np.nan)
    call('mcn_response_norm_cross_map', d_x, d_den, d_y, N, F, px, size, scale, pw, blocked, None)
    close(d_den.numpy(), wden, what='denoms')
    close(d_y.numpy(), want, what='cmrnorm')
    og = mk(rng, *x.shape)
    acts = want.copy()
    wt = O.response_norm_cross_map_undo(og, wden, x, acts, N, F, px, size, scale, pw, blocked)
    d_acts, d_t = dev(want), dev_empty(x.shape, np.nan)
    call('mcn_response_norm_cross_map_undo', dev(og), dev(wden), d_x, d_acts, d_t, N, F, px, size, scale, pw, blocked,
         0.0, 1.0, None)
    close(d_t.numpy(), wt, what='cmrnorm undo')


# ----------------------------------------------------------------------------------- neurons
NEURON_CASES = [('ident', 0, 0), ('relu', 0, 0), ('logistic', 0, 0), ('brelu', 1.5, 0), ('abs', 0, 0), ('tanh', 1.7159, 0.6667),
                ('softrelu', 0, 0), ('square', 0, 0), ('sqrt', 0, 0), ('linear', 2.0, -0.5)]


@pytest.mark.parametrize('ntype,a,b', NEURON_CASES)
@pytest.mark.parametrize('n', [128 * 97, 1001])
def test_neurons(ntype, a, b, n):
    rng = np.random.default_rng(16)
    x = mk(rng, n, scale=2.0)
    if ntype == 'sqrt':
        x = np.abs(x) + 0.1
    tid = O.NEURON_TYPES[ntype]
    want = O.neuron_fwd(x, ntype, a, b)
    d_x, d_y = dev(x), dev_empty(x.shape, np.nan)
    call('mcn_neuron_fwd', d_x, d_y, n, tid, a, b, None)
    close(d_y.numpy(), want, rtol=2e-6, what=ntype)
    g = mk(rng, n)
    for add in (0, 1):
        t0 = mk(rng, n)
        wg = O.neuron_bwd(g, want, x, ntype, a, b, target=t0.copy(), add=bool(add))
        d_t = dev(t0)
        call('mcn_neuron_bwd', dev(g), dev(want), d_x, d_t, n, tid, a, b, add, None)
        close(d_t.numpy(), wg, rtol=1e-5, what=ntype + ' grad')


def test_relu_in_place():
    rng = np.random.default_rng(17)
    x = mk(rng, 4800, 128)
    d = dev(x)
    call('mcn_neuron_fwd', d, d, x.size, 1, 0.0, 0.0, None)
    assert np.array_equal(d.numpy(), np.maximum(x, 0))


# ----------------------------------------------------------------------------------- GEMM (FC layers)
FC_SHAPES = [(160, 1024), (160, 1200), (4213, 160), (10575, 160), (512, 2048), (10, 1024), (27, 160), (2, 160), (10575, 512), (4131, 176),
             (1000, 96)]


@pytest.mark.parametrize('O_,I', FC_SHAPES)
@pytest.mark.parametrize('N', [128, 33, 72, 256])
def test_fc_gemms(O_, I, N):
    rng = np.random.default_rng(18)
    W = mk(rng, O_, I, scale=0.01)      # Fortran-order (I, O) host matrix == [O][I] in memory (layer.py:788)
    X = mk(rng, I, N)
    dY = mk(rng, O_, N)
    b = mk(rng, O_, scale=0.1)
    L = capi.lib()

    def run(A, B, Cmat, M, Nn, K, ta, tb, alpha, beta):
        nb = ws_bytes('mcn_gemm_workspace_bytes', M, Nn, K, ta, tb)
        wsb = capi.workspace(nb)
        d_c = dev(Cmat)
        call('mcn_gemm', dev(A), dev(B), d_c, M, Nn, K, ta, tb, alpha, beta, wsb, nb, None)
        return d_c.numpy()

    # fwd (layer.cu:355-360)
    y0 = mk(rng, O_, N)
    want = O.gemm(W, X, O_, N, I, beta=1.0, Cmat=y0.copy())
    close(run(W, X, y0, O_, N, I, 0, 0, 1.0, 1.0), want, what='fwd')
    # fused fwd + bias (+relu)
    want = O.gemm(W, X, O_, N, I)
    O.add_vector(want, b, O_, N, True)
    want = O.neuron_fwd(want, 'relu')
    nb = ws_bytes('mcn_gemm_workspace_bytes', O_, N, I, 0, 0)
    wsb = capi.workspace(nb)
    d_c = dev_empty((O_, N), np.nan)
    call('mcn_gemm_bias_neuron', dev(W), dev(X), dev(b), d_c, O_, N, I, 0.0, 1, 0.0, 0.0, wsb, nb, None)
    close(d_c.numpy(), want, what='fwd+bias+relu')
    # bwd-data (layer.cu:362-366)
    dx0 = mk(rng, I, N)
    want = O.gemm(W, dY, I, N, O_, ta=True, beta=1.0, Cmat=dx0.copy())
    close(run(W, dY, dx0, I, N, O_, 1, 0, 1.0, 1.0), want, what='bwd-data')
    # bwd-weights into inc with momentum (layer.cu:374-384)
    inc0 = mk(rng, O_, I, scale=0.01)
    want = O.gemm(dY, X, O_, I, N, tb=True, alpha=5e-4 / N, beta=0.9, Cmat=inc0.copy())
    close(run(dY, X, inc0, O_, I, N, 0, 1, 5e-4 / N, 0.9), want, what='bwd-weights')


@pytest.mark.parametrize('O_,I,N', [(10575, 160, 128), (4213, 160, 72), (10575, 512, 128), (1000, 96, 256), (4131, 160, 4)])
def test_fc_forward_with_softmax_partials(O_, I, N):
    """The classifier layer's forward product leaving stage 1 of the softmax (per 64-row block: max, sum exp(x - max)),
    then stage 2 alone: x and probabilities against the oracle, the logreg gradient against the oracle's, argmax exact."""
    rng = np.random.default_rng(180)
    W = mk(rng, O_, I, scale=0.05)
    X = mk(rng, I, N)
    b = mk(rng, O_, scale=0.1)
    labels = rng.integers(0, O_, (1, N)).astype(np.float32)
    want_x = O.gemm(W, X, O_, N, I)
    O.add_vector(want_x, b, O_, N, True)
    want_p = O.softmax(want_x, O_, N)
    d_W, d_X, d_x = dev(W), dev(X), dev_empty((O_, N), np.nan)
    assert capi.lib().raw('mcn_gemm_softmax_part_supported')(d_W.ptr, d_X.ptr, d_x.ptr, O_, N, I) == 1
    nb = ws_bytes('mcn_softmax_workspace_bytes', O_, N)
    part = capi.workspace(nb)
    call('mcn_gemm_bias_neuron_softmax_part', d_W, d_X, dev(b), d_x, O_, N, I, 0, 0.0, 0.0, part, nb, None)
    close(d_x.numpy(), want_x, what='fc forward')
    d_p, d_g = dev_empty((O_, N), np.nan), dev_empty((O_, N), np.nan)
    call('mcn_softmax_from_part', d_x, d_p, dev(labels), d_g, 0.5, O_, N, part, nb, None)
    got_p = d_p.numpy()
    close(got_p, want_p, what='softmax from partials')
    assert np.array_equal(np.argmax(got_p, axis=0), np.argmax(want_p, axis=0))
    assert np.allclose(got_p.sum(axis=0), 1.0, atol=1e-5)
    want_g = O.logreg_softmax_grad(labels, want_p, O_, N, 0.5)
    close(d_g.numpy(), want_g, what='logreg gradient')
    # the same partials as the stand-alone stage 1 would leave them, up to rounding of the sums
    d_p2 = dev_empty((O_, N), np.nan)
    part2 = capi.workspace(nb)
    call('mcn_softmax', d_x, d_p2, O_, N, part2, nb, None)
    close(d_p2.numpy(), got_p, rtol=1e-5, what='fused vs stand-alone softmax')


def ptr_array(bufs):
    return (C.c_void_p * len(bufs))(*[b.ptr if b is not None else None for b in bufs])


@pytest.mark.parametrize('O_,Is,N', [(160, [1024, 1200], 128), (160, [1024, 1200], 72), (160, [1024, 1200], 33), (512, [2048, 3200], 128),
                                     (27, [160], 128), (160, [16, 32, 48, 160], 130)])
def test_fc_fprop_all_inputs_in_one_launch(O_, Is, N):
    """FCLayer::fpropActs over every input at once (layer.cu:355-360) == the per-input GEMMs of the oracle + bias + neuron."""
    rng = np.random.default_rng(181)
    Ws = [mk(rng, O_, I, scale=0.02) for I in Is]
    Xs = [mk(rng, I, N) for I in Is]
    b = mk(rng, O_, scale=0.1)
    want = O.gemm(Ws[0], Xs[0], O_, N, Is[0])
    for W, X, I in zip(Ws[1:], Xs[1:], Is[1:]):
        want = O.gemm(W, X, O_, N, I, beta=1.0, Cmat=want)
    O.add_vector(want, b, O_, N, True)
    want = O.neuron_fwd(want, 'relu')
    dW, dX = [dev(w) for w in Ws], [dev(x) for x in Xs]
    d_y = dev_empty((O_, N), np.nan)
    wa, xa, ka = ptr_array(dW), ptr_array(dX), (C.c_int * len(Is))(*Is)
    assert capi.lib().raw('mcn_fc_fprop_multi_supported')(wa, xa, ka, len(Is), d_y.ptr, O_, N) == 1
    call('mcn_fc_fprop_multi', wa, xa, ka, len(Is), dev(b), d_y, O_, N, 1, 0.0, 0.0, None)
    close(d_y.numpy(), want, what='fc forward, %d inputs' % len(Is))


@pytest.mark.parametrize('O_,Is,N', [(160, [1024, 1200], 128), (160, [1024, 1200], 72), (512, [2048, 3200], 128), (160, [1024], 2),
                                     (23, [160, 30], 128)])
def test_fc_bprop_acts_all_inputs_in_one_launch(O_, Is, N):
    """FCLayer::bpropActs for every input at once (layer.cu:362-366); input 0 additionally through the gradient of the
    relu that produced it (NeuronLayer::bpropActs layer.cu:245), the last input accumulating (scaleTargets = 1)."""
    rng = np.random.default_rng(182)
    Ws = [mk(rng, O_, I, scale=0.02) for I in Is]
    dY = mk(rng, O_, N)
    acts0 = np.maximum(mk(rng, Is[0], N), 0)
    t0 = [mk(rng, I, N) for I in Is]
    betas = [0.0] * len(Is)
    betas[-1] = 1.0 if len(Is) > 1 else 0.0
    wants = []
    for i, (W, I) in enumerate(zip(Ws, Is)):
        g = O.gemm(W, dY, I, N, O_, ta=True)
        if i == 0:
            g = O.neuron_bwd(g, acts0, acts0, 'relu')
        wants.append((np.float32(betas[i]) * t0[i] + g).astype(np.float32))
    dW, dT = [dev(w) for w in Ws], [dev(t) for t in t0]
    d_a = dev(acts0)
    wa, ta_, ka = ptr_array(dW), ptr_array(dT), (C.c_int * len(Is))(*Is)
    na = ptr_array([d_a] + [None] * (len(Is) - 1))
    nt = (C.c_int * len(Is))(*([1] + [0] * (len(Is) - 1)))
    zf = (C.c_float * len(Is))(*([0.0] * len(Is)))
    d_dy = dev(dY)
    assert capi.lib().raw('mcn_fc_bprop_acts_multi_supported')(wa, ta_, ka, len(Is), d_dy.ptr, O_, N) == 1
    call('mcn_fc_bprop_acts_multi', wa, ta_, ka, (C.c_float * len(Is))(*betas), na, nt, zf, zf, len(Is), d_dy, O_, N, None)
    for i in range(len(Is)):
        close(dT[i].numpy(), wants[i], what='dX[%d]' % i)


def test_gemm_odd_sizes():
    rng = np.random.default_rng(19)
    for M, N, K, ta, tb in [(37, 45, 51, 0, 0), (37, 45, 51, 1, 0), (37, 45, 51, 0, 1), (1, 1, 1, 0, 0), (300, 130, 2049, 0, 0)]:
        A = mk(rng, *((K, M) if ta else (M, K)))
        B = mk(rng, *((N, K) if tb else (K, N)))
        want = O.gemm(A, B, M, N, K, ta=bool(ta), tb=bool(tb), alpha=0.5)
        nb = ws_bytes('mcn_gemm_workspace_bytes', M, N, K, ta, tb)
        wsb = capi.workspace(nb)
        d_c = dev_empty((M, N), np.nan)
        call('mcn_gemm', dev(A), dev(B), d_c, M, N, K, ta, tb, 0.5, 0.0, wsb, nb, None)
        close(d_c.numpy(), want, what=str((M, N, K, ta, tb)))


# ----------------------------------------------------------------------------------- NVMatrix verbs
@pytest.mark.parametrize('rows,cols', [(16, 36 * 28 * 128), (16, 52 * 52 * 128), (4800, 128), (10575, 128), (160, 33), (3, 100000)])
def test_add_vector_and_add_sum(rows, cols):
    rng = np.random.default_rng(20)
    mat = mk(rng, rows, cols)
    for per_row in (1, 0):
        vec = mk(rng, rows if per_row else cols)
        want = O.add_vector(mat.copy(), vec, rows, cols, bool(per_row), 0.5)
        d = dev(mat)
        call('mcn_add_vector', d, dev(vec), rows, cols, per_row, 0.5, None)
        close(d.numpy(), want, what='addVector')
    for axis in (1, 0):
        n_out = rows if axis == 1 else cols
        t0 = mk(rng, n_out)
        for sthis in (0.0, 0.9):
            want = O.add_sum(t0.copy(), mat, rows, cols, axis, sthis, 5e-4 / 128)
            nb = ws_bytes('mcn_add_sum_workspace_bytes', rows, cols, axis)
            wsb = capi.workspace(nb)
            d_t = dev(t0)
            call('mcn_add_sum', d_t, dev(mat), rows, cols, axis, sthis, 5e-4 / 128, wsb, nb, None)
            # long fp32 sums: error grows with sqrt(len); tolerance relative to the sum of |terms|
            bound = 2e-6 * np.abs(mat).sum(axis=axis).max() * 5e-4 / 128 + 1e-9
            assert np.max(np.abs(d_t.numpy().astype(np.float64) - want)) <= max(bound, 1e-4 * np.abs(want).max() + 1e-6)


def test_eltwise_verbs():
    rng = np.random.default_rng(21)
    for n in (128 * 1200, 1003):
        a, b, g = mk(rng, n), mk(rng, n), mk(rng, n)
        d_a = dev(a)
        call('mcn_axpby', d_a, dev(b), n, 0.9, -0.25, None)
        close(d_a.numpy(), O.axpby(a.copy(), b, 0.9, -0.25), what='axpby')
        d_o = dev_empty(n, np.nan)
        call('mcn_scale', dev(a), d_o, n, 1.5, None)
        close(d_o.numpy(), a * np.float32(1.5), what='scale')
        call('mcn_eltwise_max', dev(a), dev(b), d_o, n, None)
        mx = np.maximum(a, b)
        assert np.array_equal(d_o.numpy(), mx)
        t0 = mk(rng, n)
        d_t = dev(t0)
        call('mcn_eltwise_max_grad', dev(g), dev(a), dev(mx), d_t, n, 1, None)
        close(d_t.numpy(), t0 + g * (mx == a), what='eltmax grad')
        d_s = dev_double(1)
        call('mcn_sum_all', dev(a), n, d_s, 0, None)
        call('mcn_sum_all', dev(b), n, d_s, 1, None)
        got = d_s.numpy()[0]
        assert abs(got - (O.sum_all(a) + O.sum_all(b))) <= 1e-9 * np.abs(a).sum() + 1e-9


@pytest.mark.parametrize('rows,cols', [(160, 128), (10575, 33), (31, 77)])
def test_transpose(rows, cols):
    rng = np.random.default_rng(22)
    a = mk(rng, rows, cols)
    d_o = dev_empty((cols, rows), np.nan)
    call('mcn_transpose', dev(a), d_o, rows, cols, None)
    assert np.array_equal(d_o.numpy(), a.T)


# ----------------------------------------------------------------------------------- softmax / logreg
@pytest.mark.parametrize('K', [10575, 4213, 4131, 10, 257])
@pytest.mark.parametrize('N', [128, 33, 1])
def test_softmax_logreg(K, N):
    rng = np.random.default_rng(23)
    x = mk(rng, K, N, scale=3.0)
    labels = rng.integers(0, K, size=(1, N)).astype(np.float32)
    want = O.softmax(x, K, N)
    nb = ws_bytes('mcn_softmax_workspace_bytes', K, N)
    wsb = capi.workspace(nb)
    d_p = dev_empty((K, N), np.nan)
    call('mcn_softmax', dev(x), d_p, K, N, wsb, nb, None)
    got = d_p.numpy()
    close(got, want, what='softmax')
    assert np.array_equal(got.argmax(0), want.argmax(0))
    np.testing.assert_allclose(got.sum(0), 1.0, atol=5e-5)

    # cost on the ORACLE's probabilities so that tie decisions see identical inputs
    probs = want.copy()
    if N > 4:  # force ties and label==argmax cases (layer_kernels.cu:82-90)
        k0 = int(labels[0, 0])
        probs[:, 0] = 0.0
        probs[k0, 0] = 0.5
        probs[(k0 + 1) % K, 0] = 0.5
        k1 = int(labels[0, 1])
        probs[:, 1] = 1e-3
        probs[k1, 1] = 0.9
        probs[:, 2] = 1.0 / K  # all tied
    wlp, wcp = O.logreg_cost(labels, probs, K, N)
    nb = ws_bytes('mcn_logreg_cost_workspace_bytes', K, N)
    wsb = capi.workspace(nb)
    d_lp, d_cp, d_cost = dev_empty((1, N), np.nan), dev_empty((1, N), np.nan), dev_double(2)
    call('mcn_logreg_cost', dev(labels), dev(probs), d_lp, d_cp, K, N, d_cost, wsb, nb, None)
    assert np.array_equal(d_cp.numpy(), wcp), 'correctProbs must agree exactly'
    close(d_lp.numpy(), wlp, rtol=2e-6, what='logprob')
    cost = d_cost.numpy()
    assert abs(cost[0] + O.sum_all(wlp)) <= 1e-5 * abs(O.sum_all(wlp)) + 1e-6
    assert abs(cost[1] - (N - O.sum_all(wcp))) <= 1e-6

    for add in (0, 1):
        t0 = mk(rng, K, N)
        wg = O.logreg_softmax_grad(labels, probs, K, N, 1.0, target=t0.copy(), add=bool(add))
        d_t = dev(t0)
        call('mcn_logreg_softmax_grad', dev(labels), dev(probs), d_t, K, N, add, 1.0, None)
        close(d_t.numpy(), wg, what='logreg-softmax grad')
    wg = O.logreg_grad(labels, want, K, N, 0.5)
    d_t = dev_empty((K, N), np.nan)
    call('mcn_logreg_grad', dev(labels), dev(want), d_t, K, N, 0, 0.5, None)
    close(d_t.numpy(), wg, rtol=1e-5, what='logreg grad')


def test_softmax_grad_small():
    rng = np.random.default_rng(24)
    K, N = 37, 33
    acts = O.softmax(mk(rng, K, N), K, N)
    g = mk(rng, K, N)
    want = O.softmax_grad(acts, g, K, N)
    d_t = dev_empty((K, N), np.nan)
    call('mcn_softmax_grad', dev(acts), dev(g), d_t, K, N, 0, None)
    close(d_t.numpy(), want, rtol=1e-4, atol=1e-6, what='softmax grad')


# ----------------------------------------------------------------------------------- L2-SN (Joint identification-verification)
def paired_labels(rng, N, K, p_same=0.5):
    h = N // 2
    ids = rng.integers(0, K, size=N)
    same = rng.random(h) < p_same
    ids[h:][same] = ids[:h][same]
    return ids.astype(np.float32).reshape(1, N)


@pytest.mark.parametrize('D,N', [(160, 128), (512, 128), (160, 256), (160, 2), (23, 34)])
@pytest.mark.parametrize('m', [-1.0, 3.0])
def test_l2sn(D, N, m):
    rng = np.random.default_rng(25)
    data = mk(rng, D, N, scale=0.3)
    ids = paired_labels(rng, N, 10575)
    h = N // 2
    wout, wm, widx, wd, wl = O.l2sn_cost(ids, data, D, N, m)
    d_out, d_info, d_stats, d_cost = dev_empty((1, h), np.nan), dev_empty(4, np.nan), dev_empty(2 * h, np.nan), dev_double(1)
    call('mcn_l2sn_cost', dev(ids), dev(data), d_out, D, N, m, d_info, d_stats, d_cost, None, None, 0, None)
    stats = d_stats.numpy()
    assert np.array_equal(stats[:h], wd), 'pair distances must be bit-identical (shared reduction order)'
    assert np.array_equal(stats[h:], wl)
    info = d_info.numpy()
    assert info[0] == np.float32(wm) and int(info[1]) == widx, 'margin / chosen pair must agree exactly'
    assert np.array_equal(d_out.numpy() != 0, wout != 0), '1[d<m] decisions must agree exactly'
    close(d_out.numpy(), wout, rtol=1e-6, what='l2sn cost')
    assert abs(d_cost.numpy()[0] - O.sum_all(wout)) <= 1e-6 * abs(O.sum_all(wout)) + 1e-9
    for add in (0, 1):
        t0 = mk(rng, D, N, scale=0.01)
        wg = O.l2sn_grad(ids, data, D, N, 0.0025, m, target=t0.copy(), add=bool(add))
        d_t = dev(t0)
        call('mcn_l2sn_grad', dev(ids), dev(data), d_t, D, N, m, add, 0.0025, None, None, 0, None)
        close(d_t.numpy(), wg, rtol=1e-6, atol=1e-9, what='l2sn grad')
        d_t2 = dev(t0)
        call('mcn_l2sn_grad_from_stats', dev(data), d_t2, D, N, d_stats, d_info, add, 0.0025, None)
        assert np.array_equal(d_t2.numpy(), d_t.numpy()), 'gradient from kept statistics must equal the recomputing kernel bit for bit'
    # the two-launch form the layer graph uses: statistics on one workgroup per 8 pairs, then margin search + cost from them
    d_stats2 = dev_empty(2 * h, np.nan)
    call('mcn_l2sn_pair_stats', dev(ids), dev(data), d_stats2, D, N, None)
    assert np.array_equal(d_stats2.numpy(), stats), 'windowed pair statistics must equal the one-workgroup kernel bit for bit'
    d_out2, d_info2, d_cost2 = dev_empty((1, h), np.nan), dev_empty(4, np.nan), dev_double(1)
    call('mcn_l2sn_cost_from_stats', d_stats2, d_out2, N, m, d_info2, d_cost2, None, None, 0, None)
    assert np.array_equal(d_out2.numpy(), d_out.numpy()) and np.array_equal(d_info2.numpy(), info)
    assert d_cost2.numpy()[0] == d_cost.numpy()[0]


def test_l2sn_external_candidates():
    """Multi-GPU: the margin is searched over the all-gathered pairs of every rank."""
    rng = np.random.default_rng(26)
    D, N, G = 160, 128, 4
    h = N // 2
    datas = [mk(rng, D, N, scale=0.3) for _ in range(G)]
    idss = [paired_labels(rng, N, 10575) for _ in range(G)]
    # the equivalent single-GPU minibatch: galleries of all ranks, then probes of all ranks
    data_g = np.concatenate([d[:, :h] for d in datas] + [d[:, h:] for d in datas], axis=1)
    ids_g = np.concatenate([i[:, :h] for i in idss] + [i[:, h:] for i in idss], axis=1)
    wout, wm, widx, wd, wl = O.l2sn_cost(ids_g, np.ascontiguousarray(data_g), D, N * G, -1.0)
    stats = []
    for r in range(G):
        d_stats = dev_empty(2 * h, np.nan)
        call('mcn_l2sn_pair_stats', dev(idss[r]), dev(datas[r]), d_stats, D, N, None)
        stats.append(d_stats.numpy())
    ext_d = np.concatenate([s[:h] for s in stats])
    ext_l = np.concatenate([s[h:] for s in stats])
    assert np.array_equal(ext_d, wd) and np.array_equal(ext_l, wl)
    wg = O.l2sn_grad(ids_g, np.ascontiguousarray(data_g), D, N * G, 0.0025, -1.0)
    for r in range(G):
        d_out, d_info = dev_empty((1, h), np.nan), dev_empty(4, np.nan)
        call('mcn_l2sn_cost', dev(idss[r]), dev(datas[r]), d_out, D, N, -1.0, d_info, None, None, dev(ext_d), dev(ext_l), G * h, None)
        info = d_info.numpy()
        assert info[0] == np.float32(wm) and int(info[1]) == widx
        close(d_out.numpy().reshape(-1), wout.reshape(-1)[r * h:(r + 1) * h], rtol=1e-6, what='l2sn cost rank %d' % r)
        d_out2, d_info2 = dev_empty((1, h), np.nan), dev_empty(4, np.nan)
        call('mcn_l2sn_cost_from_stats', dev(stats[r]), d_out2, N, -1.0, d_info2, None, dev(ext_d), dev(ext_l), G * h, None)
        assert np.array_equal(d_out2.numpy(), d_out.numpy()) and np.array_equal(d_info2.numpy(), info)
        d_t = dev_empty((D, N), np.nan)
        call('mcn_l2sn_grad', dev(idss[r]), dev(datas[r]), d_t, D, N, -1.0, 0, 0.0025, dev(ext_d), dev(ext_l), G * h, None)
        got = d_t.numpy()
        close(got[:, :h], wg[:, r * h:(r + 1) * h], rtol=1e-6, atol=1e-9, what='gallery grad rank %d' % r)
        close(got[:, h:], wg[:, G * h + r * h:G * h + (r + 1) * h], rtol=1e-6, atol=1e-9, what='probe grad rank %d' % r)


@pytest.mark.parametrize('D', [2, 27, 23])
def test_l2reg(D):
    rng = np.random.default_rng(27)
    N = 128
    gt, pred = mk(rng, D, N), mk(rng, D, N)
    want = O.l2reg_cost(gt, pred, D, N)
    d_o, d_cost = dev_empty((1, N), np.nan), dev_double(1)
    call('mcn_l2reg_cost', dev(gt), dev(pred), d_o, D, N, d_cost, None)
    close(d_o.numpy(), want, rtol=1e-6, what='l2reg cost')
    assert abs(d_cost.numpy()[0] - O.sum_all(want)) <= 1e-6 * O.sum_all(want)
    t0 = mk(rng, D, N)
    wg = O.l2reg_grad(gt, pred, D, N, 0.01, target=t0.copy(), add=True)
    d_t = dev(t0)
    call('mcn_l2reg_grad', dev(gt), dev(pred), d_t, D, N, 1, 0.01, None)
    close(d_t.numpy(), wg, rtol=1e-6, what='l2reg grad')


# ----------------------------------------------------------------------------------- SGD update
@pytest.mark.parametrize('n', [768, 28800 * 48, 1001])
@pytest.mark.parametrize('use_grad,wc', [(1, 5e-4), (0, 5e-4), (1, 0.0)])
def test_sgd_update(n, use_grad, wc):
    rng = np.random.default_rng(28)
    W, inc, grad = mk(rng, n, scale=0.04), mk(rng, n, scale=1e-3), mk(rng, n, scale=1e-4)
    wW, wInc = W.copy(), inc.copy()
    O.sgd_update(wW, wInc, grad, 0.9, wc, 5e-4, bool(use_grad))
    d_W, d_inc = dev(W), dev(inc)
    call('mcn_sgd_update', d_W, d_inc, dev(grad), n, 0.9, wc, 5e-4, use_grad, None)
    close(d_inc.numpy(), wInc, rtol=1e-6, atol=1e-12, what='inc')
    close(d_W.numpy(), wW, rtol=1e-6, what='W')


def test_sgd_update_planned_matches_per_matrix_update():
    """Every matrix in one launch (mcn_sgd_update_planned) == the oracle's per-matrix Weights::update."""
    class Item(C.Structure):
        _fields_ = [('W', C.c_void_p), ('inc', C.c_void_p), ('grad', C.c_void_p), ('n', C.c_size_t),
                    ('mom', C.c_float), ('wc', C.c_float), ('eps', C.c_float), ('useGrad', C.c_int)]
    rng = np.random.default_rng(11)
    specs = [(768, 1, 5e-4), (28800 * 48, 0, 5e-4), (1001, 1, 0.0), (3, 1, 0.0), (2049, 0, 0.0), (4096, 1, 5e-4)]
    host, devs, items = [], [], (Item * len(specs))()
    for i, (n, ug, wc) in enumerate(specs):
        W, inc, g = mk(rng, n), mk(rng, n, scale=0.01), mk(rng, n, scale=0.001)
        dW, dI, dG = dev(W), dev(inc), dev(g)
        devs.append((dW, dI, dG))
        host.append((W, inc, g))
        items[i] = Item(dW.ptr, dI.ptr, dG.ptr if ug else None, n, 0.9, wc, 5e-4, ug)
    L = capi.lib()
    plan = capi.workspace(ws_bytes('mcn_sgd_plan_bytes', len(specs)))
    nb = C.c_int(0)
    L.call('mcn_sgd_plan_build', C.cast(items, C.c_void_p), len(specs), plan, plan.nbytes, C.byref(nb), None)
    assert nb.value == sum((n + 2047) // 2048 for n, _, _ in specs)
    for _ in range(2):  # two consecutive steps
        L.call('mcn_sgd_update_planned', plan, len(specs), nb.value, None)
        for (W, inc, g), (n, ug, wc) in zip(host, specs):
            O.sgd_update(W, inc, g, 0.9, wc, 5e-4, ug)
    for (W, inc, g), (dW, dI, dG), spec in zip(host, devs, specs):
        close(dW.numpy(), W, rtol=1e-6, what='W %r' % (spec,))
        close(dI.numpy(), inc, rtol=1e-6, what='inc %r' % (spec,))


@pytest.mark.parametrize('rows,cases,mb', [(7, 1000, 128), (1, 64, 128), (3, 130, 34), (9075, 300, 128), (5, 256, 128)])
def test_tile_minibatches(rows, cases, mb):
    """Every column slice of data.cu:72-98 at once: slice i is dense (rows, n_i) at offset i*mb*rows."""
    rng = np.random.default_rng(40)
    src = rng.standard_normal((rows, cases)).astype(np.float32)
    d_dst = dev_empty((rows * cases,), np.nan)
    call('mcn_tile_minibatches', dev(src), d_dst, rows, cases, mb, None)
    got = d_dst.numpy()
    for i in range(0, cases, mb):
        n = min(mb, cases - i)
        assert np.array_equal(got[i * rows:i * rows + rows * n].reshape(rows, n), src[:, i:i + n])


@pytest.mark.parametrize('D,N,classes', [(160, 128, 40), (12, 16, 4), (33, 50, 7), (160, 128, 10575)])
def test_fisher2(D, N, classes):
    """cost.fisher2 (layer_kernels.cu:310-479) against the step-by-step CPU restatement."""
    rng = np.random.default_rng(41)
    data = mk(rng, D, N, scale=0.3)
    labels = rng.integers(0, classes, N).astype(np.float32).reshape(1, N)
    if classes > N:  # nearly all identities distinct: make sure some pairs match
        labels[0, N // 2:N // 2 + 8] = labels[0, :8]
    alpha = 0.3
    want = O.fisher2_cost(labels, data, D, N, alpha)
    nbytes = call('mcn_fisher2_workspace_bytes', N)
    ws = dev_empty((nbytes // 4,), np.nan)
    d_cost = dev_double(1)
    call('mcn_fisher2_cost', dev(labels), dev(data), D, N, alpha, d_cost, ws, nbytes, None)
    got = d_cost.numpy()[0]
    assert abs(got - want * N) <= 1e-4 * abs(want * N) + 1e-9, (got, want * N)
    for add in (0, 1):
        t0 = mk(rng, D, N, scale=0.01)
        wg = O.fisher2_grad(labels, data, D, N, alpha, 0.5, target=t0.copy(), add=bool(add))
        d_t = dev(t0)
        call('mcn_fisher2_grad', dev(labels), dev(data), d_t, D, N, alpha, add, 0.5, ws, nbytes, None)
        close(d_t.numpy(), wg, rtol=1e-4, atol=1e-7, what='fisher2 grad')
    with pytest.raises(Exception):
        call('mcn_fisher2_cost', dev(labels), dev(data), D, N, alpha, d_cost, ws, 16, None)


@pytest.mark.parametrize('F,img,size,N', [(32, 16, 3, 128), (16, 9, 5, 33), (3, 12, 4, 64)])
def test_contrast_norm(F, img, size, N):
    """convContrastNorm / convContrastNormUndo (conv_util.cu:2074-2160, :2214-2217) as ContrastNormLayer uses them
    (layer.cu:897-906): mean over the clipped window, images - mean, normalise, undo with inputs = meanDiffs."""
    rng = np.random.default_rng(42)
    x = mk(rng, F * img * img, N, scale=2.0)
    scale, pw = 0.001 / size ** 2, 0.75
    mean = O.local_pool(x, N, F, img, img, size, -(size // 2), 1, img, img, 'avg')
    md = (x - mean).astype(np.float32)
    want, wden = O.contrast_norm(x, md, N, F, img, img, size, scale, pw)
    d_mean, d_den, d_out = dev_empty(x.shape, np.nan), dev_empty(x.shape, np.nan), dev_empty(x.shape, np.nan)
    call('mcn_local_pool', dev(x), d_mean, N, F, img, img, size, -(size // 2), 1, img, img, 1, None)
    call('mcn_axpby', d_mean, dev(x), x.size, -1.0, 1.0, None)
    close(d_mean.numpy(), md, what='mean diffs')
    call('mcn_contrast_norm', dev(x), d_mean, d_den, d_out, N, F, img, img, size, scale, pw, None)
    close(d_den.numpy(), wden, what='cnorm denoms')
    close(d_out.numpy(), want, what='cnorm')
    if F % 16 == 0:  # conv_util.cu:2246
        g = mk(rng, F * img * img, N)
        wt = O.response_norm_undo(g, wden, md, want.copy(), N, F, img, img, size, scale, pw)
        d_t = dev_empty(x.shape, np.nan)
        call('mcn_response_norm_undo', dev(g), d_den, d_mean, d_out, d_t, N, F, img, img, size, scale, pw, 0.0, 1.0, None)
        close(d_t.numpy(), wt, what='cnorm undo')


def test_softmax_with_fused_logreg_grad():
    """mcn_softmax_logreg_grad == mcn_softmax followed by mcn_logreg_softmax_grad, bit for bit."""
    rng = np.random.default_rng(43)
    K, N = 1000, 96
    x = mk(rng, K, N, scale=2.0)
    labels = rng.integers(0, K, (1, N)).astype(np.float32)
    nb = ws_bytes('mcn_softmax_workspace_bytes', K, N)
    ws = capi.workspace(nb)
    p1, p2, g1, g2 = (dev_empty((K, N), np.nan) for _ in range(4))
    call('mcn_softmax', dev(x), p1, K, N, ws, nb, None)
    call('mcn_logreg_softmax_grad', dev(labels), p1, g1, K, N, 0, 0.7, None)
    call('mcn_softmax_logreg_grad', dev(x), p2, dev(labels), g2, 0.7, K, N, ws, nb, None)
    assert np.array_equal(p1.numpy(), p2.numpy())
    assert np.array_equal(g1.numpy(), g2.numpy())


@pytest.mark.parametrize('K,N', [(10575, 128), (4213, 96), (257, 4), (10, 260)])
def test_softmax_with_fused_logreg_cost(K, N):
    """mcn_softmax_logreg_cost == mcn_softmax_logreg_grad followed by mcn_logreg_cost: probabilities, gradient, log p[label]
    and correctProbs bit for bit (ties at the maximum included), the two cost scalars to the last digits of a double --
    on two calls in a row over the same state buffer, the second with other data."""
    rng = np.random.default_rng(44)
    nb = ws_bytes('mcn_softmax_workspace_bytes', K, N)
    ws = capi.workspace(nb)
    sb = ws_bytes('mcn_softmax_logreg_cost_state_bytes', N)
    state = capi.workspace(sb)
    call('mcn_memset', state, 0, sb, None)
    assert capi.lib().raw('mcn_softmax_logreg_cost_supported')(K, N) == 1
    for rep in range(2):
        x = mk(rng, K, N, scale=2.0)
        labels = rng.integers(0, K, (1, N)).astype(np.float32)
        k0, k1 = int(labels[0, 0]), int(labels[0, 1])
        x[:, 0] = -1.0          # two maxima, the label one of them: correctProbs 0.5
        x[k0, 0] = x[(k0 + 3) % K, 0] = 2.5
        x[:, 1] = 0.25          # every entry tied: 1/K
        x[(k1 + 1) % K, 2] = 9.0  # a clear maximum (that is the label only by chance)
        x[int(labels[0, 3]), 3] = 11.0  # the label is the single maximum
        p1, p2, g1, g2 = (dev_empty((K, N), np.nan) for _ in range(4))
        lp1, cp1, lp2, cp2 = (dev_empty((1, N), np.nan) for _ in range(4))
        c1, c2 = dev_double(2), dev_double(2)
        call('mcn_softmax_logreg_grad', dev(x), p1, dev(labels), g1, 0.7, K, N, ws, nb, None)
        call('mcn_logreg_cost', dev(labels), p1, lp1, cp1, K, N, c1, ws, nb, None)
        call('mcn_softmax_logreg_cost', dev(x), p2, dev(labels), g2, 0.7, K, N, None, 0, ws, nb, lp2, cp2, c2, state, sb, rep, None)
        if rep == 0:  # the one-block launch on its own
            call('mcn_softmax_logreg_cost_finish', lp2, cp2, c2, state, sb, N, None)
        assert np.array_equal(p1.numpy(), p2.numpy())
        assert np.array_equal(g1.numpy(), g2.numpy())
        assert np.array_equal(lp1.numpy(), lp2.numpy())
        assert np.array_equal(cp1.numpy(), cp2.numpy())
        cp = cp2.numpy()[0]
        assert cp[0] == 0.5 and cp[1] == np.float32(1.0) / np.float32(K) and cp[3] == 1.0
        a, b = c1.numpy(), c2.numpy()
        assert np.all(np.abs(a - b) <= 1e-12 * np.abs(a)), (a, b)
        # against the oracle, on the probabilities the device produced
        wlp, wcp = O.logreg_cost(labels, p2.numpy(), K, N)
        assert np.array_equal(cp2.numpy(), wcp)
        close(lp2.numpy(), wlp, rtol=2e-6, what='logprob')


@pytest.mark.parametrize('N', [128, 96, 4])
@pytest.mark.parametrize('sT', [0.0, 1.0])
@pytest.mark.parametrize('name', ['conv2', 'conv2_39x31', 'local3', 'local3_39x31', 'local4', 'local4_39x31', 'conv2_plus', 'local3_plus',
                                  'local4_plus', 'groups2'])
def test_img_acts_through_max_pool_and_neuron(name, sT, N):
    """mcn_img_acts_max_undo == imgActs into the pool's gradient matrix, then mcn_local_max_undo_neuron, bit for bit (relu under a
    2x2 / stride 2 pool whose windows tile the maps: the DeepID stacks), with and without another consumer's share already in
    the pool's gradient; the unfused chain itself is checked against the oracle."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    rng = np.random.default_rng(45)
    hid = mk(rng, F * mY * mX, N)
    W = mk(rng, *filt_shape(g), scale=0.04)
    cY, cX = 2 * imgY, 2 * imgX
    acts = np.maximum(mk(rng, C_ * cY * cX, N), 0)  # relu outputs: plenty of exact zeros, so ties inside windows occur
    pooled = acts.reshape(C_, imgY, 2, imgX, 2, N).max(axis=(2, 4)).reshape(C_ * imgY * imgX, N)
    share = mk(rng, C_ * imgY * imgX, N)
    assert capi.lib().raw('mcn_img_acts_max_undo_supported')(N, 2, 0, 2, imgY, imgX, cY, cX, 1) == 1
    for neuron in (1, -1):  # MCN_NEURON_RELU / none
        d_pg = dev(share)
        call('mcn_conv_img_acts' if conv else 'mcn_local_img_acts', dev(hid), dev(W), d_pg, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride,
             groups, sT, 1.0, None)
        d_want = dev_empty(acts.shape, np.nan)
        if neuron >= 0:
            call('mcn_local_max_undo_neuron', dev(acts), d_pg, dev(pooled), d_want, N, C_, cY, cX, 2, 0, 2, imgY, imgX, neuron, 0.0, 0.0, None)
        else:
            call('mcn_local_max_undo', dev(acts), d_pg, dev(pooled), d_want, N, C_, cY, cX, 2, 0, 2, imgY, imgX, 0.0, 1.0, None)
        d_got = dev_empty(acts.shape, np.nan)
        call('mcn_img_acts_max_undo', dev(hid), dev(W), dev(share) if sT else None, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups,
             1 if conv else 0, sT, 1.0, dev(acts), dev(pooled), d_got, neuron, 0.0, 0.0, None)
        got, want = d_got.numpy(), d_want.numpy()
        assert np.all(np.isfinite(got))
        assert np.array_equal(got, want), '%s N=%d sT=%g neuron=%d: %d entries differ' % (name, N, sT, neuron, int((got != want).sum()))
    # the unfused chain against the oracle (once per geometry)
    if sT == 0.0 and N == 96:
        wg = O.img_acts(hid, W, N, conv=conv, **geo_kw(g))
        wu = O.local_max_undo(acts, wg, pooled, N, C_, cY, cX, 2, 0, 2, imgY, imgX)
        wu = wu * (acts > 0)
        d_got = dev_empty(acts.shape, np.nan)
        call('mcn_img_acts_max_undo', dev(hid), dev(W), None, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 1 if conv else 0, 0.0, 1.0,
             dev(acts), dev(pooled), d_got, 1, 0.0, 0.0, None)
        close(d_got.numpy(), wu, what='imgActs + max undo + relu grad')


def test_add_sum_multi_equals_per_matrix_add_sum():
    """mcn_add_sum_multi: the bias-gradient row sums of a whole DeepID2 backward pass (fc6, fc5, local4, local3, conv2, conv1
    shapes: short rows, one-chunk rows, many-chunk rows) in two launches == one mcn_add_sum per matrix, bit for bit."""
    class Item(C.Structure):
        _fields_ = [('tgt', C.c_void_p), ('mat', C.c_void_p), ('rows', C.c_int), ('cols', C.c_int), ('scaleThis', C.c_float),
                    ('scaleSum', C.c_float)]
    rng = np.random.default_rng(46)
    N = 128
    shapes = [(10575, N, 0.0), (160, N, 0.0), (64 * 16, N, 0.5), (48 * 100, N, 0.0), (32, 576 * N, 0.0), (16, 2704 * N, 0.9), (5, 1000, 0.0),
              (3, 8192 * 3 + 4, 0.0)]
    items = (Item * len(shapes))()
    keep, want = [], []
    for i, (rows, cols, sT) in enumerate(shapes):
        mat, t0 = mk(rng, rows, cols), mk(rng, rows)
        d_mat, d_t, d_w = dev(mat), dev(t0), dev(t0)
        nb = ws_bytes('mcn_add_sum_workspace_bytes', rows, cols, 1)
        ws = capi.workspace(nb)
        call('mcn_add_sum', d_w, d_mat, rows, cols, 1, sT, 0.01, ws, nb, None)
        want.append(d_w.numpy().copy())
        keep.append((d_mat, d_t, mat, t0, sT))
        items[i] = Item(d_t.ptr, d_mat.ptr, rows, cols, sT, 0.01)
    L = capi.lib()
    nb = int(L.raw('mcn_add_sum_multi_workspace_bytes')(C.cast(items, C.c_void_p), len(shapes)))
    assert nb == sum(ws_bytes('mcn_add_sum_workspace_bytes', r, c, 1) for r, c, _ in shapes)
    ws = capi.workspace(nb)
    L.call('mcn_add_sum_multi', C.cast(items, C.c_void_p), len(shapes), ws, nb, None)
    for (d_mat, d_t, mat, t0, sT), w, shp in zip(keep, want, shapes):
        got = d_t.numpy()
        assert np.array_equal(got, w), shp
        ref = sT * t0.astype(np.float64) + 0.01 * mat.astype(np.float64).sum(1)
        close(got, ref.astype(np.float32), rtol=2e-5, what='row sums %r' % (shp,))
    with pytest.raises(Exception):
        L.call('mcn_add_sum_multi', C.cast(items, C.c_void_p), 9, ws, nb, None)


@pytest.mark.parametrize('O_,Is,N', [(160, (1024, 1200), 128), (160, (768, 960), 128), (160, (1024,), 96), (512, (512, 2048), 128),
                                      (50, (34, 6, 130), 36)])
def test_fc_bprop_weights_all_inputs_in_one_launch(O_, Is, N):
    """mcn_fc_bprop_weights_multi == one NT product per input (FCLayer::bpropWeights, layer.cu:372-382): momentum-scaled old
    increment plus eps/N times dY X_i^T, each input with its own scales; ragged tiles in every dimension."""
    rng = np.random.default_rng(183)
    dY = mk(rng, O_, N)
    Xs = [mk(rng, I, N) for I in Is]
    t0 = [mk(rng, O_, I, scale=0.01) for I in Is]
    alphas = [5e-4 / N * (k + 1) for k in range(len(Is))]
    betas = [0.9, 0.0, 0.5, 1.0][:len(Is)]
    wants = [O.gemm(dY, X, O_, I, N, tb=True, alpha=a, beta=b, Cmat=t.copy()) for X, I, a, b, t in zip(Xs, Is, alphas, betas, t0)]
    dX, dT = [dev(x) for x in Xs], [dev(t) for t in t0]
    xa, ta_, ka = ptr_array(dX), ptr_array(dT), (C.c_int * len(Is))(*Is)
    d_dy = dev(dY)
    assert capi.lib().raw('mcn_fc_bprop_weights_multi_supported')(d_dy.ptr, xa, ta_, ka, len(Is), O_, N) == 1
    call('mcn_fc_bprop_weights_multi', d_dy, xa, ta_, ka, (C.c_float * len(Is))(*alphas), (C.c_float * len(Is))(*betas), len(Is), O_, N, None)
    for i in range(len(Is)):
        close(dT[i].numpy(), wants[i], what='dW[%d]' % i)


@pytest.mark.parametrize('name', ['conv1_55', 'conv2', 'conv2_plus', 'conv1_39x31', 'cifar_conv1'])
def test_fused_update_finishes_partial_gradients_bitwise(name):
    """mcn_sgd_plan_build_fused: the update launch sums the module-chunk partials of a conv weight gradient
    (mcn_conv_weight_acts_partials) and the stage-1 row sums of its bias gradient (mcn_add_sum_multi_stage1) itself.  Weights,
    increments AND the stored gradients must equal, bit for bit, the chain of separate launches it replaces
    (mcn_conv_weight_acts_reduced -> mcn_add_sum -> mcn_sgd_update_planned): same summation code, same order."""
    class Item(C.Structure):
        _fields_ = [('W', C.c_void_p), ('inc', C.c_void_p), ('grad', C.c_void_p), ('n', C.c_size_t),
                    ('mom', C.c_float), ('wc', C.c_float), ('eps', C.c_float), ('useGrad', C.c_int)]

    class Parts(C.Structure):
        _fields_ = [('data', C.c_void_p), ('count', C.c_int), ('kind', C.c_int), ('scale', C.c_float)]

    class RItem(C.Structure):
        _fields_ = [('tgt', C.c_void_p), ('mat', C.c_void_p), ('rows', C.c_int), ('cols', C.c_int), ('scaleThis', C.c_float),
                    ('scaleSum', C.c_float)]
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 128
    rng = np.random.default_rng(77)
    img, hid = dev(mk(rng, C_ * imgY * imgX, N)), dev(mk(rng, F * mY * mX, N))
    K = (C_ // groups) * fs * fs
    nW = K * F
    W0, inc0 = mk(rng, nW), mk(rng, nW, scale=0.01)
    b0, binc0 = mk(rng, F), mk(rng, F, scale=0.01)
    other0 = mk(rng, 5000)  # an ordinary matrix in the same plan
    L = capi.lib()
    nbytes = ws_bytes('mcn_conv_weight_acts_reduced_workspace_bytes', C_, F, fs, mY, mX, groups)
    rows, cols = F, mY * mX * N
    bias_bytes = ws_bytes('mcn_add_sum_workspace_bytes', rows, cols, 1)
    assert bias_bytes > 0, 'shape must take the two-stage row sum'
    out = []
    for fused in (False, True):
        dW, dInc, dG = dev(W0), dev(inc0), dev_empty(nW, 0)
        dB, dBinc, dBg = dev(b0), dev(binc0), dev_empty(F, 0)
        dO, dOinc, dOg = dev(other0), dev(other0 * 0.01), dev(other0 * 0.001)
        items = (Item * 3)(Item(dW.ptr, dInc.ptr, dG.ptr, nW, 0.9, 5e-4, 5e-4, 1), Item(dB.ptr, dBinc.ptr, dBg.ptr, F, 0.9, 0.0, 1e-3, 1),
                           Item(dO.ptr, dOinc.ptr, dOg.ptr, 5000, 0.9, 5e-4, 5e-4, 1))
        plan = capi.workspace(ws_bytes('mcn_sgd_plan_bytes', 3))
        nb = C.c_int(0)
        wsb = capi.workspace(nbytes)
        bws = capi.workspace(bias_bytes)
        ritems = (RItem * 1)(RItem(dBg.ptr, hid.ptr, rows, cols, 0.0, 1e-3 / N))
        if not fused:
            L.call('mcn_sgd_plan_build', C.cast(items, C.c_void_p), 3, plan, plan.nbytes, C.byref(nb), None)
            for _ in range(2):
                call('mcn_conv_weight_acts_reduced', img, hid, dG, wsb, nbytes, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 0.0, 5e-4 / N, None)
                L.call('mcn_add_sum_multi', C.cast(ritems, C.c_void_p), 1, bws, bias_bytes, None)
                L.call('mcn_sgd_update_planned', plan, 3, nb.value, None)
        else:
            cap = C.c_int(0)
            chunks = L.raw('mcn_conv_weight_acts_partials_count')(N, N, imgY, imgX, C_, F, fs, mY, mX, stride, groups, C.byref(cap))
            assert chunks > 0
            parts = (Parts * 3)(Parts(wsb.ptr, chunks, 1, 1.0), Parts(bws.ptr, bias_bytes // 4 // rows, 2, 1e-3 / N), Parts(None, 0, 0, 1.0))
            L.call('mcn_sgd_plan_build_fused', C.cast(items, C.c_void_p), C.cast(parts, C.c_void_p), 3, plan, plan.nbytes, C.byref(nb), None)
            for _ in range(2):
                got = C.c_int(0)
                call('mcn_conv_weight_acts_partials', img, hid, wsb, nbytes, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 5e-4 / N, C.byref(got),
                     None, 0.0, None, None)
                assert got.value == chunks
                pout = (C.c_void_p * 1)(bws.ptr)
                L.call('mcn_add_sum_multi_stage1', C.cast(ritems, C.c_void_p), 1, pout, None)
                L.call('mcn_sgd_update_planned', plan, 3, nb.value, None)
        out.append([x.numpy().copy() for x in (dW, dInc, dG, dB, dBinc, dBg, dO, dOinc)])
    for a, b, what in zip(out[0], out[1], ('W', 'inc', 'grad', 'bias', 'biasInc', 'biasGrad', 'other', 'otherInc')):
        assert np.array_equal(a, b), '%s: %s differs between the fused and the separate tail' % (name, what)
    assert np.abs(out[0][2]).max() > 0 and np.abs(out[0][5]).max() > 0


@pytest.mark.parametrize('name', ['conv1_55', 'conv1_39x31', 'conv1_plus', 'cifar_conv1', 'conv2'])
def test_conv_weight_acts_partials_leave_the_bias_gradient_in_pieces(name):
    """Where the strip form runs (first-layer shapes) the launch also sums the gradient rows it stages: slabs of numFilters
    floats whose sum is ConvLayer::bpropBiases' row sum (layer.cu:464-474) at the kernel tolerance; elsewhere it says so."""
    g = GEOMS[name]
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, conv = g
    N = 128
    rng = np.random.default_rng(78)
    img, hid = mk(rng, C_ * imgY * imgX, N), mk(rng, F * mY * mX, N)
    L = capi.lib()
    cap = C.c_int(-1)
    chunks = L.raw('mcn_conv_weight_acts_partials_count')(N, N, imgY, imgX, C_, F, fs, mY, mX, stride, groups, C.byref(cap))
    nbytes = ws_bytes('mcn_conv_weight_acts_reduced_workspace_bytes', C_, F, fs, mY, mX, groups)
    wsb = capi.workspace(nbytes)
    bias = dev_empty((chunks * F,), np.nan)
    got, done = C.c_int(0), C.c_int(-1)
    call('mcn_conv_weight_acts_partials', dev(img), dev(hid), wsb, nbytes, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups, 5e-4 / N,
         C.byref(got), bias, 2e-3 / N, C.byref(done), None)
    assert got.value == chunks and done.value == cap.value
    K = (C_ // groups) * fs * fs
    parts = wsb.numpy(shape=(chunks * K * F,), dtype=np.float32).reshape(chunks, K * F)
    want = O.weight_acts(img, hid, N, partial_sum=0, conv=True, sO=5e-4 / N, **geo_kw(g))
    close(parts.astype(np.float64).sum(0).astype(np.float32).reshape(K, F), want, what=name + ' weight gradient from the slabs')
    if name.startswith('conv1') or name == 'cifar_conv1':
        assert done.value == 1, 'first-layer shapes take the strip form'
    if done.value:
        wb = np.zeros(F, np.float32)
        O.add_sum(wb, hid, F, mY * mX * N, 1, 0.0, 2e-3 / N)
        close(bias.numpy().reshape(chunks, F).astype(np.float64).sum(0).astype(np.float32), wb, what=name + ' bias gradient from the slabs')
