"""Random-sparse connectivity through the C ABI against the CPU oracle's restatement of the reference's
*_sparse_rand kernels (filter_acts.cu:401-560, img_acts.cu:556-890, weight_acts.cu:414-560): forward, backward-data and
weight gradient, shared and unshared filters, at the per-kernel tolerance 1e-4*max + 1e-6; the channel passes are exact."""
import ctypes as C

import numpy as np
import pytest

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import capi  # noqa: E402
from kutil import call, close, dev, dev_empty, ws_bytes  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def mk(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def conns(rng, groups, channels, fC):
    """layer.py:830-833: overSample permutations of the channels, one after the other."""
    over = groups * fC // channels
    return np.concatenate([rng.permutation(channels) for _ in range(over)]).astype(np.int32)


# name, img, C, fC, groups, F (total), fs, pad, stride
GEOMS = [
    ('c16_fc8_g4', 12, 16, 8, 4, 64, 3, -1, 1),      # overSample 2
    ('c32_fc8_g4', 9, 32, 8, 4, 64, 5, -2, 2),       # overSample 1: a plain permutation of the channels
    ('c8_fc4_g6', 10, 8, 4, 6, 96, 3, 0, 1),         # overSample 3
    ('c64_fc16_g8_rect', (7, 9), 64, 16, 8, 128, 3, -1, 2),   # overSample 2, rectangular maps
]


def geo(g):
    name, img, C_, fC, groups, F, fs, pad, stride = g
    imgY, imgX = img if isinstance(img, tuple) else (img, img)
    mod = lambda s: 1 + int(np.ceil((2 * -pad + s - fs) / float(stride)))
    return imgY, imgX, C_, fC, groups, F, fs, mod(imgY), mod(imgX), pad, stride


@pytest.mark.parametrize('conv', [True, False], ids=['conv', 'local'])
@pytest.mark.parametrize('N', [128, 36])
@pytest.mark.parametrize('g', GEOMS, ids=[g[0] for g in GEOMS])
def test_sparse_products(g, N, conv):
    rng = np.random.default_rng(70)
    imgY, imgX, C_, fC, groups, F, fs, mY, mX, pad, stride = geo(g)
    M, px = mY * mX, imgY * imgX
    ci = conns(rng, groups, C_, fC)
    d_ci = capi.DeviceBuffer(array=ci)
    images = mk(rng, C_ * px, N)
    filters = mk(rng, (1 if conv else M) * fC * fs * fs, F, scale=0.1)
    hid = mk(rng, F * M, N)
    gb = ws_bytes('mcn_sparse_gathered_bytes', N, imgY, imgX, fC, groups)
    ib = ws_bytes('mcn_sparse_img_acts_workspace_bytes', N, imgY, imgX, fC, groups)
    assert gb == 4 * groups * fC * px * N and ib > gb
    ws = capi.workspace(ib)
    kind = 'conv' if conv else 'local'

    # forward, plain and accumulating
    for sT in (0.0, 1.0):
        t0 = mk(rng, F * M, N)
        want = O.filter_acts_sparse(images, filters, ci, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, targets=t0.copy(),
                                    sT=sT, sO=0.5, conv=conv)
        d_t = dev(t0)
        call('mcn_%s_filter_acts_sparse' % kind, dev(images), dev(filters), d_t, d_ci, ws, ib, N, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad,
             stride, groups, sT, 0.5, None)
        close(d_t.numpy(), want, what='%s filterActsSparse sT=%g' % (kind, sT))

    # backward-data: the sum over the overSample copies the reference's kernel writes
    for sT in (0.0, 1.0):
        t0 = mk(rng, C_ * px, N)
        want = O.img_acts_sparse_summed(hid, filters, ci, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, targets=t0.copy(),
                                        sT=sT, sO=1.0, conv=conv)
        d_t = dev(t0)
        call('mcn_%s_img_acts_sparse' % kind, dev(hid), dev(filters), d_t, d_ci, ws, ib, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride,
             groups, sT, 1.0, None)
        close(d_t.numpy(), want, what='%s imgActsSparse sT=%g' % (kind, sT))

    # weight gradient
    if conv:
        for ps in (0, 1):
            want = O.weight_acts_sparse(images, hid, ci, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, partial_sum=ps,
                                        sO=1.0 / N, conv=True)
            d_t = dev_empty(want.shape, np.nan)
            call('mcn_conv_weight_acts_sparse', dev(images), dev(hid), d_t, d_ci, ws, ib, N, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride,
                 groups, ps, 0.0, 1.0 / N, None)
            close(d_t.numpy(), want, what='conv weightActsSparse partialSum=%d' % ps)
    else:
        t0 = mk(rng, M * fC * fs * fs, F, scale=0.01)
        want = O.weight_acts_sparse(images, hid, ci, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, targets=t0.copy(), sT=0.9,
                                    sO=1.0 / N, conv=False)
        d_t = dev(t0)
        call('mcn_local_weight_acts_sparse', dev(images), dev(hid), d_t, d_ci, ws, ib, N, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride,
             groups, 0.9, 1.0 / N, None)
        close(d_t.numpy(), want, what='local weightActsSparse with momentum')

    # <filterActs(x), h> == <x, imgActs(h)>: the two sparse products are transposes of each other
    fwd = O.filter_acts_sparse(images, filters, ci, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, conv=conv)
    d_t = dev_empty((C_ * px, N), np.nan)
    call('mcn_%s_img_acts_sparse' % kind, dev(hid), dev(filters), d_t, d_ci, ws, ib, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride,
         groups, 0.0, 1.0, None)
    lhs = float((fwd.astype(np.float64) * hid).sum())
    rhs = float((images.astype(np.float64) * d_t.numpy()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0)


def test_channel_passes_are_exact():
    rng = np.random.default_rng(71)
    N, C_, px, over = 20, 12, 35, 3
    ci = np.concatenate([rng.permutation(C_) for _ in range(over)]).astype(np.int32)
    d_ci = capi.DeviceBuffer(array=ci)
    images = mk(rng, C_ * px, N + 4)[:, :N]   # a column slice: imgStride = N + 4
    wide = mk(rng, C_ * px, N + 4)
    wide[:, :N] = images
    d_g = dev_empty((over * C_ * px, N), np.nan)
    call('mcn_gather_channels', dev(wide), N + 4, d_ci, d_g, N, C_, over * C_, px, None)
    got = d_g.numpy().reshape(over * C_, px, N)
    assert np.array_equal(got, images.reshape(C_, px, N)[ci])
    d_inv = capi.DeviceBuffer(nbytes=4 * over * C_)
    call('mcn_invert_channel_indices', d_ci, C.cast(d_inv.ptr, C.POINTER(C.c_int)), C_, over, None)
    inv = d_inv.numpy((over * C_,), np.int32)
    for s in range(over):
        assert np.array_equal(ci[inv[s * C_:(s + 1) * C_]], np.arange(C_))
        assert inv[s * C_:(s + 1) * C_].min() >= s * C_ and inv[s * C_:(s + 1) * C_].max() < (s + 1) * C_
    t0 = mk(rng, C_ * px, N)
    d_t = dev(t0)
    call('mcn_scatter_channels_sum', d_g, C.cast(d_inv.ptr, C.POINTER(C.c_int)), d_t, N, C_, over, px, 1.0, 1.0, None)
    # every permutation sends each channel back to itself: over copies of the image on top of the target
    want = t0.astype(np.float64) + over * np.ascontiguousarray(images).astype(np.float64)
    close(d_t.numpy(), want.astype(np.float32), rtol=1e-6, what='scatter of a gather')


def test_sparse_arguments_are_checked():
    L = capi.lib()
    a = capi.DeviceBuffer(nbytes=1 << 16)
    args = lambda fC, groups, wsb: (a.ptr, a.ptr, a.ptr, a.ptr, a.ptr, wsb, 4, 4, 4, 4, 8, fC, 16 * groups, 3, 2, 2, 0, 1, groups,
                                     C.c_float(0), C.c_float(1), None)
    with pytest.raises(capi.McnError, match='numGroups > 1'):
        L.call('mcn_conv_filter_acts_sparse', *args(4, 1, 1 << 16))
    with pytest.raises(capi.McnError, match='divisible by 4'):
        L.call('mcn_conv_filter_acts_sparse', *args(2, 4, 1 << 16))
    with pytest.raises(capi.McnError, match='workspace too small'):
        L.call('mcn_conv_filter_acts_sparse', *args(4, 2, 16))
