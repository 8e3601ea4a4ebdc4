"""An independent second formulation of the oracle's core products (CPU only, float64): the oracle and the HIP kernels were
written from the same reading of the reference's layouts (filter_acts.cu:39-43, layer.py:943, :958), so a shared mis-reading
of an index would pass every oracle-vs-kernel test.  Here the oracle is compared with formulations that share nothing with
it but the layout statement itself:

  * shared-weight conv forward / backward-data / backward-weights  <->  torch.nn.functional.conv2d and its autograd
    (groups 1 and 2, zero padding, stride 2, rectangular images, modules hanging over the right / bottom edge);
  * unshared local layers  <->  unfold + einsum with one filter bank per module, and autograd of that;
  * max / avg pooling with the reference's ceil output rule (layer.py:1018) and clipped average (conv_util.cuh:166,187)
    <->  max_pool2d / avg_pool2d(ceil_mode=True); max-undo / avg-undo <-> their autograd (no ties in random data);
  * cross-map response normalisation (conv_util.cu:1038-1089)  <->  torch local_response_norm;
  * softmax, logreg cost, FC products  <->  float64 numpy.

This does not pin the oracle to the reference (nothing can here: SURVEY 8c) -- it removes the shared-misreading risk.
torch is imported in this CPU test only; no GPU process of the test suite imports it.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F_  # noqa: E402

from oracle import oracle as K  # noqa: E402

torch.set_num_threads(4)
RNG = np.random.default_rng(11)


def to_nchw(a, C, H, W):
    """(C*H*W, N) images-fastest  ->  (N, C, H, W) float64"""
    return torch.from_numpy(np.ascontiguousarray(a.reshape(C, H, W, -1).transpose(3, 0, 1, 2)).astype(np.float64))


def from_nchw(t):
    n, c, h, w = t.shape
    return t.detach().numpy().transpose(1, 2, 3, 0).reshape(c * h * w, n)


def conv_weight(w, fC, fs, F):
    """(fC*fs*fs, F)  ->  torch (F, fC, fs, fs)"""
    return torch.from_numpy(np.ascontiguousarray(w.reshape(fC, fs, fs, F).transpose(3, 0, 1, 2)).astype(np.float64))


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err, ref = np.max(np.abs(a - b)), np.max(np.abs(b))
    assert err <= tol * ref + 1e-7, 'max abs err %.3e vs max %.3e' % (err, ref)


def modules(img, fs, pad, stride):
    return 1 + int(np.ceil((2 * pad + img - fs) / float(stride)))  # layer.py:885


CONV_CASES = [
    # C, H, W, F, fs, pad, stride, groups, N
    (3, 11, 9, 16, 4, 0, 1, 1, 8),     # DeepID conv1 in small, rectangular
    (16, 9, 9, 32, 3, 0, 1, 1, 4),     # conv2
    (3, 12, 12, 16, 5, 2, 1, 1, 4),    # CIFAR 5x5 pad 2
    (4, 10, 13, 8, 3, 1, 2, 1, 4),     # stride 2: the last module hangs over the edge
    (8, 9, 9, 16, 3, 1, 1, 2, 4),      # two groups
    (8, 11, 8, 32, 4, 2, 2, 2, 3),     # groups + padding + stride 2, rectangular
]


def torch_conv(x, w, C, H, W, Fn, fs, pad, stride, groups, mY, mX):
    # extra zeros on the right / bottom so that torch also evaluates the modules that reach past the padded image
    needY = (mY - 1) * stride + fs - (H + 2 * pad)
    needX = (mX - 1) * stride + fs - (W + 2 * pad)
    xp = F_.pad(x, (pad, pad + max(needX, 0), pad, pad + max(needY, 0)))
    y = F_.conv2d(xp, w, stride=stride, groups=groups)
    assert y.shape[2] >= mY and y.shape[3] >= mX
    return y[:, :, :mY, :mX]


@pytest.mark.parametrize('C,H,W,Fn,fs,pad,stride,groups,N', CONV_CASES)
def test_conv_products_against_torch_conv2d(C, H, W, Fn, fs, pad, stride, groups, N):
    mY, mX = modules(H, fs, pad, stride), modules(W, fs, pad, stride)
    fC = C // groups
    img = RNG.standard_normal((C * H * W, N)).astype(np.float32)
    w = RNG.standard_normal((fC * fs * fs, Fn)).astype(np.float32)
    geo = dict(imgY=H, imgX=W, C_=C, F=Fn, fs=fs, mY=mY, mX=mX, pad=-pad, stride=stride, groups=groups)
    x = to_nchw(img, C, H, W).requires_grad_(True)
    wt = conv_weight(w, fC, fs, Fn).requires_grad_(True)
    y = torch_conv(x, wt, C, H, W, Fn, fs, pad, stride, groups, mY, mX)
    got = K.filter_acts(img, w, N, conv=True, **geo)
    close(got, from_nchw(y))
    # backward: a random output gradient through autograd
    g = RNG.standard_normal((Fn * mY * mX, N)).astype(np.float32)
    y.backward(to_nchw(g, Fn, mY, mX))
    close(K.img_acts(g, w, N, conv=True, **geo), from_nchw(x.grad))
    dw = K.weight_acts(img, g, N, partial_sum=0, conv=True, **geo)
    want_dw = wt.grad.numpy().transpose(1, 2, 3, 0).reshape(fC * fs * fs, Fn)
    close(dw, want_dw, tol=5e-5)
    # scaleTargets / scaleOutput (cudaconv2.cuh:36-78): tgt = sT * tgt + sO * product
    t0 = RNG.standard_normal(got.shape).astype(np.float32)
    close(K.filter_acts(img, w, N, conv=True, targets=t0.copy(), sT=0.5, sO=-2.0, **geo), 0.5 * t0 - 2.0 * from_nchw(y))


LOCAL_CASES = [
    (32, 6, 6, 48, 3, 0, 1, 4),    # local3 in small
    (48, 5, 5, 64, 2, 0, 1, 4),    # local4
    (4, 7, 5, 8, 3, 1, 1, 3),      # padded, rectangular
    (4, 8, 8, 8, 3, 0, 2, 3),      # stride 2
]


@pytest.mark.parametrize('C,H,W,Fn,fs,pad,stride,N', LOCAL_CASES)
def test_local_products_against_unfold_einsum(C, H, W, Fn, fs, pad, stride, N):
    mY, mX = modules(H, fs, pad, stride), modules(W, fs, pad, stride)
    M = mY * mX
    img = RNG.standard_normal((C * H * W, N)).astype(np.float32)
    w = RNG.standard_normal((M * C * fs * fs, Fn)).astype(np.float32)  # one (C*fs*fs, F) bank per module (layer.py:958)
    geo = dict(imgY=H, imgX=W, C_=C, F=Fn, fs=fs, mY=mY, mX=mX, pad=-pad, stride=stride, groups=1)
    x = to_nchw(img, C, H, W).requires_grad_(True)
    needY = (mY - 1) * stride + fs - (H + 2 * pad)
    needX = (mX - 1) * stride + fs - (W + 2 * pad)
    xp = F_.pad(x, (pad, pad + max(needX, 0), pad, pad + max(needY, 0)))
    patches = F_.unfold(xp, fs, stride=stride)                        # (N, C*fs*fs, L), rows ordered (c, fy, fx)
    L_x = (xp.shape[3] - fs) // stride + 1
    keep = [my * L_x + mx for my in range(mY) for mx in range(mX)]
    patches = patches[:, :, keep]                                     # (N, K, M)
    wt = torch.from_numpy(w.reshape(M, C * fs * fs, Fn).astype(np.float64)).requires_grad_(True)
    y = torch.einsum('nkm,mkf->nfm', patches, wt)                     # (N, F, M)
    got = K.filter_acts(img, w, N, conv=False, **geo)
    close(got, y.detach().numpy().transpose(1, 2, 0).reshape(Fn * M, N))
    g = RNG.standard_normal((Fn * M, N)).astype(np.float32)
    y.backward(torch.from_numpy(g.reshape(Fn, M, N).transpose(2, 0, 1).astype(np.float64)))
    close(K.img_acts(g, w, N, conv=False, **geo), from_nchw(x.grad))
    close(K.weight_acts(img, g, N, conv=False, **geo), wt.grad.numpy().reshape(M * C * fs * fs, Fn), tol=5e-5)


POOL_CASES = [
    # channels, H, W, size, stride, N
    (4, 8, 8, 2, 2, 4),     # DeepID pools: windows tile the map
    (4, 9, 7, 2, 2, 4),     # odd sizes: the last window is clipped
    (3, 10, 10, 3, 2, 4),   # CIFAR 3/2: overlapping windows
    (2, 11, 13, 3, 2, 3),
]


@pytest.mark.parametrize('ch,H,W,size,stride,N', POOL_CASES)
@pytest.mark.parametrize('mode', ['max', 'avg'])
def test_pooling_against_torch_ceil_mode(ch, H, W, size, stride, N, mode):
    outY = int(np.ceil((H - size) / float(stride))) + 1  # layer.py:1018 with start = 0
    outX = int(np.ceil((W - size) / float(stride))) + 1
    img = RNG.standard_normal((ch * H * W, N)).astype(np.float32)
    x = to_nchw(img, ch, H, W).requires_grad_(True)
    if mode == 'max':
        y = F_.max_pool2d(x, size, stride, ceil_mode=True)
    else:
        y = F_.avg_pool2d(x, size, stride, ceil_mode=True, count_include_pad=False)
    assert tuple(y.shape[2:]) == (outY, outX)
    got = K.local_pool(img, N, ch, H, W, size, 0, stride, outY, outX, mode)
    close(got, from_nchw(y), tol=1e-6)
    g = RNG.standard_normal((ch * outY * outX, N)).astype(np.float32)
    y.backward(to_nchw(g, ch, outY, outX))
    if mode == 'max':
        back = K.local_max_undo(img, g, got, N, ch, H, W, size, 0, stride, outY, outX)
    else:
        back = K.local_avg_undo(g, N, ch, H, W, size, 0, stride, outY, outX)
    close(back, from_nchw(x.grad), tol=1e-5)


@pytest.mark.parametrize('ch,size', [(16, 5), (32, 9), (16, 4)])
def test_cross_map_response_norm_against_torch_lrn(ch, size):
    px, N = 6, 4
    add_scale, pw = 0.001 / size, 0.75  # (layer.py:1033-1037: the parser divides `scale` by the window size)
    img = (RNG.standard_normal((ch * px, N)) * 20).astype(np.float32)
    acts, denoms = K.response_norm_cross_map(img, N, ch, px, size, add_scale, pw)
    x = torch.from_numpy(img.reshape(ch, px, N).transpose(2, 0, 1).astype(np.float64)).requires_grad_(True)  # (N, C, px)
    if size % 2 == 1:
        y = F_.local_response_norm(x, size, alpha=add_scale * size, beta=pw, k=1.0)
    else:
        # even windows: the reference's window is [f - size/2, f - size/2 + size) (conv_util.cu:1068-1070)
        sq = F_.pad(x * x, (0, 0, size // 2, size - size // 2 - 1))
        den = 1.0 + add_scale * sum(sq[:, k:k + ch] for k in range(size))
        y = x * den.pow(-pw)
    want = y.detach().numpy().transpose(1, 2, 0).reshape(ch * px, N)
    close(acts, want, tol=1e-5)
    g = RNG.standard_normal((ch * px, N)).astype(np.float32)
    y.backward(torch.from_numpy(g.reshape(ch, px, N).transpose(2, 0, 1).astype(np.float64)))
    back = K.response_norm_cross_map_undo(g, denoms, img, acts.copy(), N, ch, px, size, add_scale, pw)
    close(back, x.grad.numpy().transpose(1, 2, 0).reshape(ch * px, N), tol=2e-5)


def test_softmax_logreg_and_fc_against_float64():
    Kc, N, I = 37, 12, 20
    x = (RNG.standard_normal((Kc, N)) * 3).astype(np.float32)
    p = K.softmax(x, Kc, N)
    e = np.exp(x.astype(np.float64) - x.max(0, keepdims=True))
    want = e / e.sum(0, keepdims=True)
    close(p, want, tol=1e-5)
    labels = RNG.integers(0, Kc, size=(1, N)).astype(np.float32)
    logp, correct = K.logreg_cost(labels, p, Kc, N)
    idx = labels[0].astype(int)
    close(np.asarray(logp).reshape(-1), np.log(want[idx, np.arange(N)]), tol=1e-5)
    assert np.array_equal(np.asarray(correct).reshape(-1) > 0, want.argmax(0) == idx)
    g = K.logreg_softmax_grad(labels, p, Kc, N, 1.0)
    onehot = np.zeros((Kc, N))
    onehot[idx, np.arange(N)] = 1
    close(g, onehot - want, tol=1e-5)  # ascent direction: d(log p[label]) / dx (layer_kernels.cu:154-170)
    # FC products in the device orientation: Y[O x N] = W^T[O x I] X[I x N]
    W = RNG.standard_normal((Kc, I)).astype(np.float32)  # memory image [outputs][inputs]
    X = RNG.standard_normal((I, N)).astype(np.float32)
    close(K.gemm(W, X, Kc, N, I), W.astype(np.float64) @ X, tol=1e-5)
    dY = RNG.standard_normal((Kc, N)).astype(np.float32)
    close(K.gemm(W, dY, I, N, Kc, ta=True), W.astype(np.float64).T @ dY, tol=1e-5)
    close(K.gemm(dY, X, Kc, I, N, tb=True), dY.astype(np.float64) @ X.T.astype(np.float64), tol=1e-5)
