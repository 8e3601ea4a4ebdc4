"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (the CPU
restatement of the cuda-convnet compute core, see convnet_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  PARITY UNPINNED (no golden vectors exist in the reference); soft
pins are in tests/test_oracle_pins.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NEURON_TYPES = {'ident': 0, 'relu': 1, 'logistic': 2, 'brelu': 3, 'abs': 4, 'tanh': 5,
                'softrelu': 6, 'square': 7, 'sqrt': 8, 'linear': 9}


def build():
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_sum_all.restype = C.c_double
        _LIB.orc_l2sn_cost.restype = C.c_float
        _LIB.orc_l2sn_grad.restype = C.c_float
        _LIB.orc_l2sn_optimal_thr.restype = C.c_float
    return _LIB


def _p(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float32, (type(a), getattr(a, 'dtype', None))
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f(v):
    return C.c_float(v)


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def filter_acts(images, filters, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups=1,
                targets=None, sT=0.0, sO=1.0, conv=True, img_stride=None):
    M = mY * mX
    if targets is None:
        targets = np.zeros((F * M, N), np.float32)
    lib().orc_filter_acts(_p(images), _p(filters), _p(targets), N, img_stride or N, imgY, imgX, C_, F, fs,
                          mY, mX, pad, stride, groups, _f(sT), _f(sO), int(conv))
    return targets


def img_acts(hid, filters, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups=1,
             targets=None, sT=0.0, sO=1.0, conv=True):
    if targets is None:
        targets = np.zeros((C_ * imgY * imgX, N), np.float32)
    lib().orc_img_acts(_p(hid), _p(filters), _p(targets), N, imgY, imgX, C_, F, fs, mY, mX, pad, stride,
                       groups, _f(sT), _f(sO), int(conv))
    return targets


def weight_acts(images, hid, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups=1, partial_sum=0,
                targets=None, sT=0.0, sO=1.0, conv=True, img_stride=None):
    M = mY * mX
    fc = C_ // groups
    if not conv:
        partial_sum = 1
    ps = partial_sum if partial_sum > 0 else M
    if targets is None:
        targets = np.zeros(((M // ps) * fc * fs * fs, F), np.float32)
    lib().orc_weight_acts(_p(images), _p(hid), _p(targets), N, img_stride or N, imgY, imgX, C_, F, fs, mY, mX,
                          pad, stride, groups, ps, _f(sT), _f(sO))
    return targets


def _pi(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int))


def filter_acts_sparse(images, filters, conns, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups,
                       targets=None, sT=0.0, sO=1.0, conv=True, img_stride=None):
    """conv/localFilterActsSparse: conns (groups*fC,) int32 = colorIndices."""
    if targets is None:
        targets = np.zeros((F * mY * mX, N), np.float32)
    lib().orc_filter_acts_sparse(_p(images), _p(filters), _p(targets), _pi(conns), N, img_stride or N, imgY, imgX, C_, fC, F, fs,
                                 mY, mX, pad, stride, groups, _f(sT), _f(sO), int(conv))
    return targets


def img_acts_sparse(hid, filters, conns, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, conv=True):
    """conv/localImgActsSparse: the (overSample, C, pixels, N) copies the reference's kernel writes."""
    over = groups * fC // C_
    targets = np.zeros((over * C_ * imgY * imgX, N), np.float32)
    lib().orc_img_acts_sparse(_p(hid), _p(filters), _p(targets), _pi(conns), N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride,
                              groups, _f(0.0), _f(1.0), int(conv))
    return targets


def img_acts_sparse_summed(hid, filters, conns, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups,
                           targets=None, sT=0.0, sO=1.0, conv=True):
    """... followed by the sum over the copies that ConvLayer::bpropActs takes (layer.cu:500-508), with the accumulate
    rule every other backward product has (the reference's sum overwrites the target even when scaleTargets == 1)."""
    over = groups * fC // C_
    copies = img_acts_sparse(hid, filters, conns, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, conv)
    acc = np.zeros((C_ * imgY * imgX, N), np.float32)
    for s in range(over):
        acc += copies.reshape(over, C_ * imgY * imgX, N)[s]
    if targets is None or sT == 0.0:
        out = np.float32(sO) * acc
        if targets is not None:
            targets[...] = out
            return targets
        return out
    targets[...] = np.float32(sT) * targets + np.float32(sO) * acc
    return targets


def weight_acts_sparse(images, hid, conns, N, imgY, imgX, C_, fC, F, fs, mY, mX, pad, stride, groups, partial_sum=0,
                       targets=None, sT=0.0, sO=1.0, conv=True, img_stride=None):
    M = mY * mX
    if not conv:
        partial_sum = 1
    ps = partial_sum if partial_sum > 0 else M
    if targets is None:
        targets = np.zeros(((M // ps) * fC * fs * fs, F), np.float32)
    lib().orc_weight_acts_sparse(_p(images), _p(hid), _p(targets), _pi(conns), N, img_stride or N, imgY, imgX, C_, fC, F, fs, mY, mX,
                                 pad, stride, groups, ps, _f(sT), _f(sO))
    return targets


def local_pool(imgs, N, F, imgY, imgX, subsX, start, stride, outY, outX, mode):
    target = np.empty((F * outY * outX, N), np.float32)
    lib().orc_local_pool(_p(imgs), _p(target), N, F, imgY, imgX, subsX, start, stride, outY, outX,
                         0 if mode == 'max' else 1)
    return target


def local_max_undo(imgs, max_grads, max_acts, N, F, imgY, imgX, subsX, start, stride, outY, outX,
                   target=None, sT=0.0, sO=1.0):
    if target is None:
        target = np.zeros((F * imgY * imgX, N), np.float32)
    lib().orc_local_max_undo(_p(imgs), _p(max_grads), _p(max_acts), _p(target), N, F, imgY, imgX, subsX,
                             start, stride, outY, outX, _f(sT), _f(sO))
    return target


def local_avg_undo(avg_grads, N, F, imgY, imgX, subsX, start, stride, outY, outX, target=None, sT=0.0, sO=1.0):
    if target is None:
        target = np.zeros((F * imgY * imgX, N), np.float32)
    lib().orc_local_avg_undo(_p(avg_grads), _p(target), N, F, imgY, imgX, subsX, start, stride, outY, outX,
                             _f(sT), _f(sO))
    return target


def response_norm(images, N, F, imgY, imgX, size, add_scale, pow_scale):
    denoms = np.empty_like(images)
    target = np.empty_like(images)
    lib().orc_response_norm(_p(images), _p(denoms), _p(target), N, F, imgY, imgX, size, _f(add_scale), _f(pow_scale))
    return target, denoms


def contrast_norm(images, mean_diffs, N, F, imgY, imgX, size, add_scale, pow_scale):
    """convContrastNorm (conv_util.cu:2074-2160) -> (target, denoms)."""
    denoms = np.empty_like(images)
    target = np.empty_like(images)
    lib().orc_contrast_norm(_p(images), _p(mean_diffs), _p(denoms), _p(target), N, F, imgY, imgX, size, _f(add_scale), _f(pow_scale))
    return target, denoms


def response_norm_undo(out_grads, denoms, inputs, acts, N, F, imgY, imgX, size, add_scale, pow_scale,
                       target=None, sT=0.0, sO=1.0):
    """NOTE: overwrites `acts` like the reference (conv_util.cu:2226)."""
    if target is None:
        target = np.zeros_like(out_grads)
    lib().orc_response_norm_undo(_p(out_grads), _p(denoms), _p(inputs), _p(acts), _p(target), N, F, imgY, imgX,
                                 size, _f(add_scale), _f(pow_scale), _f(sT), _f(sO))
    return target


def response_norm_cross_map(images, N, F, img_pixels, size, add_scale, pow_scale, blocked=False):
    denoms = np.empty_like(images)
    target = np.empty_like(images)
    lib().orc_response_norm_cross_map(_p(images), _p(denoms), _p(target), N, F, img_pixels, size, _f(add_scale),
                                      _f(pow_scale), int(blocked))
    return target, denoms


def response_norm_cross_map_undo(out_grads, denoms, inputs, acts, N, F, img_pixels, size, add_scale, pow_scale,
                                 blocked=False, target=None, sT=0.0, sO=1.0):
    if target is None:
        target = np.zeros_like(out_grads)
    lib().orc_response_norm_cross_map_undo(_p(out_grads), _p(denoms), _p(inputs), _p(acts), _p(target), N, F,
                                           img_pixels, size, _f(add_scale), _f(pow_scale), int(blocked),
                                           _f(sT), _f(sO))
    return target


def neuron_fwd(x, ntype, a=0.0, b=0.0, out=None):
    if out is None:
        out = np.empty_like(x)
    lib().orc_neuron_fwd(_p(x), _p(out), C.c_size_t(x.size), NEURON_TYPES[ntype], _f(a), _f(b))
    return out


def neuron_bwd(acts_grad, acts, inputs, ntype, a=0.0, b=0.0, target=None, add=False):
    if target is None:
        target = np.zeros_like(acts_grad)
    lib().orc_neuron_bwd(_p(acts_grad), _p(acts), _p(inputs), _p(target), C.c_size_t(acts_grad.size),
                         NEURON_TYPES[ntype], _f(a), _f(b), int(add))
    return target


def gemm(A, B, M, N, K, ta=False, tb=False, alpha=1.0, beta=0.0, Cmat=None):
    if Cmat is None:
        Cmat = np.zeros((M, N), np.float32)
    lib().orc_gemm(_p(A), _p(B), _p(Cmat), M, N, K, int(ta), int(tb), _f(alpha), _f(beta))
    return Cmat


def add_vector(mat, vec, rows, cols, per_row, scale=1.0):
    lib().orc_add_vector(_p(mat), _p(vec), rows, cols, int(per_row), _f(scale))
    return mat


def add_sum(tgt, mat, rows, cols, axis, scale_this, scale_sum):
    lib().orc_add_sum(_p(tgt), _p(mat), rows, cols, axis, _f(scale_this), _f(scale_sum))
    return tgt


def axpby(a, b, sa, sb):
    lib().orc_axpby(_p(a), _p(b), C.c_size_t(a.size), _f(sa), _f(sb))
    return a


def sum_all(a):
    return float(lib().orc_sum_all(_p(a), C.c_size_t(a.size)))


def softmax(x, K, N):
    p = np.empty((K, N), np.float32)
    lib().orc_softmax(_p(x), _p(p), K, N)
    return p


def logreg_cost(labels, probs, K, N):
    lp = np.empty((1, N), np.float32)
    cp = np.empty((1, N), np.float32)
    lib().orc_logreg_cost(_p(labels), _p(probs), _p(lp), _p(cp), K, N)
    return lp, cp


def logreg_softmax_grad(labels, probs, K, N, coeff, target=None, add=False):
    if target is None:
        target = np.zeros((K, N), np.float32)
    lib().orc_logreg_softmax_grad(_p(labels), _p(probs), _p(target), K, N, int(add), _f(coeff))
    return target


def logreg_grad(labels, probs, K, N, coeff, target=None, add=False):
    if target is None:
        target = np.zeros((K, N), np.float32)
    lib().orc_logreg_grad(_p(labels), _p(probs), _p(target), K, N, int(add), _f(coeff))
    return target


def softmax_grad(acts, acts_grad, K, N, target=None, add=False):
    if target is None:
        target = np.zeros((K, N), np.float32)
    lib().orc_softmax_grad(_p(acts), _p(acts_grad), _p(target), K, N, int(add))
    return target


def l2sn_cost(idens, data, D, N, m=-1.0):
    """Returns (output[N/2], m, m_idx, dists[N/2], labels[N/2])."""
    out = np.empty((1, N // 2), np.float32)
    dists = np.empty(N // 2, np.float32)
    labels = np.empty(N // 2, np.float32)
    idx = C.c_int(-1)
    m_used = lib().orc_l2sn_cost(_p(idens), _p(data), _p(out), D, N, _f(m), C.byref(idx), _p(dists), _p(labels))
    return out, float(m_used), idx.value, dists, labels


def l2sn_optimal_thr(dists, labels):
    """getOptimalThr (layer_kernels.cu:482-498) over the given candidates -> (m, index)."""
    dists, labels = _c(dists).reshape(-1), _c(labels).reshape(-1)
    idx = C.c_int(-1)
    m = lib().orc_l2sn_optimal_thr(_p(dists), _p(labels), dists.size, C.byref(idx))
    return float(m), idx.value


def l2sn_grad(idens, data, D, N, coeff, m=-1.0, target=None, add=False):
    if target is None:
        target = np.zeros((D, N), np.float32)
    lib().orc_l2sn_grad(_p(idens), _p(data), _p(target), D, N, _f(m), int(add), _f(coeff))
    return target


def l2reg_cost(ground, data, D, N):
    out = np.empty((1, N), np.float32)
    lib().orc_l2reg_cost(_p(ground), _p(data), _p(out), D, N)
    return out


def l2reg_grad(ground, data, D, N, coeff, target=None, add=False):
    if target is None:
        target = np.zeros((D, N), np.float32)
    lib().orc_l2reg_grad(_p(ground), _p(data), _p(target), D, N, int(add), _f(coeff))
    return target


def sgd_update(W, inc, grad, mom, wc, eps, use_grad):
    lib().orc_sgd_update(_p(W), _p(inc), _p(grad) if grad is not None else None, C.c_size_t(W.size),
                         _f(mom), _f(wc), _f(eps), int(use_grad))


def fisher2_cost(labels, data, D, N, alpha):
    """computeFisher2Cost (layer_kernels.cu:310-369) -> the scalar the layer multiplies by numCases (layer.cu:1031-1034)."""
    lib().orc_fisher2_cost.restype = C.c_float
    return float(lib().orc_fisher2_cost(_p(_c(labels)), _p(_c(data)), D, N, _f(alpha)))


def fisher2_grad(labels, data, D, N, alpha, coeff, target=None, add=False):
    """computeFisher2Grad (layer_kernels.cu:371-479)."""
    if target is None:
        target = np.zeros((D, N), np.float32)
    lib().orc_fisher2_grad(_p(_c(labels)), _p(_c(data)), _p(target), D, N, _f(alpha), int(add), _f(coeff))
    return target


# ---- image-op layers outside the DeepID stacks
def bed_of_nails(imgs, N, C_, imgY, imgX, start, stride, outY, outX, target=None, sT=0.0, sO=1.0):
    if target is None:
        target = np.zeros((C_ * outY * outX, N), np.float32)
    lib().orc_bed_of_nails(_p(imgs), _p(target), N, C_, imgY, imgX, start, stride, outY, outX, _f(sT), _f(sO))
    return target


def bed_of_nails_undo(grads, N, C_, imgY, imgX, start, stride, outY, outX, target=None, sT=0.0, sO=1.0):
    if target is None:
        target = np.zeros((C_ * imgY * imgX, N), np.float32)
    lib().orc_bed_of_nails_undo(_p(grads), _p(target), N, C_, imgY, imgX, start, stride, outY, outX, _f(sT), _f(sO))
    return target


def gaussian_blur(imgs, filt, N, C_, imgY, imgX, horiz, target=None, sT=0.0, sO=1.0):
    if target is None:
        target = np.zeros((C_ * imgY * imgX, N), np.float32)
    filt = np.ascontiguousarray(filt, np.float32).ravel()
    lib().orc_gaussian_blur(_p(imgs), _p(filt), _p(target), N, C_, imgY, imgX, filt.size, int(horiz), _f(sT), _f(sO))
    return target


def resize_bilinear(imgs, N, C_, img_size, tgt_size, scale):
    target = np.empty((C_ * tgt_size * tgt_size, N), np.float32)
    lib().orc_resize_bilinear(_p(imgs), _p(target), N, C_, img_size, tgt_size, _f(scale))
    return target


def rgb_to_yuv(imgs, N, pixels):
    target = np.empty((3 * pixels, N), np.float32)
    lib().orc_rgb_to_yuv(_p(imgs), _p(target), N, pixels)
    return target


def rgb_to_lab(imgs, N, pixels, center):
    target = np.empty((3 * pixels, N), np.float32)
    lib().orc_rgb_to_lab(_p(imgs), _p(target), N, pixels, int(center))
    return target


def shift(imgs, filt, N, C_, img_pixels):
    filt = np.ascontiguousarray(filt, np.float32)
    tgt_pixels, nf = filt.shape
    target = np.empty((nf * C_ * tgt_pixels, N), np.float32)
    lib().orc_shift(_p(imgs), _p(filt), _p(target), N, C_, img_pixels, tgt_pixels, nf)
    return target
