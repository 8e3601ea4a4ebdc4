"""Helpers shared by the GPU parity tests: device upload/download through the C ABI
and the stated fp32 tolerance."""
import ctypes as C

import numpy as np

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import capi  # noqa: E402

RTOL, ATOL = 1e-4, 1e-6


def close(got, want, rtol=RTOL, atol=ATOL, what=''):
    """max|got-want| <= rtol*max|want| + atol  (per tensor)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.all(np.isfinite(got)), '%s: non-finite values in the HIP result' % what
    err = np.max(np.abs(got - want)) if got.size else 0.0
    bound = rtol * (np.max(np.abs(want)) if want.size else 0.0) + atol
    assert err <= bound, '%s: max abs err %.3e > %.3e' % (what, err, bound)


def close_but_for_decisions(got, want, tol, what='', atol=1e-9, outlier_frac=1e-3, outlier_factor=50):
    """Gradients of a whole training step from two differently tiled forward passes (GPU vs oracle, 32 images per rank vs 128 on one
    GPU).  Rounding differences stay within tol x max|want|; but a relu input or two max-pool candidates that lie within rounding
    distance of each other get DECIDED differently now and then (tools/flip_probe.py: DeepID2+, one step -- all 290 entries of
    local3's increments beyond 1e-4 x max sit in two (module, filter) columns of one pool window), and that moves one image's term
    of one filter: a discrete difference of about 1/minibatch of a column.  So: at most `outlier_frac` of the entries beyond the
    bound, none beyond `outlier_factor` times it."""
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    bound = tol * float(np.max(np.abs(want))) + atol
    frac = float((d > bound).mean())
    assert frac <= outlier_frac and float(d.max()) <= outlier_factor * bound, \
        '%s: max abs err %.3e (bound %.3e), %.4f %% of the entries beyond the bound' % (what, float(d.max()), bound, 100 * frac)


def dev(a):
    return capi.DeviceBuffer(array=np.ascontiguousarray(a, dtype=np.float32))


def dev_empty(shape, fill=None):
    b = capi.DeviceBuffer(shape=shape)
    if fill is not None:
        b.upload(np.full(shape, fill, np.float32))
    return b


def call(name, *args):
    conv = []
    protos = capi.lib().protos[name][1]
    for (typ, _), a in zip(protos, args):
        if typ == 'float':
            a = C.c_float(a)
        conv.append(a)
    return capi.lib().call(name, *conv)


def ws_bytes(name, *args):
    return int(capi.lib().raw(name)(*args))


def dev_double(n=1):
    b = capi.DeviceBuffer(nbytes=8 * n)
    b.dtype = np.dtype(np.float64)
    b.shape = (n,)
    b.zero()
    return b
