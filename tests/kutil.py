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
