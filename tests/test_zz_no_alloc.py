"""The allocation-free entry point mcn_conv_weight_acts_reduced_static (static device scratch instead of a caller
workspace; no counterpart in the reference) against mcn_conv_weight_acts_reduced on the same inputs, through the C ABI.

Two calls in a row on one GPU, the second with a smaller input: the static array still holds the first call's
partials, which the second must not see.  Both entry points run the same kernels in the same order, so the results
are compared bit for bit (stricter than the 1e-4 fp32 bar of the other kernel tests); the first result is also held to
the CPU oracle at that bar.
"""
import ctypes as C

import numpy as np
import pytest

from kutil import call, close, dev, ws_bytes, capi
from oracle import oracle as O

pytestmark = pytest.mark.gpu

# (imgY, imgX, C, F, fs, mY, mX, pad, stride, groups), N
BIG = ((55, 55, 3, 16, 4, 52, 52, 0, 1, 1), 128)      # conv1 of DeepID2: strip form, 468 strips of partials
SMALL = ((18, 14, 16, 32, 3, 16, 12, 0, 1, 1), 96)    # conv2 at 39x31, ragged batch: per-module form, fewer chunks


def run_both(geom, N, seed, sT):
    imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups = geom
    rng = np.random.default_rng(seed)
    img = (rng.standard_normal((C_ * imgY * imgX, N))).astype(np.float32)
    hid = (rng.standard_normal((F * mY * mX, N))).astype(np.float32)
    K = (C_ // groups) * fs * fs
    g0 = (rng.standard_normal((K, F)) * 0.01).astype(np.float32)
    d_img, d_hid = dev(img), dev(hid)
    nbytes = ws_bytes('mcn_conv_weight_acts_reduced_workspace_bytes', C_, F, fs, mY, mX, groups)
    assert nbytes <= ws_bytes('mcn_conv_weight_acts_reduced_static_max_bytes')
    wsb = capi.workspace(nbytes)
    d_ref = dev(g0)
    call('mcn_conv_weight_acts_reduced', d_img, d_hid, d_ref, wsb, nbytes, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups,
         sT, 5e-4 / N, None)
    d_out = dev(g0)
    call('mcn_conv_weight_acts_reduced_static', d_img, d_hid, d_out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, groups,
         sT, 5e-4 / N, None)
    got, ref = d_out.numpy(), d_ref.numpy()
    kw = dict(imgY=imgY, imgX=imgX, C_=C_, F=F, fs=fs, mY=mY, mX=mX, pad=pad, stride=stride, groups=groups)
    tmp = O.weight_acts(img, hid, N, partial_sum=1, conv=True, sO=5e-4 / N, **kw)
    want = g0.copy().reshape(-1)
    O.add_sum(want, tmp, mY * mX, K * F, 0, sT, 1.0)
    return got, ref, want.reshape(K, F)


def test_static_scratch_matches_workspace_entry_point_twice():
    got, ref, want = run_both(*BIG, seed=31, sT=0.0)
    assert np.array_equal(got, ref), 'first call (large input) differs from mcn_conv_weight_acts_reduced'
    close(got, want, what='static scratch, large input vs oracle')
    got, ref, want = run_both(*SMALL, seed=32, sT=1.0)
    assert np.array_equal(got, ref), 'second call (smaller input) differs from mcn_conv_weight_acts_reduced'
    close(got, want, what='static scratch, smaller input vs oracle')


def test_static_scratch_rejects_oversized_shapes():
    L = capi.lib()
    d = dev(np.zeros(16, np.float32))
    # 256 channels x 5x5 x 256 filters over 64x64 modules: the partials alone exceed 4 MiB
    rc = L.raw('mcn_conv_weight_acts_reduced_static')(d.ptr, d.ptr, d.ptr, 128, 128, 68, 68, 256, 256, 5, 64, 64, 0, 1, 1,
                                                      C.c_float(0), C.c_float(1), None)
    assert rc == -3 and b'static scratch' in L.raw('mcn_last_error')()
