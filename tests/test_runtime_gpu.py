"""Runtime entry points of the C ABI that the layer graph's scheduling rests on: stream ticks (mcn_stream_tick)."""
import ctypes as C

import numpy as np
import pytest

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import capi  # noqa: E402
from kutil import call, dev  # noqa: E402

pytestmark = pytest.mark.gpu


def test_stream_ticks_arrive_in_stream_order():
    L = capi.lib()
    flag = C.c_void_p()
    L.call('mcn_tick_alloc', C.byref(flag))
    view = C.cast(flag.value, C.POINTER(C.c_int))
    assert view[0] == 0
    stream = C.c_void_p()
    L.call('mcn_stream_create', C.byref(stream))
    a = dev(np.ones((1 << 20,), np.float32))
    try:
        for k in range(1, 50):
            call('mcn_scale', a, a, 1 << 20, 1.0001, stream)           # work ahead of the tick
            call('mcn_stream_tick', C.cast(flag.value, C.POINTER(C.c_int)), k, stream)
        L.call('mcn_stream_sync', stream)
        assert view[0] == 49                                            # the last tick of the stream
        call('mcn_stream_tick', C.cast(flag.value, C.POINTER(C.c_int)), 7, stream)
        L.call('mcn_stream_sync', stream)
        assert view[0] == 7                                             # a plain store: callers keep their ticks increasing
    finally:
        L.call('mcn_stream_destroy', stream)
        L.call('mcn_tick_free', C.cast(flag.value, C.POINTER(C.c_int)))


def test_armed_ticks_ride_on_the_next_launch():
    """mcn_arm_tick: the tick is stored by the next kernel itself where it has a tick slot (imgActs here), by a one-thread kernel
    behind any other launch, and by mcn_flush_tick when nothing is launched; an armed tick goes out before a later one."""
    L = capi.lib()
    flag = C.c_void_p()
    L.call('mcn_tick_alloc', C.byref(flag))
    fl = C.cast(flag.value, C.POINTER(C.c_int))
    a = dev(np.ones((4096,), np.float32))
    rng = np.random.default_rng(5)
    N, img, C_, F, fs = 64, 6, 16, 16, 3
    m = img - fs + 1
    hid = dev(rng.standard_normal((F * m * m, N)).astype(np.float32))
    w = dev(rng.standard_normal((C_ * fs * fs, F)).astype(np.float32))
    tgt = capi.DeviceBuffer(shape=(C_ * img * img, N))
    try:
        call('mcn_arm_tick', fl, 11, None)
        assert fl[0] == 0                                   # nothing launched yet
        call('mcn_conv_img_acts', hid, w, tgt, N, img, img, C_, F, fs, m, m, 0, 1, 1, 0.0, 1.0, None)   # a kernel with a tick slot
        L.call('mcn_device_sync')
        assert fl[0] == 11
        assert np.all(np.isfinite(tgt.numpy()))
        call('mcn_arm_tick', fl, 12, None)
        call('mcn_scale', a, a, 4096, 2.0, None)            # no tick slot: a tick kernel follows it
        L.call('mcn_device_sync')
        assert fl[0] == 12
        call('mcn_arm_tick', fl, 13, None)
        L.call('mcn_flush_tick')
        L.call('mcn_device_sync')
        assert fl[0] == 13
        call('mcn_arm_tick', fl, 14, None)
        call('mcn_stream_tick', fl, 15, None)               # the armed one first, then this one
        L.call('mcn_device_sync')
        assert fl[0] == 15
        L.call('mcn_flush_tick')                            # nothing armed: no effect
        L.call('mcn_device_sync')
        assert fl[0] == 15
    finally:
        L.call('mcn_tick_free', fl)
