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
