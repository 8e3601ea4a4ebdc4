"""Stream gates (mcn_stream_signal / mcn_stream_gate): cross-stream ordering on one device by a spinning one-thread kernel
instead of a barrier packet.  No counterpart in the reference (single stream, convnet.cu:139-156)."""
import ctypes as C

import numpy as np
import pytest

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import capi  # noqa: E402
from kutil import call, dev, dev_empty  # noqa: E402

pytestmark = pytest.mark.gpu


def streams():
    L = capi.lib()
    a, b = C.c_void_p(), C.c_void_p()
    L.call('mcn_stream_create', C.byref(a))
    L.call('mcn_stream_create', C.byref(b))
    return L, a, b


def gate_alloc(L):
    g = C.c_void_p()
    L.call('mcn_gate_alloc', C.byref(g))
    return C.cast(g, C.POINTER(C.c_int))


def test_gate_orders_two_streams_and_a_lost_signal_times_out():
    L, sa, sb = streams()
    g = gate_alloc(L)
    ok = C.c_int(-1)
    L.call('mcn_gate_selftest', g, sa, sb, C.byref(ok))
    assert ok.value == 1, 'two fresh streams of one device must make progress independently'
    # consumer first: stream A gates on 5, then doubles x; producer: stream B writes x = 3 and signals 5
    n = 1 << 20
    x = dev_empty((n,), 0.0)
    three = dev(np.full(n, 3.0, np.float32))
    for rep in range(1, 6):
        L.call('mcn_memset', x, 0, n * 4, None)
        L.call('mcn_device_sync')
        L.call('mcn_stream_gate', g, 5 * rep, 2000, sa)
        call('mcn_scale', x, x, n, 2.0, sa)
        call('mcn_scale', three, x, n, 1.0, sb)
        L.call('mcn_stream_signal', g, 5 * rep, sb)
        L.call('mcn_device_sync')
        assert np.all(x.numpy() == 6.0), 'the gated stream ran ahead of the signal'
    t = C.c_int(-1)
    L.call('mcn_gate_timed_out', g, C.byref(t), sa)
    assert t.value == 0
    # a gate whose signal never comes ends by itself and says so
    L.call('mcn_stream_gate', g, 1000, 30, sa)
    L.call('mcn_gate_timed_out', g, C.byref(t), sa)
    assert t.value == 1
    L.call('mcn_gate_free', g)
    L.call('mcn_stream_destroy', sa)
    L.call('mcn_stream_destroy', sb)


def test_one_stream_is_reported_as_not_concurrent():
    L, sa, sb = streams()
    g = gate_alloc(L)
    ok = C.c_int(-1)
    with pytest.raises(Exception):
        L.call('mcn_gate_selftest', g, sa, sa, C.byref(ok))  # the same stream twice: refused, not deadlocked
    L.call('mcn_gate_free', g)
    L.call('mcn_stream_destroy', sa)
    L.call('mcn_stream_destroy', sb)
