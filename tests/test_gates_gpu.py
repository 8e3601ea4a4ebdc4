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


@pytest.mark.parametrize('O_,I,N', [(10575, 160, 128), (10575, 512, 128), (4213, 160, 72), (2500, 96, 4)])
def test_classifier_backward_data_adds_the_l2sn_gradient_behind_a_gate(O_, I, N):
    """mcn_fc_bprop_acts_l2sn: W^T dY through the K-split form, the L2-SN gradient of the same matrix added by the reduce pass.
    Bitwise what mcn_gemm(transA) + mcn_l2sn_grad_from_stats(add = 1) leave; with a gate the statistics are produced on ANOTHER
    stream, queued AFTER the product, and the reduce pass waits for the signal itself."""
    from kutil import ws_bytes, dev_double
    L, sa, sb = streams()
    g = gate_alloc(L)
    rng = np.random.default_rng(77)
    W = dev((rng.standard_normal((O_, I)) * 0.01).astype(np.float32))
    dY = dev((rng.standard_normal((O_, N)) * 0.1).astype(np.float32))
    feats = (rng.standard_normal((I, N)) * 0.3).astype(np.float32)
    h = N // 2
    ids = rng.integers(0, 50, size=N)
    same = rng.random(h) < 0.5
    ids[h:][same] = ids[:h][same]
    d_feats, d_ids = dev(feats), dev(ids.astype(np.float32).reshape(1, N))
    wsb = ws_bytes('mcn_gemm_workspace_bytes', I, N, O_, 1, 0)
    ws, ws2 = dev_empty((max(wsb, 16) // 4,)), dev_empty((max(wsb, 16) // 4,))
    assert L.raw('mcn_fc_bprop_acts_l2sn_supported')(W.ptr, dY.ptr, d_feats.ptr, I, N, O_) == 1
    coeff = 0.0025

    def stats_on(stream):
        st, out, info, cost = dev_empty(2 * h, np.nan), dev_empty((1, h), np.nan), dev_empty(4, np.nan), dev_double(1)
        call('mcn_l2sn_pair_stats', d_ids, d_feats, st, I, N, stream)
        call('mcn_l2sn_cost_from_stats', st, out, N, -1.0, info, cost, None, None, 0, stream)
        return st, info, (out, cost)

    # reference: three launches in stream order
    st, info, keep = stats_on(None)
    want = dev_empty((I, N), np.nan)
    call('mcn_gemm', W, dY, want, I, N, O_, 1, 0, 1.0, 0.0, ws, wsb, None)
    call('mcn_l2sn_grad_from_stats', d_feats, want, I, N, st, info, 1, coeff, None)
    L.call('mcn_device_sync')
    # (a) same stream, no gate
    got = dev_empty((I, N), np.nan)
    call('mcn_fc_bprop_acts_l2sn', W, dY, got, I, N, O_, ws, wsb, d_feats, st, info, coeff, None, 0, 0, None)
    L.call('mcn_device_sync')
    assert np.array_equal(got.numpy(), want.numpy())
    # (b) the product on stream A first, the statistics on stream B afterwards, then the signal
    for rep in (1, 2):
        got2 = dev_empty((I, N), np.nan)
        st2, info2 = dev_empty(2 * h, np.nan), dev_empty(4, np.nan)
        out2, cost2 = dev_empty((1, h), np.nan), dev_double(1)  # (before the launch: an upload on the null stream waits for stream A)
        L.call('mcn_device_sync')
        call('mcn_fc_bprop_acts_l2sn', W, dY, got2, I, N, O_, ws2, wsb, d_feats, st2, info2, coeff, g, 7 * rep, 2000, sa)
        call('mcn_l2sn_pair_stats', d_ids, d_feats, st2, I, N, sb)
        call('mcn_l2sn_cost_from_stats', st2, out2, N, -1.0, info2, cost2, None, None, 0, sb)
        L.call('mcn_stream_signal', g, 7 * rep, sb)
        L.call('mcn_device_sync')
        assert np.array_equal(got2.numpy(), want.numpy()), 'the reduce pass ran ahead of the signal'
    t = C.c_int(-1)
    L.call('mcn_gate_timed_out', g, C.byref(t), sa)
    assert t.value == 0
    L.call('mcn_gate_free', g)
    L.call('mcn_stream_destroy', sa)
    L.call('mcn_stream_destroy', sb)
