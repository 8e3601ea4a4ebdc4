#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C ABI at the DeepID2 layer shapes (N = 128): HIP-event time per call,
algorithmic TFLOP/s and GB/s.  python tools/kbench.py [filter]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ccn_loader  # noqa: E402

ccn_loader.load()
from cuda_convnet_amd import capi  # noqa: E402
from kutil import call, dev, dev_empty, ws_bytes  # noqa: E402

N = 128
# name: (imgY, imgX, C, F, fs, mY, mX, pad, stride, conv)
CONVS = {
    'conv1': (55, 55, 3, 16, 4, 52, 52, 0, 1, True),
    'conv2': (26, 26, 16, 32, 3, 24, 24, 0, 1, True),
    'local3': (12, 12, 32, 48, 3, 10, 10, 0, 1, False),
    'local4': (5, 5, 48, 64, 2, 4, 4, 0, 1, False),
    'conv1+': (55, 55, 3, 32, 4, 52, 52, 0, 1, True),
    'conv2+': (26, 26, 32, 128, 3, 24, 24, 0, 1, True),
    'local3+': (12, 12, 128, 128, 3, 10, 10, 0, 1, False),
    'local4+': (5, 5, 128, 128, 2, 4, 4, 0, 1, False),
}
FCS = {'fc5a': (160, 1024), 'fc5b': (160, 1200), 'fc6': (10575, 160), 'fc6+': (10575, 512)}


def timeit(fn, reps=30):
    L = capi.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.call('mcn_event_create', C.byref(e0))
    L.call('mcn_event_create', C.byref(e1))
    for _ in range(3):
        fn()
    L.call('mcn_device_sync')
    L.call('mcn_event_record', e0, None)
    for _ in range(reps):
        fn()
    L.call('mcn_event_record', e1, None)
    L.call('mcn_event_sync', e1)
    ms = C.c_float()
    L.call('mcn_event_elapsed_ms', C.byref(ms), e0, e1)
    return ms.value * 1e3 / reps


OPF = sys.argv[2] if len(sys.argv) > 2 else ''


def report(name, fn, flops, nbytes):
    if OPF and OPF not in name:
        return
    us = timeit(fn)
    print('%-28s %8.2f us  %7.2f TFLOP/s  %7.1f GB/s' % (name, us, flops / us / 1e6, nbytes / us / 1e3), flush=True)


def skip(flt, name, family):
    """layer filter: substring of the name (or of the family: 'conv' / 'fc' select all); 'conv2$' = exactly conv2"""
    if flt.endswith('$'):
        return name != flt[:-1]
    return bool(flt) and flt not in name and flt not in family


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    rng = np.random.default_rng(0)
    ws = capi.workspace(64 << 20)
    for name, (imgY, imgX, C_, F, fs, mY, mX, pad, stride, conv) in CONVS.items():
        if skip(flt, name, 'conv'):
            continue
        M, K = mY * mX, C_ * fs * fs
        img = dev(rng.standard_normal((C_ * imgY * imgX, N)).astype(np.float32))
        W = dev((rng.standard_normal((K * (1 if conv else M), F)) * 0.04).astype(np.float32))
        hid = dev(rng.standard_normal((F * M, N)).astype(np.float32))
        b = dev(np.zeros(F if conv else F * M, np.float32))
        out, dimg, dW = dev_empty((F * M, N)), dev_empty((C_ * imgY * imgX, N)), dev_empty(W.shape)
        flops = 2.0 * N * M * F * K
        nbytes = 4.0 * (img.nbytes / 4 + W.nbytes / 4 + hid.nbytes / 4)
        pre = 'mcn_conv_' if conv else 'mcn_local_'
        report(name + ' filterActs+bias+relu', (lambda: call('mcn_filter_acts_bias_neuron', img, W, b, out, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, 1, int(conv), 0 if conv else 1, 1, 0.0, 0.0, None)), flops, nbytes)
        if conv and mY % 2 == 0 and mX % 2 == 0:
            pout = dev_empty((F * (mY // 2) * (mX // 2), N))
            report(name + ' maxPool 2x2', (lambda: call('mcn_local_pool', out, pout, N, F, mY, mX, 2, 0, 2, mY // 2, mX // 2, 0, None)), 0.0, 1.25 * hid.nbytes)
            report(name + ' filterActs+bias+relu+pool', (lambda: call('mcn_filter_acts_bias_neuron_pool', img, W, b, out, pout, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, 1, int(conv), 0 if conv else 1, 1, 0.0, 0.0, None)), flops, nbytes)
        if C_ > 3:
            report(name + ' imgActs', (lambda: call(pre + 'img_acts', hid, W, dimg, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, 1, 0.0, 1.0, None)), flops, nbytes)
        if C_ > 3:  # ... and on through a 2x2 max pool + relu below (the DeepID stacks)
            cY, cX = 2 * imgY, 2 * imgX
            acts = dev(np.maximum(rng.standard_normal((C_ * cY * cX, N)), 0).astype(np.float32))
            pooled = dev_empty((C_ * imgY * imgX, N))
            call('mcn_local_pool', acts, pooled, N, C_, cY, cX, 2, 0, 2, imgY, imgX, 0, None)
            dact = dev_empty((C_ * cY * cX, N))
            ub = 4.0 * (2 * acts.nbytes / 4 + 2 * pooled.nbytes / 4)
            report(name + ' maxUndo+relu', (lambda: call('mcn_local_max_undo_neuron', acts, dimg, pooled, dact, N, C_, cY, cX, 2, 0, 2, imgY, imgX, 1, 0.0, 0.0, None)), 0.0, ub)
            report(name + ' imgActs+maxUndo+relu', (lambda: call('mcn_img_acts_max_undo', hid, W, None, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, 1, int(conv), 0.0, 1.0, acts, pooled, dact, 1, 0.0, 0.0, None)), flops, nbytes + ub)
        if conv:
            report(name + ' weightActs(reduced)', (lambda: call('mcn_conv_weight_acts_reduced', img, hid, dW, ws, ws.nbytes, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, 1, 0.0, 1.0, None)), flops, nbytes)
        else:
            report(name + ' weightActs', (lambda: call('mcn_local_weight_acts', img, hid, dW, N, N, imgY, imgX, C_, F, fs, mY, mX, pad, stride, 1, 0.9, 1.0, None)), flops, nbytes)
    for name, (O_, I) in FCS.items():
        if skip(flt, name, 'fc'):
            continue
        W = dev((rng.standard_normal((O_, I)) * 0.01).astype(np.float32))
        X = dev(rng.standard_normal((I, N)).astype(np.float32))
        dY = dev(rng.standard_normal((O_, N)).astype(np.float32))
        b = dev(np.zeros(O_, np.float32))
        Y, dX, dW = dev_empty((O_, N)), dev_empty((I, N)), dev_empty((O_, I))
        flops, nbytes = 2.0 * N * I * O_, 4.0 * (O_ * I + I * N + O_ * N)
        report(name + ' fwd (NN)', (lambda: call('mcn_gemm_bias_neuron', W, X, b, Y, O_, N, I, 0.0, 0, 0.0, 0.0, ws, ws.nbytes, None)), flops, nbytes)
        report(name + ' bwd-data (TN)', (lambda: call('mcn_gemm', W, dY, dX, I, N, O_, 1, 0, 1.0, 0.0, ws, ws.nbytes, None)), flops, nbytes)
        if O_ >= 512 and N % 4 == 0:  # the classifier under a softmax + logreg cost: stage 1 in the product, stage 2 + gradient + cost counts
            nb = ws_bytes('mcn_softmax_workspace_bytes', O_, N)
            part = capi.workspace(nb)
            sb = ws_bytes('mcn_softmax_logreg_cost_state_bytes', N)
            state = capi.workspace(sb)
            call('mcn_memset', state, 0, sb, None)
            labels = dev(rng.integers(0, O_, (1, N)).astype(np.float32))
            probs, grad, lp, cp = dev_empty((O_, N)), dev_empty((O_, N)), dev_empty((1, N)), dev_empty((1, N))
            from kutil import dev_double
            cost = dev_double(2)
            call('mcn_gemm_bias_neuron_softmax_part', W, X, b, Y, O_, N, I, 0, 0.0, 0.0, part, nb, None)
            report(name + ' fwd + softmax stage 1', (lambda: call('mcn_gemm_bias_neuron_softmax_part', W, X, b, Y, O_, N, I, 0, 0.0, 0.0, part, nb, None)), flops, nbytes)
            report(name + ' softmax+grad+cost', (lambda: call('mcn_softmax_logreg_cost', Y, probs, labels, grad, 1.0, O_, N, part, nb, ws, ws.nbytes, lp, cp, cost, state, sb, 1, None)), 4.0 * N * O_, 4.0 * 3 * N * O_)
            report(name + ' softmax+grad+cost w/o finish', (lambda: call('mcn_softmax_logreg_cost', Y, probs, labels, grad, 1.0, O_, N, part, nb, ws, ws.nbytes, lp, cp, cost, state, sb, 0, None)), 4.0 * N * O_, 4.0 * 3 * N * O_)
            call('mcn_softmax_logreg_cost_finish', lp, cp, cost, state, sb, N, None)
        report(name + ' bwd-weights (NT)', (lambda: call('mcn_gemm', dY, X, dW, O_, I, N, 0, 1, 0.001, 0.9, ws, ws.nbytes, None)), flops, nbytes)


if __name__ == '__main__':
    main()
