"""TEST INFRASTRUCTURE ONLY: whole-net CPU oracle.

Restates the reference's layer-graph orchestration (Layer::fprop/bprop
layer.cu:75-158, WeightLayer :258-343, FCLayer :350-384, ConvLayer :438-521,
LocalUnsharedLayer :527-575, SoftmaxLayer :582-607, pool :685-736, norm
:847-885, cost layers :920-1116, ConvNet::fprop/bprop/updateWeights
convnet.cu:190-245, TrainingWorker::run worker.cu:88-102, Weights::update
weights.cuh:150-163) over numpy arrays, calling the C oracle kernels.

It is deliberately independent of the product's C++ layer graph so that the
two can be compared.  All activation matrices are held in the reference's
memory layout, (features, numCases) row-major; the FC "soft transpose"
(layer.cu:98-101) is folded into the GEMM orientation.

PARITY UNPINNED (see convnet_oracle.cpp header); tests/test_oracle_pins.py
holds the soft pins (grad-check < 0.02, log shapes, ln K first cost).
"""
import numpy as np

from . import oracle as K

PASS_TRAIN, PASS_TEST, PASS_GC = 0, 1, 2


class Weights:
    """weights.cuh:41-198, including shared matrices (`Weights(Weights& srcWeights, float epsW)`, :59-68): a sharer
    has its own epsW, weight decay 0, and the owner's momentum / useGrad / device matrices (copyToGPU :141-145);
    the update counter lives in the owner (:165-181) and only the owner updates (:150-163)."""

    def __init__(self, w, inc, eps, wc, mom, use_grad, fortran, src=None):
        # host arrays keep the caller's layout; `W`/`inc` below are the flat memory images
        self.src = src.src if (src is not None and src.src is not None) else src
        self.hW, self.hInc = w, inc
        self.fortran = fortran
        if self.src is not None:
            o = self.src
            self.W, self.inc, self._grad = o.W, o.inc, None
            self.eps, self.wc, self.mom, self.use_grad = eps, 0.0, o.mom, o.use_grad
        else:
            self.W = self._mem(w).copy()
            self.inc = self._mem(inc).copy()
            self._grad = np.zeros_like(self.W) if use_grad else None
            self.eps, self.wc, self.mom, self.use_grad = eps, wc, mom, use_grad
        self._num_updates = 0

    @property
    def grad(self):
        return self.src._grad if self.src is not None else self._grad

    @grad.setter
    def grad(self, g):
        if self.src is not None:
            self.src._grad = g
        else:
            self._grad = g

    @property
    def num_updates(self):  # weights.cuh:176-181
        return self.src._num_updates if self.src is not None else self._num_updates

    @num_updates.setter
    def num_updates(self, v):  # weights.cuh:165-170
        if self.src is not None:
            self.src._num_updates = v
        else:
            self._num_updates = v

    def _mem(self, a):
        return np.ascontiguousarray(a.T) if self.fortran else np.ascontiguousarray(a)

    def get_grad(self, dp=False):
        if dp and self.grad is None:
            self.grad = np.zeros_like(self.W)
        return self.grad if (self.use_grad or dp) else self.inc

    def update(self, comm=None):
        if self.src is None and self.eps > 0:  # "Only true owner of weights updates" (weights.cuh:152)
            if comm is not None:  # data-parallel: raw scaled gradients are summed over the replicas first
                comm.all_reduce(self.get_grad(True))
            K.sgd_update(self.W, self.inc, self.grad, self.mom, self.wc, self.eps, self.use_grad or comm is not None)
            self.num_updates = 0

    def copy_to_host(self):
        if self.src is not None:  # weights.cuh:124-130
            return
        self.hW[...] = self.W.T if self.fortran else self.W
        self.hInc[...] = self.inc.T if self.fortran else self.inc


class Layer:
    def __init__(self, net, d):
        self.net, self.d = net, d
        self.name, self.type = d['name'], d['type']
        self.outputs = d['outputs']
        self.grad_consumer = bool(d['gradConsumer'])
        self.acts_target, self.acts_grad_target = d['actsTarget'], d['actsGradTarget']
        self.prev, self.next = [], []
        self.num_grad_producers_next = 0
        self._found = False
        self.acts = None
        self._acts_grad = None
        self.rcvd_f = self.rcvd_b = 0

    # -- graph ---------------------------------------------------------
    def is_grad_consumer(self):
        if not self._found:
            for p in self.prev:
                self.grad_consumer |= p.is_grad_consumer()
            self._found = True
        return self.grad_consumer

    def is_grad_producer(self):
        return True

    def get_acts_grad(self):
        if self.acts_grad_target >= 0:
            return self.prev[self.acts_grad_target].get_acts_grad()
        n = self.net.num_cases
        if self._acts_grad is None or self._acts_grad.shape != (self.outputs, n):
            self._acts_grad = np.zeros((self.outputs, n), np.float32)
        return self._acts_grad

    def reset(self):
        self.rcvd_f = self.rcvd_b = 0

    # -- forward (layer.cu:75-114) --------------------------------------
    def fprop_one(self, pt):
        self.rcvd_f += 1
        if self.rcvd_f == len(self.prev):
            self.fprop([p.acts for p in self.prev], pt)

    def fprop(self, inputs, pt):
        self.inputs = inputs
        n = self.net.num_cases
        if self.acts_target >= 0:
            self.acts = inputs[self.acts_target]
        elif self.acts is None or self.acts.shape != (self.outputs, n) or any(self.acts is i for i in inputs):
            self.acts = np.zeros((self.outputs, n), np.float32)
        self.rcvd_f = len(self.prev)
        if self.acts_target >= 0:
            self.fprop_acts(self.acts_target, 0, pt)
        for i in range(len(self.prev)):
            if i != self.acts_target:
                self.fprop_acts(i, 1 if (self.acts_target >= 0 or i > 0) else 0, pt)
        for nx in self.next:
            nx.fprop_one(pt)

    # -- backward (layer.cu:116-158) ------------------------------------
    def bprop(self, pt):
        if self.rcvd_b == self.num_grad_producers_next:
            self.rcvd_b += 1
            self.bprop_v(self.get_acts_grad(), pt)

    def bprop_v(self, v, pt):
        self.bprop_common(v, pt)
        if self.is_grad_producer():
            for i, p in enumerate(self.prev):
                if p.is_grad_consumer() and self.acts_grad_target != i:
                    self.bprop_acts(v, i, 1 if p.rcvd_b > 0 else 0, pt)
                    p.rcvd_b += 1
            t = self.acts_grad_target
            if t >= 0 and self.prev[t].is_grad_consumer():
                self.bprop_acts(v, t, 1 if self.prev[t].rcvd_b > 0 else 0, pt)
                self.prev[t].rcvd_b += 1
            for p in self.prev:
                if p.is_grad_consumer():
                    p.bprop(pt)

    def bprop_common(self, v, pt):
        pass

    def fprop_acts(self, i, scale_targets, pt):
        pass

    def bprop_acts(self, v, i, scale_targets, pt):
        pass

    def update_weights(self):
        pass

    def copy_to_host(self):
        pass


class DataLayer(Layer):
    def is_grad_producer(self):
        return False

    def fprop_data(self, data, pt):
        self.acts = data[self.d['dataIdx']]
        for nx in self.next:
            nx.fprop_one(pt)


class NeuronLayer(Layer):
    def __init__(self, net, d):
        super().__init__(net, d)
        nd = d['neuron']
        self.ntype = nd['type']
        self.a = float(nd['params'].get('a', 0.0))
        self.b = float(nd['params'].get('b', 0.0))

    def fprop_acts(self, i, st, pt):
        if self.ntype == 'ident' and self.acts is self.inputs[0]:
            return
        K.neuron_fwd(self.inputs[0], self.ntype, self.a, self.b, out=self.acts)

    def bprop_acts(self, v, i, st, pt):
        tgt = self.prev[0].get_acts_grad()
        if self.ntype == 'ident' and tgt is v:
            return
        K.neuron_bwd(v, self.acts, self.inputs[0], self.ntype, self.a, self.b, target=tgt, add=st > 0)


class WeightLayer(Layer):
    def __init__(self, net, d, use_grad, fortran):
        super().__init__(net, d)
        self.weights = []
        for i, (w, wi, e, c, m) in enumerate(zip(d['weights'], d['weightsInc'], d['epsW'], d['wc'], d['momW'])):
            li, mi = d['weightSourceLayerIndices'][i], d['weightSourceMatrixIndices'][i]
            src = None
            if li == len(net.layers):  # the layer being built (layer.cu:280-281)
                src = self.weights[mi]
            elif li >= 0:  # layer.cu:282-285
                src = net.layers[li].weights[mi]
            self.weights.append(Weights(w, wi, e, c, m, use_grad, fortran, src=src))
        self.biases = Weights(d['biases'], d['biasesInc'], d['epsB'], 0.0, d['momB'], True, False)
        # biases are vectors: their memory image is layout independent
        self.biases.W = np.ascontiguousarray(d['biases']).reshape(-1).copy()
        self.biases.inc = np.ascontiguousarray(d['biasesInc']).reshape(-1).copy()
        self.biases.grad = np.zeros_like(self.biases.W)
        self.w_step, self.b_step = 0.001, 0.002  # layer.cu:294-295

    def bprop_common(self, v, pt):  # layer.cu:306-317
        if self.biases.eps > 0:
            self.bprop_biases(v, pt)
        for i, w in enumerate(self.weights):
            if w.eps > 0:
                self.bprop_weights(v, i, pt)
                w.num_updates += 1

    def grad_scale(self, eps, pt):
        """eps / (numCases * replicas); 1 for gradient checking (layer.cu:370,379)."""
        return 1.0 if pt == PASS_GC else eps / (self.net.num_cases * self.net.world)

    def update_weights(self):
        for w in self.weights:
            w.update(self.net.comm)
        self.biases.update(self.net.comm)

    def copy_to_host(self):
        for w in self.weights:
            w.copy_to_host()
        self.d['biases'].reshape(-1)[...] = self.biases.W
        self.d['biasesInc'].reshape(-1)[...] = self.biases.inc


class FCLayer(WeightLayer):
    def __init__(self, net, d):
        super().__init__(net, d, False, True)
        self.w_step, self.b_step = 0.1, 0.01  # layer.cu:351-352

    def fprop_acts(self, i, st, pt):  # layer.cu:355-360
        n, I = self.net.num_cases, self.prev[i].outputs
        K.gemm(self.weights[i].W, self.inputs[i], self.outputs, n, I, beta=float(st), Cmat=self.acts)
        if st == 0:
            K.add_vector(self.acts, self.biases.W, self.outputs, n, True)

    def bprop_acts(self, v, i, st, pt):  # layer.cu:362-366
        n, I = self.net.num_cases, self.prev[i].outputs
        K.gemm(self.weights[i].W, v, I, n, self.outputs, ta=True, beta=float(st), Cmat=self.prev[i].get_acts_grad())

    def bprop_biases(self, v, pt):  # layer.cu:368-372
        n = self.net.num_cases
        K.add_sum(self.biases.grad, v, self.outputs, n, 1, 0.0, self.grad_scale(self.biases.eps, pt))

    def bprop_weights(self, v, i, pt):  # layer.cu:374-384
        n, I = self.net.num_cases, self.prev[i].outputs
        w = self.weights[i]
        dp = self.net.comm is not None
        if dp:  # raw gradient; momentum is folded in after the exchange (algebraically identical)
            scale_inc = 1.0 if w.num_updates > 0 else 0.0
        else:
            scale_inc = (1.0 if (w.num_updates == 0 and pt != PASS_GC) else 0.0) * w.mom
        K.gemm(v, self.prev[i].acts, self.outputs, I, n, tb=True, alpha=self.grad_scale(w.eps, pt), beta=scale_inc,
               Cmat=w.get_grad(dp))


class LocalLayer(WeightLayer):
    def __init__(self, net, d, use_grad, conv):
        super().__init__(net, d, use_grad, False)
        self.conv = conv

    def geo(self, i):
        d = self.d
        return dict(imgY=d['imgSizeY'][i], imgX=d['imgSizeX'][i], C_=d['channels'][i], F=d['filters'],
                    fs=d['filterSize'][i], mY=d['modulesY'], mX=d['modulesX'], pad=d['padding'][i],
                    stride=d['stride'][i], groups=d['groups'][i])

    def sparse(self, i):
        return bool(self.d.get('randSparse', [False] * (i + 1))[i])

    def sgeo(self, i):
        """geometry + the channel table of a random-sparse input (layer.cu:410-431)"""
        g = self.geo(i)
        g['fC'] = self.d['filterChannels'][i]
        return np.asarray(self.d['filterConns'][i], np.int32), g

    def fprop_acts(self, i, st, pt):  # layer.cu:443-462, :530-543
        d, n = self.d, self.net.num_cases
        if self.sparse(i):
            conns, g = self.sgeo(i)
            K.filter_acts_sparse(self.inputs[i], self.weights[i].W, conns, n, targets=self.acts, sT=float(st), sO=1.0, conv=self.conv, **g)
        else:
            K.filter_acts(self.inputs[i], self.weights[i].W, n, targets=self.acts, sT=float(st), sO=1.0,
                          conv=self.conv, **self.geo(i))
        if st == 0:
            if self.conv and d['sharedBiases']:
                K.add_vector(self.acts, self.biases.W, d['filters'], d['modules'] * n, True)
            else:
                K.add_vector(self.acts, self.biases.W, d['filters'] * d['modules'], n, True)

    def bprop_biases(self, v, pt):  # layer.cu:464-474, :545-549
        d, n = self.d, self.net.num_cases
        sc = self.grad_scale(self.biases.eps, pt)
        if self.conv and d['sharedBiases']:
            K.add_sum(self.biases.grad, v, d['filters'], d['modules'] * n, 1, 0.0, sc)
        else:
            K.add_sum(self.biases.grad, v, d['filters'] * d['modules'], n, 1, 0.0, sc)

    def bprop_acts(self, v, i, st, pt):  # layer.cu:498-513, :566-575
        if self.sparse(i):
            conns, g = self.sgeo(i)
            K.img_acts_sparse_summed(v, self.weights[i].W, conns, self.net.num_cases, targets=self.prev[i].get_acts_grad(),
                                     sT=float(st), sO=1.0, conv=self.conv, **g)
            return
        K.img_acts(v, self.weights[i].W, self.net.num_cases, targets=self.prev[i].get_acts_grad(),
                   sT=float(st), sO=1.0, conv=self.conv, **self.geo(i))

    def weight_acts(self, i, *a, **kw):
        if self.sparse(i):
            conns, g = self.sgeo(i)
            kw.update(g)
            return K.weight_acts_sparse(a[0], a[1], conns, *a[2:], **kw)
        kw.update(self.geo(i))
        return K.weight_acts(*a, **kw)


class ConvLayer(LocalLayer):
    def __init__(self, net, d):
        super().__init__(net, d, True, True)

    def bprop_weights(self, v, i, pt):  # layer.cu:476-496
        d, n = self.d, self.net.num_cases
        w = self.weights[i]
        ps = d['partialSum']
        scale_w = self.grad_scale(w.eps, pt)
        if ps > 0:
            tmp = self.weight_acts(i, self.prev[i].acts, v, n, partial_sum=ps, sT=0.0, sO=scale_w, conv=True)
            rows = d['modules'] // ps
            K.add_sum(w.grad.reshape(-1), tmp, rows, tmp.size // rows, 0, 1.0 if w.num_updates > 0 else 0.0, 1.0)
        else:
            self.weight_acts(i, self.prev[i].acts, v, n, partial_sum=0, targets=w.grad,
                             sT=1.0 if w.num_updates > 0 else 0.0, sO=scale_w, conv=True)


class LocalUnsharedLayer(LocalLayer):
    def __init__(self, net, d):
        super().__init__(net, d, False, False)

    def bprop_weights(self, v, i, pt):  # layer.cu:551-564
        n = self.net.num_cases
        w = self.weights[i]
        dp = self.net.comm is not None
        if dp:
            scale_inc = 1.0 if w.num_updates > 0 else 0.0
        else:
            scale_inc = (1.0 if (pt != PASS_GC and w.num_updates == 0) else 0.0) * w.mom
        self.weight_acts(i, self.prev[i].acts, v, n, targets=w.get_grad(dp), sT=scale_inc, sO=self.grad_scale(w.eps, pt), conv=False)


class SoftmaxLayer(Layer):
    def fprop_acts(self, i, st, pt):  # layer.cu:585-595
        self.acts[...] = K.softmax(self.inputs[0], self.outputs, self.net.num_cases)

    def bprop_acts(self, v, i, st, pt):  # layer.cu:597-607
        n = self.net.num_cases
        tgt = self.prev[0].get_acts_grad()
        if len(self.next) == 1 and self.next[0].type == 'cost.logreg':
            labels = self.next[0].prev[0].acts
            K.logreg_softmax_grad(labels, self.acts, self.outputs, n, self.next[0].coeff, target=tgt, add=st == 1)
        else:
            K.softmax_grad(self.acts, v, self.outputs, n, target=tgt, add=st == 1)


class EltwiseSumLayer(Layer):
    def fprop_acts(self, i, st, pt):  # layer.cu:618-624
        c = np.float32(self.d['coeffs'][i])
        if st == 0:
            self.acts[...] = self.inputs[i] * c
        else:
            K.axpby(self.acts, self.inputs[i], 1.0, float(c))

    def bprop_acts(self, v, i, st, pt):  # layer.cu:626-633
        c = np.float32(self.d['coeffs'][i])
        tgt = self.prev[i].get_acts_grad()
        if st == 0:
            tgt[...] = v * c
        else:
            K.axpby(tgt, v, float(st), float(c))


class EltwiseMaxLayer(Layer):
    def fprop_acts(self, i, st, pt):  # layer.cu:643-649
        if i == 1:
            np.maximum(self.inputs[1], self.inputs[0], out=self.acts)
        elif i > 1:
            np.maximum(self.acts, self.inputs[i], out=self.acts)

    def bprop_acts(self, v, i, st, pt):  # layer.cu:651-653, layer_kernels.cu:172-182
        tgt = self.prev[i].get_acts_grad()
        g = v * (self.acts == self.inputs[i]).astype(np.float32)
        if st != 0:
            tgt += g
        else:
            tgt[...] = g


class PoolLayer(Layer):
    def args(self):
        d = self.d
        return (self.net.num_cases, d['channels'], d['imgSizeY'], d['imgSizeX'], d['sizeX'], d['start'], d['stride'],
                d['outputsY'], d['outputsX'])

    def fprop_acts(self, i, st, pt):  # layer.cu:714-716, :730-732
        self.acts[...] = K.local_pool(self.inputs[0], *self.args(), mode=self.d['pool'])

    def bprop_acts(self, v, i, st, pt):  # layer.cu:718-720, :734-736
        tgt = self.prev[0].get_acts_grad()
        if self.d['pool'] == 'max':
            K.local_max_undo(self.prev[0].acts, v, self.acts, *self.args(), target=tgt, sT=float(st), sO=1.0)
        else:
            K.local_avg_undo(v, *self.args(), target=tgt, sT=float(st), sO=1.0)


class NailbedLayer(Layer):
    def args(self):
        d = self.d
        return (self.net.num_cases, d['channels'], d['imgSizeY'], d['imgSizeX'], d['start'], d['stride'], d['outputsY'], d['outputsX'])

    def fprop_acts(self, i, st, pt):  # layer.cu:751-753
        self.acts[...] = K.bed_of_nails(self.inputs[0], *self.args())

    def bprop_acts(self, v, i, st, pt):  # layer.cu:755-757
        K.bed_of_nails_undo(v, *self.args(), target=self.prev[0].get_acts_grad(), sT=float(st), sO=1.0)


class GaussianBlurLayer(Layer):
    def blur(self, x, target=None, st=0.0):
        d, n = self.d, self.net.num_cases
        f = np.ascontiguousarray(d['filter'], np.float32).reshape(-1)
        rows = K.gaussian_blur(x, f, n, d['channels'], d['imgSizeY'], d['imgSizeX'], 1)
        return K.gaussian_blur(rows, f, n, d['channels'], d['imgSizeY'], d['imgSizeX'], 0, target=target, sT=st, sO=1.0)

    def fprop_acts(self, i, st, pt):  # layer.cu:769-772
        self.acts[...] = self.blur(self.inputs[0])

    def bprop_acts(self, v, i, st, pt):  # layer.cu:776-780
        self.blur(v, target=self.prev[0].get_acts_grad(), st=float(st))


class ResizeLayer(Layer):
    def fprop_acts(self, i, st, pt):  # layer.cu:798-800
        d = self.d
        self.acts[...] = K.resize_bilinear(self.inputs[0], self.net.num_cases, d['channels'], d['imgSize'], d['tgtSize'], d['scale'])

    def bprop_acts(self, v, i, st, pt):  # layer.cu:803-805
        raise AssertionError('resize layers cannot propagate gradients')


class RGBToYUVLayer(Layer):
    def fprop_acts(self, i, st, pt):  # layer.cu:815-817
        self.acts[...] = K.rgb_to_yuv(self.inputs[0], self.net.num_cases, self.d['imgPixels'])

    def bprop_acts(self, v, i, st, pt):  # layer.cu:820-822
        raise AssertionError('colour transforms cannot propagate gradients')


class RGBToLABLayer(RGBToYUVLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:833-835
        self.acts[...] = K.rgb_to_lab(self.inputs[0], self.net.num_cases, self.d['imgPixels'], int(self.d['center']))


class ShiftLayer(Layer):
    def fprop_acts(self, i, st, pt):  # layer.cu:1129-1131
        d = self.d
        self.acts[...] = K.shift(self.inputs[0], np.ascontiguousarray(d['filter'], np.float32), self.net.num_cases, d['channels'], d['imgPixels'])

    def bprop_acts(self, v, i, st, pt):  # layer.cu:1135-1139 (empty in the reference)
        raise AssertionError('shift layers cannot propagate gradients')


class ResponseNormLayer(Layer):
    cross = False

    def fprop_acts(self, i, st, pt):  # layer.cu:855-857, :879-881
        d, n = self.d, self.net.num_cases
        if self.cross:
            t, self.denoms = K.response_norm_cross_map(self.inputs[0], n, d['channels'], d['imgPixels'], d['size'],
                                                       d['scale'], d['pow'], d['blocked'])
        else:
            t, self.denoms = K.response_norm(self.inputs[0], n, d['channels'], d['imgSizeY'], d['imgSizeX'], d['size'],
                                             d['scale'], d['pow'])
        self.acts[...] = t

    def bprop_acts(self, v, i, st, pt):  # layer.cu:859-861, :883-885 (destroys own acts)
        d, n = self.d, self.net.num_cases
        tgt = self.prev[0].get_acts_grad()
        if self.cross:
            K.response_norm_cross_map_undo(v, self.denoms, self.prev[0].acts, self.acts, n, d['channels'], d['imgPixels'],
                                           d['size'], d['scale'], d['pow'], d['blocked'], target=tgt, sT=float(st), sO=1.0)
        else:
            K.response_norm_undo(v, self.denoms, self.prev[0].acts, self.acts, n, d['channels'], d['imgSizeY'],
                                 d['imgSizeX'], d['size'], d['scale'], d['pow'], target=tgt, sT=float(st), sO=1.0)


class CrossMapResponseNormLayer(ResponseNormLayer):
    cross = True


class ContrastNormLayer(ResponseNormLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:897-902
        d, n = self.d, self.net.num_cases
        x = self.inputs[0]
        mean = K.local_pool(x, n, d['channels'], d['imgSizeY'], d['imgSizeX'], d['size'], -(d['size'] // 2), 1,
                            d['imgSizeY'], d['imgSizeX'], 'avg')
        self.mean_diffs = (np.float32(-1) * mean + np.float32(1) * x).astype(np.float32)  # _meanDiffs.add(images, -1, 1)
        t, self.denoms = K.contrast_norm(x, self.mean_diffs, n, d['channels'], d['imgSizeY'], d['imgSizeX'], d['size'],
                                         d['scale'], d['pow'])
        self.acts[...] = t

    def bprop_acts(self, v, i, st, pt):  # layer.cu:904-906 (destroys own acts)
        d, n = self.d, self.net.num_cases
        K.response_norm_undo(v, self.denoms, self.mean_diffs, self.acts, n, d['channels'], d['imgSizeY'], d['imgSizeX'],
                             d['size'], d['scale'], d['pow'], target=self.prev[0].get_acts_grad(), sT=float(st), sO=1.0)


class CostLayer(Layer):
    def __init__(self, net, d):
        super().__init__(net, d)
        self.coeff = float(d['coeff'])
        self.costv = []

    def bprop(self, pt):  # layer.cu:929-933
        if self.coeff != 0:
            super().bprop(pt)

    def is_grad_producer(self):
        return self.coeff != 0


class LogregCostLayer(CostLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:969-981
        if i == 0:
            labels, probs = self.inputs
            n = self.net.num_cases
            lp, cp = K.logreg_cost(labels, probs, probs.shape[0], n)
            self.acts = lp
            self.costv = [-K.sum_all(lp), n - K.sum_all(cp)]

    def bprop_acts(self, v, i, st, pt):  # layer.cu:983-994
        assert i == 1
        p = self.prev[1]
        if len(p.next) > 1 or p.type != 'softmax':
            K.logreg_grad(self.prev[0].acts, p.acts, p.outputs, self.net.num_cases, self.coeff,
                          target=p.get_acts_grad(), add=st == 1)


class SumOfSquaresCostLayer(CostLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:1004-1008
        self.acts = self.inputs[0] * self.inputs[0]
        self.costv = [K.sum_all(self.acts)]

    def bprop_acts(self, v, i, st, pt):  # layer.cu:1010-1012
        K.axpby(self.prev[i].get_acts_grad(), self.inputs[0], float(st), -2 * self.coeff)


class L2SNCostLayer(CostLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:1060-1071
        if i == 0:
            labels, data = self.inputs
            self.m_eff = self.d['m']
            if self.net.comm is not None and self.m_eff < 0:
                # data-parallel: the data-dependent margin is searched over the pairs of ALL replicas
                # (rank-major order == the pair order of the single-GPU global minibatch)
                _, _, _, dists, plab = K.l2sn_cost(labels, data, data.shape[0], self.net.num_cases, 1.0)
                self.m_eff, _ = K.l2sn_optimal_thr(self.net.comm.all_gather(dists), self.net.comm.all_gather(plab))
            out, self.m_used, self.m_idx, self.dists, self.pair_labels = K.l2sn_cost(
                labels, data, data.shape[0], self.net.num_cases, self.m_eff)
            self.acts = out
            self.costv = [K.sum_all(out)]

    def bprop_acts(self, v, i, st, pt):  # layer.cu:1073-1082
        if i == 1:
            data = self.prev[1].acts
            K.l2sn_grad(self.prev[0].acts, data, data.shape[0], self.net.num_cases, self.coeff, self.m_eff,
                        target=self.prev[1].get_acts_grad(), add=st == 1)


class Fisher2CostLayer(CostLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:1025-1036
        if i == 0:
            if self.net.comm is not None:
                raise ValueError('cost.fisher2 couples every pair of the minibatch: not defined for data-parallel replicas')
            labels, data = self.inputs
            n = self.net.num_cases
            self.costv = [float(np.float32(K.fisher2_cost(labels, data, data.shape[0], n, self.d['alpha'])) * np.float32(n))]

    def bprop_acts(self, v, i, st, pt):  # layer.cu:1038-1047
        if i == 1:
            data = self.prev[1].acts
            K.fisher2_grad(self.prev[0].acts, data, data.shape[0], self.net.num_cases, self.d['alpha'], self.coeff,
                           target=self.prev[1].get_acts_grad(), add=st == 1)


class L2regCostLayer(CostLayer):
    def fprop_acts(self, i, st, pt):  # layer.cu:1093-1105
        if i == 0:
            ground, predict = self.inputs
            out = K.l2reg_cost(np.ascontiguousarray(ground), predict, predict.shape[0], self.net.num_cases)
            self.acts = out
            self.costv = [K.sum_all(out)]

    def bprop_acts(self, v, i, st, pt):  # layer.cu:1107-1116
        assert i == 1
        ground, predict = self.inputs
        K.l2reg_grad(np.ascontiguousarray(ground), predict, predict.shape[0], self.net.num_cases, self.coeff,
                     target=self.prev[1].get_acts_grad(), add=st == 1)


_TYPES = {'fc': FCLayer, 'conv': ConvLayer, 'local': LocalUnsharedLayer, 'pool': PoolLayer,
          'rnorm': ResponseNormLayer, 'cmrnorm': CrossMapResponseNormLayer, 'cnorm': ContrastNormLayer, 'softmax': SoftmaxLayer,
          'nailbed': NailbedLayer, 'blur': GaussianBlurLayer, 'resize': ResizeLayer, 'rgb2yuv': RGBToYUVLayer,
          'rgb2lab': RGBToLABLayer, 'shift': ShiftLayer,
          'eltsum': EltwiseSumLayer, 'eltmax': EltwiseMaxLayer, 'neuron': NeuronLayer, 'data': DataLayer,
          'cost.logreg': LogregCostLayer, 'cost.sum2': SumOfSquaresCostLayer, 'cost.l2-sn': L2SNCostLayer,
          'cost.l2-reg': L2regCostLayer, 'cost.fisher2': Fisher2CostLayer}


class OracleNet:
    """ConvNet (convnet.cu:44-81) + TrainingWorker (worker.cu:88-102) on the CPU oracle."""

    def __init__(self, layers, minibatch_size=128, comm=None):
        """comm: None, or a replica group with .world, .all_reduce(array) (in-place sum) and
        .all_gather(array) (rank-major concatenation) -- the data-parallel algorithm of the product
         restated on the CPU."""
        self.minibatch_size = minibatch_size
        self.comm = comm
        self.world = comm.world if comm is not None else 1
        self.layers, self.costs, self.data_layers = [], [], []
        for d in layers:
            l = _TYPES[d['type']](self, d)
            self.layers.append(l)
            if d['type'] == 'data':
                self.data_layers.append(l)
            if d['type'].startswith('cost.'):
                self.costs.append(l)
            for i in d.get('inputs', []):
                l.prev.append(self.layers[i])
        for l in self.layers:
            for p in l.prev:
                p.next.append(l)
                p.num_grad_producers_next += l.is_grad_producer()
        self.num_cases = 0

    def layer(self, name):
        return next(l for l in self.layers if l.name == name)

    def fprop(self, data, pt=PASS_TRAIN):
        self.num_cases = data[0].shape[1]
        self.data = data
        for l in self.layers:
            l.reset()
        for dl in self.data_layers:
            dl.fprop_data(data, pt)

    def bprop(self, pt=PASS_TRAIN):
        for c in self.costs:
            c.bprop(pt)
        for l in self.layers:
            l.reset()

    def update_weights(self):
        for l in self.layers:
            l.update_weights()

    def get_cost(self):
        return {c.name: list(c.costv) for c in self.costs}

    def cost_value(self):
        return sum(c.coeff * c.costv[0] for c in self.costs)

    def run_batch(self, data, test=False):
        """data: list of (dims, numCases) float32 arrays.  Returns ({name: [sums]}, numCases)."""
        n_total = data[0].shape[1]
        tot = {}
        for s in range(0, n_total, self.minibatch_size):
            mini = [np.ascontiguousarray(d[:, s:s + self.minibatch_size]) for d in data]
            self.fprop(mini, PASS_TEST if test else PASS_TRAIN)
            for k, v in self.get_cost().items():
                acc = tot.setdefault(k, [0.0] * len(v))
                for j, x in enumerate(v):
                    acc[j] += x
            if not test:
                self.bprop(PASS_TRAIN)
                self.update_weights()
        return tot, n_total

    def write_features(self, data, layer_idx):
        """FeatureWorker::run (worker.cu:191-214): (numCases, features) row-major."""
        n_total = data[0].shape[1]
        out = np.empty((n_total, self.layers[layer_idx].outputs), np.float32)
        for s in range(0, n_total, self.minibatch_size):
            mini = [np.ascontiguousarray(d[:, s:s + self.minibatch_size]) for d in data]
            self.fprop(mini, PASS_TEST)
            out[s:s + mini[0].shape[1]] = self.layers[layer_idx].acts.T
        return out

    def run_multiview(self, data, num_views, logreg_idx):
        """MultiviewTestWorker::run (worker.cu:142-174): the batch holds num_views views of numCases/num_views real cases,
        view-major; the softmax outputs of the views are averaged and the logreg cost is taken of the average.  The
        other cost layers report what their last view left (ConvNet::getCost adds every cost layer, convnet.cu:240-245)."""
        n_total = data[0].shape[1]
        assert n_total % num_views == 0
        real = n_total // num_views
        mb = self.minibatch_size
        logreg = self.layers[logreg_idx]
        tot, cases = {}, 0
        for i in range((real + mb - 1) // mb):
            sm = None
            for v in range(num_views):
                s, e = v * real + i * mb, min((v + 1) * real, v * real + (i + 1) * mb)
                self.fprop([np.ascontiguousarray(d[:, s:e]) for d in data], PASS_TEST)
                acts = logreg.prev[1].acts
                sm = acts.copy() if v == 0 else (sm + acts).astype(np.float32)
            sm = (sm * np.float32(1.0 / num_views)).astype(np.float32)
            logreg.fprop([logreg.prev[0].acts, sm], PASS_TEST)
            for k, vals in self.get_cost().items():
                acc = tot.setdefault(k, [0.0] * len(vals))
                for j, x in enumerate(vals):
                    acc[j] += x
            cases += self.num_cases
        return tot, cases

    def sync_with_host(self):
        for l in self.layers:
            l.copy_to_host()

    # -- the reference's only self-check (convnet.cu:257-330) ---------------
    def check_gradients(self, data, max_elems=40, seed=0, thresh=0.02, central=False):
        """Finite-difference check with the reference's steps and threshold.
        Unlike the reference, at most `max_elems` entries of each matrix are
        perturbed (seeded choice); norms are taken over that subset.  central=True
        replaces the reference's forward difference (err - base)/(N*eps) by
        (err(+eps) - err(-eps))/(2*N*eps), which removes the O(eps) curvature term
        that dominates at the FC step of 0.1.  Call once per net: like the
        reference (which exits after the check) conv gradients accumulate."""
        rng = np.random.default_rng(seed)
        self.fprop(data, PASS_GC)
        base = self.cost_value()
        self.bprop(PASS_GC)
        n = self.num_cases
        results = []
        for l in self.layers:
            if not isinstance(l, WeightLayer):
                continue
            mats = [('%s weights[%d]' % (l.name, i), w, l.w_step) for i, w in enumerate(l.weights)]
            mats.append(('%s biases' % l.name, l.biases, l.b_step))
            for name, w, eps in mats:
                flat = w.W.reshape(-1)
                anal = (w.get_grad().reshape(-1) * (-1.0 / n)).astype(np.float64)
                idx = rng.choice(flat.size, size=min(max_elems, flat.size), replace=False)
                num = np.zeros(len(idx))
                for j, e in enumerate(idx):
                    v = flat[e]
                    flat[e] = v + np.float32(eps)
                    self.fprop(data, PASS_GC)
                    up = self.cost_value()
                    if central:
                        flat[e] = v - np.float32(eps)
                        self.fprop(data, PASS_GC)
                        num[j] = (up - self.cost_value()) / (2 * n * eps)
                    else:
                        num[j] = (up - base) / (n * eps)
                    flat[e] = v
                an = anal[idx]
                rel = np.linalg.norm(num - an) / max(np.linalg.norm(an), 1e-30)
                results.append((name, float(rel), bool(rel < thresh)))
        return results
