"""Directional-derivative gradient check that uses nothing but the drop-in boundary
(initModel / startBatch / finishBatch / syncWithHost semantics), so the same code checks the
CPU oracle and the HIP path.

For every weight matrix M: the analytic gradient g = -dE/dM comes out of ONE training step with
eps = 1, mom = 0, wc = 0 (then weightsInc = g / N, layer.cu:378-381, weights.cuh:150-163); the
numeric value is the central difference of the total cost E = sum_c coeff_c * cost_c along
D = sign(g) * u, u ~ U(0.5, 1.5):   (E(M + tD) - E(M - tD)) / 2t   vs   -sum(D * g).
All elements move at once, so the signal is the L1 mass of the gradient and fp32 rounding of the
cost (which limits the reference's per-element check, convnet.cu:280-330) is negligible."""
import copy

import numpy as np


def _with_hyper(layers, eps, mom, wc):
    L = copy.deepcopy(layers)
    for l in L:
        if 'weights' in l:
            l['epsW'] = [eps] * len(l['weights'])
            l['momW'] = [mom] * len(l['weights'])
            l['wc'] = [wc] * len(l['weights'])
            l['epsB'], l['momB'] = eps, mom
    return L


def total_cost(layers, costs):
    return sum(l['coeff'] * costs[l['name']][0] for l in layers if l['type'].startswith('cost.'))


def directional_check(layers, data, train_step, test_cost, t=2e-3, seed=0):
    """train_step(layers, data): one fprop+bprop+update, stepped weights/incs written back into `layers`.
    test_cost(layers, data) -> {cost name: [values]} of a test pass.  Returns [(matrix name, numeric, analytic, rel err)]."""
    n = data[0].shape[1]
    Lg = _with_hyper(layers, 1.0, 0.0, 0.0)
    train_step(Lg, data)
    rng = np.random.default_rng(seed)
    out = []
    for li, l in enumerate(layers):
        if 'weights' not in l:
            continue
        mats = [('weights', i) for i in range(len(l['weights']))] + [('biases', None)]
        for key, i in mats:
            inc = Lg[li]['weightsInc'][i] if i is not None else Lg[li]['biasesInc']
            g = inc.astype(np.float64) * n  # = -dE/dM
            D = (np.sign(g) * rng.uniform(0.5, 1.5, g.shape)).astype(np.float32)
            vals = []
            for sgn in (+1, -1):
                Lp = copy.deepcopy(layers)
                tgt = Lp[li]['weights'][i] if i is not None else Lp[li]['biases']
                tgt += np.float32(sgn * t) * D
                vals.append(total_cost(Lp, test_cost(Lp, data)))
            num = (vals[0] - vals[1]) / (2 * t)
            anal = -float(np.sum(D.astype(np.float64) * g))
            name = '%s %s' % (l['name'], key if i is None else '%s[%d]' % (key, i))
            out.append((name, num, anal, abs(num - anal) / max(abs(anal), 1e-30)))
    return out
