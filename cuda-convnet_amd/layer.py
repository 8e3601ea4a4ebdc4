"""Net-definition (.cfg) parser: layer-def + layer-params files -> the `layers`
list-of-dicts that `_ConvNet.initModel` consumes.

Python-3 counterpart of the reference's layer.py (LayerParser.parse_layers
layer.py:247-288, LocalLayerParser.parse :855-925, ConvLayerParser :927-947,
LocalUnsharedLayerParser :949-962, FCLayerParser :777-791, PoolLayerParser
:986-1022, NormLayerParser :1024-1067, cost parsers :1069-1159, neuron parsers
:1236-1246).  Same cfg grammar, same dict keys, same
buffer-sharing optimisation (LayerWithInputParser.optimize :307-323) and the
same implicit-neuron detaching (:238-244, :566-582).

Additions over the reference (which only supports square images, layer.py:351-354):
every spatial layer also carries imgSizeY/imgSizeX, modulesY/modulesX,
outputsY/outputsX.  A data layer may declare `imgSizeX=` (and `channels=`) in
its cfg section; downstream layers inherit the geometry of their input.
"""
import configparser
import re
from collections import OrderedDict
from math import ceil

import numpy as np


class LayerParsingError(Exception):
    pass


# neuron type -> (param names, usesActs, usesInputs)       (layer.py:1236-1246)
NEURONS = OrderedDict([
    ('abs', ((), False, True)),
    ('brelu', (('a',), True, False)),
    ('ident', ((), False, False)),
    ('linear', (('a', 'b'), True, False)),
    ('logistic', ((), True, False)),
    ('relu', ((), True, False)),
    ('softrelu', ((), True, False)),
    ('sqrt', ((), True, False)),
    ('square', ((), False, True)),
    ('tanh', (('a', 'b'), True, False)),
])

UNSUPPORTED = ('rscale',)   # parsed by the reference (layer.py:460-490) but unknown to its own initModel (convnet.cu:86-134)


def parse_neuron(name, s):
    m = re.match(r'^\s*(\w+)\s*(?:\[([\d,\.\s\-e]*)\])?\s*$', s)
    if m and m.group(1) in NEURONS:
        pnames, uses_acts, uses_inputs = NEURONS[m.group(1)]
        vals = [float(v) for v in m.group(2).split(',')] if m.group(2) else []
        if len(vals) == len(pnames):
            params = dict(zip(pnames, vals))
            if m.group(1) == 'tanh':  # abs(tanh(bx)) symmetry, layer.py:84-90 keeps b positive for abstanh only
                pass
            return {'type': m.group(1), 'params': params, 'usesActs': uses_acts, 'usesInputs': uses_inputs}
    raise LayerParsingError("Layer '%s': unable to parse neuron type '%s'. Valid: %s" % (name, s, ', '.join(NEURONS)))


class Cfg:
    """Typed accessors over a configparser section (MyConfigParser layer.py:93-131)."""

    def __init__(self, path):
        self.cp = configparser.ConfigParser(dict_type=OrderedDict, interpolation=None)
        self.cp.optionxform = str
        with open(path) as f:
            self.cp.read_file(f)
        self.path = path

    def sections(self):
        return self.cp.sections()

    def has(self, sec, opt=None):
        return self.cp.has_section(sec) if opt is None else self.cp.has_option(sec, opt)

    def get(self, sec, opt, conv=str, typestr=None, default=None):
        if not self.cp.has_option(sec, opt):
            if default is not None:
                return default
            raise LayerParsingError("Layer '%s': required parameter '%s' missing" % (sec, opt))
        try:
            return conv(self.cp.get(sec, opt).strip())
        except ValueError:
            raise LayerParsingError("Layer '%s': parameter '%s' must be %s" % (sec, opt, typestr or conv.__name__))

    def get_int(self, sec, opt, default=None):
        return self.get(sec, opt, int, 'int', default)

    def get_float(self, sec, opt, default=None):
        return self.get(sec, opt, float, 'float', default)

    def get_bool(self, sec, opt, default=None):
        def conv(v):
            if v.lower() in ('1', 'true', 'yes', 'on'):
                return True
            if v.lower() in ('0', 'false', 'no', 'off'):
                return False
            raise ValueError(v)
        return self.get(sec, opt, conv, 'bool', default)

    def get_list(self, sec, opt, conv=str, typestr='strings', default=None):
        v = self.get(sec, opt, default=default)
        if isinstance(v, list):
            return v
        try:
            return [conv(x.strip()) for x in v.split(',')]
        except ValueError:
            raise LayerParsingError("Layer '%s': parameter '%s' must be ','-delimited list of %s" % (sec, opt, typestr))


def _check(cond, msg):
    if not cond:
        raise LayerParsingError(msg)


def _geom_of(layer, channels):
    """(Y, X) of a layer's output viewed as `channels` maps; square unless declared."""
    pixels = layer['outputs'] // channels
    g = layer.get('geom')
    if g and g[1] * g[2] * g[0] == layer['outputs'] and g[0] == channels:
        return g[1], g[2]
    if g and g[1] * g[2] == pixels:
        return g[1], g[2]
    s = int(np.sqrt(pixels))
    return s, s


class _Ctx:
    def __init__(self, data_dims, num_classes, rng):
        self.data_dims, self.num_classes, self.rng = data_dims, num_classes, rng


def _base(name, cfg):
    return {'name': name, 'type': cfg.get(name, 'type'), 'outputs': 0, 'requiresParams': False,
            'usesActs': True, 'usesInputs': True, 'forceOwnActs': True, 'gradConsumer': False}


def _with_inputs(name, cfg, prev, num_inputs=-1):
    dic = _base(name, cfg)
    names = [p['name'] for p in prev]
    ins = [s.strip() for s in cfg.get(name, 'inputs').split(',')]
    for i in ins:
        _check(i in names, "Layer '%s': input layer '%s' not defined" % (name, i))
    dic['inputs'] = [names.index(i) for i in ins]
    for i in dic['inputs']:
        _check(prev[i]['outputs'] != 0, "Layer '%s': input layer '%s' does not produce any output" % (name, names[i]))
    dic['numInputs'] = [prev[i]['outputs'] for i in dic['inputs']]
    dic['neuron'] = cfg.get(name, 'neuron', default='')
    _check(num_inputs <= 0 or len(ins) == num_inputs, "Layer '%s': number of inputs must be %d" % (name, num_inputs))
    return dic


def _parse_data(name, cfg, prev, ctx):
    dic = _base(name, cfg)
    dic['dataIdx'] = cfg.get_int(name, 'dataIdx')
    dic['outputs'] = int(ctx.data_dims[dic['dataIdx']])
    if cfg.has(name, 'imgSizeX'):
        x = cfg.get_int(name, 'imgSizeX')
        ch = cfg.get_int(name, 'channels', default=3)
        _check(dic['outputs'] % (ch * x) == 0, "Layer '%s': %d outputs not divisible by channels*imgSizeX" % (name, dic['outputs']))
        dic['geom'] = (ch, dic['outputs'] // (ch * x), x)
    return dic


def _parse_weight_common(dic, name, cfg, prev):
    dic['requiresParams'] = True
    dic['gradConsumer'] = True
    n_in = len(dic['inputs'])
    dic['initW'] = cfg.get_list(name, 'initW', float, 'floats', default=[0.01] * n_in)
    dic['initB'] = cfg.get_float(name, 'initB', default=0.0)
    dic['weightSource'] = cfg.get_list(name, 'weightSource', default=[''] * n_in)
    _check(len(dic['initW']) == n_in and len(dic['weightSource']) == n_in,
           "Layer '%s': initW/weightSource list length does not match number of inputs" % name)
    names = [p['name'] for p in prev]
    dic['weightSourceLayerIndices'], dic['weightSourceMatrixIndices'] = [], []
    for i, src in enumerate(dic['weightSource']):
        li = mi = -1
        if src != '':
            m = re.match(r'^\s*([^\s\[]+)(?:\[(\d+)\])?\s*$', src)
            _check(m is not None, "Layer '%s': unable to parse weight sharing source '%s'" % (name, src))
            sname, mi = m.group(1), int(m.group(2) or 0)
            _check(sname in names or sname == name, "Layer '%s': weight sharing source layer '%s' does not exist." % (name, sname))
            li = names.index(sname) if sname != name else len(names)
            srcl = prev[li] if sname != name else dic
            _check(srcl['type'] == dic['type'], "Layer '%s': weight sharing source layer '%s' is of type '%s'; should be '%s'." % (name, sname, srcl['type'], dic['type']))
            _check(sname != name or mi < i, "Layer '%s': weight sharing source '%s[%d]' not defined yet." % (name, name, mi))
        dic['weightSourceLayerIndices'].append(li)
        dic['weightSourceMatrixIndices'].append(mi)


def _make_weights(dic, prev, ctx, rows, cols, order):
    dic['weights'], dic['weightsInc'] = [], []
    for i in range(len(dic['inputs'])):
        li = dic['weightSourceLayerIndices'][i]
        if li >= 0:  # shared matrix (layer.py:706-713)
            src = prev[li] if li < len(prev) else dic
            w = src['weights'][dic['weightSourceMatrixIndices'][i]]
            _check(w.shape == (rows[i], cols[i]), "Layer '%s': weight sharing source matrix '%s' has wrong shape" % (dic['name'], dic['weightSource'][i]))
            dic['weights'].append(w)
            dic['weightsInc'].append(src['weightsInc'][dic['weightSourceMatrixIndices'][i]])
        else:
            w = np.array(dic['initW'][i] * ctx.rng.standard_normal((rows[i], cols[i])), dtype=np.float32, order=order)
            dic['weights'].append(w)
            dic['weightsInc'].append(np.zeros_like(w))


def _make_biases(dic, rows, cols, order):
    dic['biases'] = np.require(dic['initB'] * np.ones((rows, cols), dtype=np.float32), requirements=order)
    dic['biasesInc'] = np.zeros_like(dic['biases'])


def _parse_fc(name, cfg, prev, ctx):
    dic = _with_inputs(name, cfg, prev)
    _parse_weight_common(dic, name, cfg, prev)
    dic['usesActs'] = False
    dic['outputs'] = cfg.get_int(name, 'outputs')
    _check(dic['outputs'] >= 1, "Layer '%s': parameter 'outputs' must be greater than or equal to 1" % name)
    _make_weights(dic, prev, ctx, dic['numInputs'], [dic['outputs']] * len(dic['numInputs']), 'F')
    _make_biases(dic, 1, dic['outputs'], 'F')
    return dic


def _parse_local(name, cfg, prev, ctx, conv):
    dic = _with_inputs(name, cfg, prev)
    _parse_weight_common(dic, name, cfg, prev)
    n_in = len(dic['inputs'])
    dic['usesActs'] = False
    dic['channels'] = cfg.get_list(name, 'channels', int, 'ints')
    dic['padding'] = cfg.get_list(name, 'padding', int, 'ints', default=[0] * n_in)
    dic['stride'] = cfg.get_list(name, 'stride', int, 'ints', default=[1] * n_in)
    dic['filterSize'] = cfg.get_list(name, 'filterSize', int, 'ints')
    dic['filters'] = cfg.get_list(name, 'filters', int, 'ints')
    dic['groups'] = cfg.get_list(name, 'groups', int, 'ints', default=[1] * n_in)
    dic['randSparse'] = cfg.get_list(name, 'randSparse', lambda x: x.lower() in ('true', '1'), 'bools', default=[False] * n_in)
    dic['initW'] = cfg.get_list(name, 'initW', float, 'floats')
    for k in ('channels', 'padding', 'stride', 'filterSize', 'filters', 'groups', 'randSparse', 'initW'):
        _check(len(dic[k]) == n_in, "Layer '%s': %s list length does not match number of inputs" % (name, k))
    _check(min(dic['stride']) >= 1 and min(dic['filterSize']) >= 1 and min(dic['padding']) >= 0
           and min(dic['channels']) >= 1 and min(dic['groups']) >= 1, "Layer '%s': parameter out of range" % name)

    dic['imgPixels'] = [ni // c for ni, c in zip(dic['numInputs'], dic['channels'])]
    yx = [_geom_of(prev[i], c) for i, c in zip(dic['inputs'], dic['channels'])]
    dic['imgSizeY'], dic['imgSizeX'] = [g[0] for g in yx], [g[1] for g in yx]
    dic['imgSize'] = list(dic['imgSizeX'])
    dic['filters'] = [f * g for f, g in zip(dic['filters'], dic['groups'])]
    dic['filterPixels'] = [fs * fs for fs in dic['filterSize']]
    mod = lambda img: [1 + int(ceil((2 * p + s - fs) / float(st)))
                       for p, s, fs, st in zip(dic['padding'], img, dic['filterSize'], dic['stride'])]
    mx, my = mod(dic['imgSizeX']), mod(dic['imgSizeY'])
    dic['filterChannels'] = [c // g for c, g in zip(dic['channels'], dic['groups'])]
    if any(dic['randSparse']):  # then filterChannels must be given for every input (layer.py:888-890)
        dic['filterChannels'] = cfg.get_list(name, 'filterChannels', int, 'ints', default=dic['filterChannels'])
        _check(len(dic['filterChannels']) == n_in, "Layer '%s': filterChannels list length does not match number of inputs" % name)
    _check(len(set(mx)) == 1 and len(set(my)) == 1 and len(set(dic['filters'])) == 1,
           "Layer '%s': all inputs must produce equally-dimensioned output." % name)
    dic['modulesX'], dic['modulesY'] = mx[0], my[0]
    dic['modules'] = mx[0] * my[0]
    dic['filters'] = dic['filters'][0]
    dic['outputs'] = dic['modules'] * dic['filters']
    dic['geom'] = (dic['filters'], dic['modulesY'], dic['modulesX'])
    dic['filterConns'] = [[] for _ in range(n_in)]
    for i in range(n_in):
        _check(dic['numInputs'][i] % dic['imgPixels'][i] == 0 and dic['imgSizeY'][i] * dic['imgSizeX'][i] == dic['imgPixels'][i],
               "Layer '%s[%d]': has %d dimensional input, not interpretable as %d-channel images" % (name, i, dic['numInputs'][i], dic['channels'][i]))
        _check(dic['channels'][i] <= 3 or dic['channels'][i] % 4 == 0,
               "Layer '%s[%d]': number of channels must be smaller than 4 or divisible by 4" % (name, i))
        _check(dic['filterSize'][i] <= 2 * dic['padding'][i] + min(dic['imgSizeX'][i], dic['imgSizeY'][i]),
               "Layer '%s[%d]': filter size (%d) greater than image size + 2 * padding" % (name, i, dic['filterSize'][i]))
        if dic['randSparse'][i]:  # random sparse connectivity, layer.py:908-914
            ch, fch, grp = dic['channels'][i], dic['filterChannels'][i], dic['groups'][i]
            _check(grp > 1, "Layer '%s[%d]': number of groups must be greater than 1 when using random sparse connectivity" % (name, i))
            _check(fch >= 1 and ch % fch == 0, "Layer '%s[%d]': parameter 'channels' must be divisible by 'filterChannels'" % (name, i))
            _check(fch % 4 == 0, "Layer '%s[%d]': parameter 'filterChannels' must be divisible by 4" % (name, i))
            _check((grp * fch) % ch == 0, "Layer '%s[%d]': parameter 'groups * filterChannels' must be divisible by 'channels'" % (name, i))
            _check(dic['weightSourceLayerIndices'][i] < 0, "Layer '%s[%d]': weight sharing with random sparse connectivity is not supported" % (name, i))
            # gen_rand_conns, layer.py:830-833: overSample permutations of the channels (initCFunc is not supported)
            dic['filterConns'][i] = [int(x) for _ in range(grp * fch // ch) for x in ctx.rng.permutation(ch)]
        else:
            if dic['groups'][i] > 1:
                _check(dic['channels'][i] % (4 * dic['groups'][i]) == 0, "Layer '%s[%d]': parameter 'channels' must be divisible by '4 * groups'" % (name, i))
            _check(dic['channels'][i] % dic['groups'][i] == 0, "Layer '%s[%d]': parameter 'channels' must be divisible by 'groups'" % (name, i))
        _check(dic['filters'] % (16 * dic['groups'][i]) == 0, "Layer '%s[%d]': parameter 'filters * groups' must be divisible by %d" % (name, i, 16 * dic['groups'][i]))
        dic['padding'][i] = -dic['padding'][i]  # layer.py:922
    dic['overSample'] = [g * fc // c for g, fc, c in zip(dic['groups'], dic['filterChannels'], dic['channels'])]

    rows = [fp * fc for fp, fc in zip(dic['filterPixels'], dic['filterChannels'])]
    if conv:
        dic['partialSum'] = cfg.get_int(name, 'partialSum')
        dic['sharedBiases'] = cfg.get_bool(name, 'sharedBiases', default=True)
        _check(dic['partialSum'] == 0 or dic['modules'] % dic['partialSum'] == 0,
               "Layer '%s': convolutional layer produces %d outputs per filter, but given partialSum parameter (%d) does not divide this number" % (name, dic['modules'], dic['partialSum']))
        _make_weights(dic, prev, ctx, rows, [dic['filters']] * n_in, 'C')
        _make_biases(dic, dic['filters'] if dic['sharedBiases'] else dic['modules'] * dic['filters'], 1, 'C')
    else:
        _make_weights(dic, prev, ctx, [dic['modules'] * r for r in rows], [dic['filters']] * n_in, 'C')
        _make_biases(dic, dic['modules'] * dic['filters'], 1, 'C')
    return dic


def _parse_softmax(name, cfg, prev, ctx):
    dic = _with_inputs(name, cfg, prev, 1)
    dic['outputs'] = prev[dic['inputs'][0]]['outputs']
    return dic


def _parse_neuron_layer(name, cfg, prev, ctx, neuron_str=None):
    dic = _with_inputs(name, cfg, prev, 1)
    dic['outputs'] = dic['numInputs'][0]
    dic['neuron'] = parse_neuron(name, neuron_str if neuron_str is not None else cfg.get(name, 'neuron'))
    dic['usesActs'], dic['usesInputs'] = dic['neuron']['usesActs'], dic['neuron']['usesInputs']
    dic['forceOwnActs'] = False
    if prev[dic['inputs'][0]].get('geom'):
        dic['geom'] = prev[dic['inputs'][0]]['geom']
    return dic


def _parse_eltsum(name, cfg, prev, ctx):
    dic = _with_inputs(name, cfg, prev)
    _check(len(set(dic['numInputs'])) == 1, "Layer '%s': all inputs must have the same dimensionality." % name)
    dic['outputs'] = dic['numInputs'][0]
    dic['usesInputs'] = dic['usesActs'] = dic['forceOwnActs'] = False
    dic['coeffs'] = cfg.get_list(name, 'coeffs', float, 'floats', default=[1.0] * len(dic['inputs']))
    if prev[dic['inputs'][0]].get('geom'):
        dic['geom'] = prev[dic['inputs'][0]]['geom']
    return dic


def _parse_eltmax(name, cfg, prev, ctx):
    dic = _with_inputs(name, cfg, prev)
    _check(len(dic['inputs']) >= 2, "Layer '%s': elementwise max layer must have at least 2 inputs" % name)
    _check(len(set(dic['numInputs'])) == 1, "Layer '%s': all inputs must have the same dimensionality." % name)
    dic['outputs'] = dic['numInputs'][0]
    return dic


def _grad_consumers_below(dic, prev):
    if dic['gradConsumer']:
        return True
    return any(_grad_consumers_below(prev[i], prev) for i in dic.get('inputs', []))


def _parse_pool(name, cfg, prev, ctx):
    dic = _with_inputs(name, cfg, prev, 1)
    dic['channels'] = cfg.get_int(name, 'channels')
    dic['sizeX'] = cfg.get_int(name, 'sizeX')
    dic['start'] = cfg.get_int(name, 'start', default=0)
    dic['stride'] = cfg.get_int(name, 'stride')
    dic['outputsX'] = cfg.get_int(name, 'outputsX', default=0)
    dic['pool'] = cfg.get(name, 'pool')
    dic['usesActs'] = dic['usesInputs'] = True  # layer.py:1000-1001 compares the literal 'pool' != 'avg'
    dic['imgPixels'] = dic['numInputs'][0] // dic['channels']
    dic['imgSizeY'], dic['imgSizeX'] = _geom_of(prev[dic['inputs'][0]], dic['channels'])
    dic['imgSize'] = dic['imgSizeX']
    _check(1 <= dic['sizeX'] <= min(dic['imgSizeX'], dic['imgSizeY']), "Layer '%s': parameter 'sizeX' out of range" % name)
    _check(1 <= dic['stride'] <= dic['sizeX'], "Layer '%s': parameter 'stride' must be in the range 1-%d" % (name, dic['sizeX']))
    _check(dic['outputsX'] >= 0 and dic['channels'] >= 1, "Layer '%s': parameter out of range" % name)
    if _grad_consumers_below(dic, prev):
        _check(dic['channels'] % 16 == 0, "Layer '%s': parameter 'channels' must be divisible by 16" % name)
    _check(dic['pool'] in ('max', 'avg'), "Layer '%s': parameter 'pool' must be one of 'max', 'avg'" % name)
    _check(dic['numInputs'][0] % dic['imgPixels'] == 0 and dic['imgSizeY'] * dic['imgSizeX'] == dic['imgPixels'],
           "Layer '%s': has %d dimensional input, not interpretable as %d-channel images" % (name, dic['numInputs'][0], dic['channels']))
    auto = lambda s: int(ceil((s - dic['start'] - dic['sizeX']) / float(dic['stride']))) + 1
    if dic['outputsX'] <= 0:
        dic['outputsX'], dic['outputsY'] = auto(dic['imgSizeX']), auto(dic['imgSizeY'])
    else:
        dic['outputsY'] = dic['outputsX'] if dic['imgSizeY'] == dic['imgSizeX'] else auto(dic['imgSizeY'])
    dic['outputs'] = dic['outputsX'] * dic['outputsY'] * dic['channels']
    dic['geom'] = (dic['channels'], dic['outputsY'], dic['outputsX'])
    return dic


def _parse_norm(name, cfg, prev, ctx, cross):
    dic = _with_inputs(name, cfg, prev, 1)
    dic['requiresParams'] = True
    dic['channels'] = cfg.get_int(name, 'channels')
    dic['size'] = cfg.get_int(name, 'size')
    dic['blocked'] = cfg.get_bool(name, 'blocked', default=False)
    dic['imgPixels'] = dic['numInputs'][0] // dic['channels']
    dic['imgSizeY'], dic['imgSizeX'] = _geom_of(prev[dic['inputs'][0]], dic['channels'])
    dic['imgSize'] = dic['imgSizeX']
    dic['usesInputs'] = True
    _check(dic['channels'] >= 1, "Layer '%s': parameter 'channels' out of range" % name)
    if cross:
        _check(2 <= dic['size'] <= dic['channels'], "Layer '%s': parameter 'size' must be in the range 2-%d" % (name, dic['channels']))
        _check(dic['channels'] % 16 == 0, "Layer '%s': number of channels must be divisible by 16 when using crossMap" % name)
    else:
        _check(1 <= dic['size'] <= min(dic['imgSizeX'], dic['imgSizeY']), "Layer '%s': parameter 'size' out of range" % name)
        _check(dic['channels'] <= 3 or dic['channels'] % 4 == 0, "Layer '%s': number of channels must be smaller than 4 or divisible by 4" % name)
    _check(dic['imgSizeY'] * dic['imgSizeX'] == dic['imgPixels'], "Layer '%s': input not interpretable as %d-channel images" % (name, dic['channels']))
    dic['outputs'] = dic['imgPixels'] * dic['channels']
    dic['geom'] = (dic['channels'], dic['imgSizeY'], dic['imgSizeX'])
    return dic


def _no_grads_below(dic, prev):
    """LayerWithInputParser.verify_no_grads, layer.py:363-365."""
    _check(not _grad_consumers_below(dic, prev),
           "Layer '%s': layers of type '%s' cannot propagate gradient and must not be placed over layers with parameters."
           % (dic['name'], dic['type']))


def _image_op(name, cfg, prev, channels=None):
    """Common part of the image-op parsers (layer.py:367-530, :1161-1205): one input seen as `channels` maps."""
    dic = _with_inputs(name, cfg, prev, 1)
    dic['forceOwnActs'] = dic['usesActs'] = dic['usesInputs'] = False
    dic['channels'] = cfg.get_int(name, 'channels') if channels is None else channels
    _check(dic['channels'] >= 1, "Layer '%s': parameter 'channels' out of range" % name)
    _check(dic['numInputs'][0] % dic['channels'] == 0,
           "Layer '%s': has %d dimensional input, not interpretable as %d-channel images" % (name, dic['numInputs'][0], dic['channels']))
    dic['imgPixels'] = dic['numInputs'][0] // dic['channels']
    dic['imgSizeY'], dic['imgSizeX'] = _geom_of(prev[dic['inputs'][0]], dic['channels'])
    dic['imgSize'] = dic['imgSizeX']
    _check(dic['imgSizeY'] * dic['imgSizeX'] == dic['imgPixels'],
           "Layer '%s': has %d dimensional input, not interpretable as %d-channel images" % (name, dic['numInputs'][0], dic['channels']))
    return dic


def _square(dic):
    _check(dic['imgSizeY'] == dic['imgSizeX'], "Layer '%s': layers of type '%s' need square images" % (dic['name'], dic['type']))


def _parse_nailbed(name, cfg, prev, ctx):  # NailbedLayerParser layer.py:367-394
    dic = _image_op(name, cfg, prev)
    dic['stride'] = cfg.get_int(name, 'stride')
    _check(dic['stride'] >= 1, "Layer '%s': parameter 'stride' out of range" % name)
    out = lambda s: (s + dic['stride'] - 1) // dic['stride']
    dic['outputsX'], dic['outputsY'] = out(dic['imgSizeX']), out(dic['imgSizeY'])
    dic['start'] = min(dic['imgSizeX'] - dic['stride'] * (dic['outputsX'] - 1), dic['imgSizeY'] - dic['stride'] * (dic['outputsY'] - 1)) // 2
    dic['outputs'] = dic['channels'] * dic['outputsX'] * dic['outputsY']
    dic['geom'] = (dic['channels'], dic['outputsY'], dic['outputsX'])
    return dic


def _parse_blur(name, cfg, prev, ctx):  # GaussianBlurLayerParser layer.py:396-428
    dic = _image_op(name, cfg, prev)
    dic['outputs'] = dic['numInputs'][0]
    dic['filterSize'] = cfg.get_int(name, 'filterSize')
    dic['stdev'] = cfg.get_float(name, 'stdev')
    _check(dic['filterSize'] in (3, 5, 7, 9), "Layer '%s': parameter 'filterSize' must be one of 3, 5, 7, 9" % name)
    _check(dic['stdev'] > 0, "Layer '%s': parameter 'stdev' must be positive" % name)
    f = np.array([np.exp(-(dic['filterSize'] // 2 - i) ** 2 / float(2 * dic['stdev'] ** 2)) for i in range(dic['filterSize'])],
                 dtype=np.float32).reshape(1, dic['filterSize'])
    dic['filter'] = np.ascontiguousarray(f / f.sum(), dtype=np.float32)
    _check(dic['filterSize'] <= min(dic['imgSizeX'], dic['imgSizeY']),
           "Later '%s': filter size (%d) must be smaller than image size (%d)." % (name, dic['filterSize'], min(dic['imgSizeX'], dic['imgSizeY'])))
    dic['geom'] = (dic['channels'], dic['imgSizeY'], dic['imgSizeX'])
    return dic


def _parse_resize(name, cfg, prev, ctx):  # ResizeLayerParser layer.py:430-458
    dic = _image_op(name, cfg, prev)
    _square(dic)
    dic['scale'] = cfg.get_float(name, 'scale')
    _check(0.5 <= dic['scale'] <= 2, "Layer '%s': parameter 'scale' must be in the range 0.5-2" % name)
    dic['tgtSize'] = int(np.floor(dic['imgSize'] / dic['scale']))
    dic['tgtPixels'] = dic['tgtSize'] ** 2
    dic['outputs'] = dic['channels'] * dic['tgtPixels']
    dic['geom'] = (dic['channels'], dic['tgtSize'], dic['tgtSize'])
    dic['forceOwnActs'] = True   # every output reads four neighbours: never in the input's own matrix (scale == 1 would share it)
    _no_grads_below(dic, prev)
    return dic


def _parse_colour(name, cfg, prev, ctx, lab):  # ColorTransformLayerParser layer.py:492-530
    dic = _image_op(name, cfg, prev, channels=3)
    dic['outputs'] = dic['numInputs'][0]
    if lab:
        dic['center'] = cfg.get_bool(name, 'center', default=False)
    dic['geom'] = (3, dic['imgSizeY'], dic['imgSizeX'])
    _no_grads_below(dic, prev)
    return dic


def _parse_shift(name, cfg, prev, ctx):  # ShiftLayerParser layer.py:1161-1205
    dic = _image_op(name, cfg, prev)
    _square(dic)
    fs = dic['filterSize'] = cfg.get_int(name, 'filterSize')
    _check(2 <= fs <= dic['imgSize'], "Later '%s': filter size (%d) must be smaller than image size (%d)." % (name, fs, dic['imgSize']))
    out = dic['outputsX'] = dic['imgSize'] - fs + 1
    dic['outputs'] = out ** 2 * dic['channels'] * (fs - 1) * (fs - 1)
    filt = np.zeros((out * out, (fs - 1) * (fs - 1)), np.float32)
    for dx in range(fs - 1):
        for dy in range(fs - 1):
            for x in range(out):
                for y in range(out):
                    filt[x * out + y, dx * (fs - 1) + dy] = (x + dx) * dic['imgSize'] + (y + dy)
    dic['filter'] = filt
    dic['forceOwnActs'] = True   # a gather: never in the input's own matrix
    dic['geom'] = ((fs - 1) * (fs - 1) * dic['channels'], out, out)
    # the reference's ShiftLayer::bpropActs is empty (layer.cu:1135-1139): whatever is below would train on an unwritten
    # gradient matrix.  Refused here like the other layers that cannot propagate a gradient.
    _no_grads_below(dic, prev)
    return dic


def _parse_cost(name, cfg, prev, ctx, num_inputs):
    dic = _with_inputs(name, cfg, prev, num_inputs)
    dic['requiresParams'] = True
    del dic['neuron']
    t = dic['type']
    if t in ('cost.logreg', 'cost.l2-sn', 'cost.fisher2'):
        _check(dic['numInputs'][0] == 1, "Layer '%s': dimensionality of first input must be 1" % name)
    if t == 'cost.logreg':
        _check(prev[dic['inputs'][1]]['type'] == 'softmax', "Layer '%s': second input must be softmax layer" % name)
        if ctx.num_classes:
            _check(dic['numInputs'][1] == ctx.num_classes,
                   "Layer '%s': softmax input '%s' must produce %d outputs, because that is the number of classes in the dataset"
                   % (name, prev[dic['inputs'][1]]['name'], ctx.num_classes))
    if t == 'cost.l2-sn':
        dic['m'] = cfg.get_float(name, 'm')
    if t == 'cost.fisher2':  # Fisher2CostParser layer.py:1110-1122
        dic['alpha'] = cfg.get_float(name, 'alpha')
    return dic


PARSERS = {
    'data': _parse_data,
    'fc': _parse_fc,
    'conv': lambda n, c, p, x: _parse_local(n, c, p, x, True),
    'local': lambda n, c, p, x: _parse_local(n, c, p, x, False),
    'softmax': _parse_softmax,
    'eltsum': _parse_eltsum,
    'eltmax': _parse_eltmax,
    'neuron': _parse_neuron_layer,
    'pool': _parse_pool,
    'rnorm': lambda n, c, p, x: _parse_norm(n, c, p, x, False),
    'cmrnorm': lambda n, c, p, x: _parse_norm(n, c, p, x, True),
    'cnorm': lambda n, c, p, x: _parse_norm(n, c, p, x, False),
    'nailbed': _parse_nailbed,
    'blur': _parse_blur,
    'resize': _parse_resize,
    'rgb2yuv': lambda n, c, p, x: _parse_colour(n, c, p, x, False),
    'rgb2lab': lambda n, c, p, x: _parse_colour(n, c, p, x, True),
    'shift': _parse_shift,
    'cost.logreg': lambda n, c, p, x: _parse_cost(n, c, p, x, 2),
    'cost.sum2': lambda n, c, p, x: _parse_cost(n, c, p, x, 1),
    'cost.l2-sn': lambda n, c, p, x: _parse_cost(n, c, p, x, 2),
    'cost.l2-reg': lambda n, c, p, x: _parse_cost(n, c, p, x, 2),
    'cost.fisher2': lambda n, c, p, x: _parse_cost(n, c, p, x, 2),
}


class _FakeCfg:
    """Lets the implicit neuron layer reuse the ordinary parser (FakeConfigParser layer.py:134-139)."""

    def __init__(self, d):
        self.d = d

    def get(self, sec, opt, conv=str, typestr=None, default=None):
        return self.d.get(opt, default)

    def has(self, sec, opt=None):
        return opt in self.d


def _detach_neurons(layers, ctx):
    """layer.py:238-244 + :566-582: `neuron=x` on a layer becomes a separate
    '<name>_neuron' layer placed right after it; later layers are re-indexed."""
    out = []
    for idx, l in enumerate(layers):
        out.append(l)
        if l['type'] != 'neuron' and l.get('neuron'):
            wish = '%s_neuron' % l['name']
            names = set(x['name'] for x in layers)
            nm = wish
            k = 1
            while nm in names:
                nm = '%s.%d' % (wish, k)
                k += 1
            fake = _FakeCfg({'type': 'neuron', 'inputs': l['name'], 'neuron': l['neuron']})
            nl = _parse_neuron_layer(nm, fake, out, ctx, neuron_str=l['neuron'])
            for later in layers[idx + 1:]:
                if 'inputs' in later:
                    later['inputs'] = [i + (i >= len(out) - 1) for i in later['inputs']]
                if 'weightSourceLayerIndices' in later:
                    later['weightSourceLayerIndices'] = [i + (i >= len(out)) for i in later['weightSourceLayerIndices']]
            out.append(nl)
    return out


def _optimize(dic, layers):
    """Buffer sharing (LayerParser.optimize layer.py:148-150, LayerWithInputParser.optimize :307-323)."""
    dic['actsTarget'] = dic['actsGradTarget'] = -1
    if 'inputs' not in dic or dic['forceOwnActs']:
        return
    for i, inp in enumerate(dic['inputs']):
        l = layers[inp]
        n_users = sum(1 for ll in layers if 'inputs' in ll and inp in ll['inputs'])
        if l['outputs'] == dic['outputs'] and n_users == 1:
            if not l['usesActs'] and not dic['usesInputs']:
                dic['actsTarget'] = i
            dic['actsGradTarget'] = i


def _add_params(dic, pcfg):
    """Hyper-parameters from the layer-params file (WeightLayerParser.add_params
    layer.py:641-651, NormLayerParser.add_params :1033-1037, CostParser.add_params :1079-1081)."""
    name, t = dic['name'], dic['type']
    if t in ('fc', 'conv', 'local'):
        dic['epsW'] = pcfg.get_list(name, 'epsW', float, 'floats')
        dic['epsB'] = pcfg.get_float(name, 'epsB')
        dic['momW'] = pcfg.get_list(name, 'momW', float, 'floats')
        dic['momB'] = pcfg.get_float(name, 'momB')
        dic['wc'] = pcfg.get_list(name, 'wc', float, 'floats')
        for k in ('epsW', 'momW', 'wc'):
            _check(len(dic[k]) == len(dic['inputs']), "Layer '%s': %s list length does not match number of inputs" % (name, k))
        dic['gradConsumer'] = dic['epsB'] > 0 or any(w > 0 for w in dic['epsW'])
    elif t in ('rnorm', 'cmrnorm', 'cnorm'):
        dic['scale'] = pcfg.get_float(name, 'scale')
        dic['scale'] /= dic['size'] if t == 'cmrnorm' else dic['size'] ** 2
        dic['pow'] = pcfg.get_float(name, 'pow')
    elif t.startswith('cost.'):
        dic['coeff'] = pcfg.get_float(name, 'coeff')


def parse_layers(layer_cfg_path, param_cfg_path, data_dims, num_classes=0, seed=0, conserve_mem=0, layers=None):
    """Returns the `layers` list for `_ConvNet.initModel`.

    data_dims: {dataIdx: dimensionality} (what the data provider's
    get_data_dims(idx) returns in the reference, layer.py:971).
    layers: an existing list (e.g. from a checkpoint) whose hyper-parameters
    are only refreshed from the params file (layer.py:276-284).
    """
    ctx = _Ctx(data_dims, num_classes, np.random.default_rng(seed))
    if not layers:
        cfg = Cfg(layer_cfg_path)
        layers = []
        for name in cfg.sections():
            _check(cfg.has(name, 'type'), "Layer '%s': no type given" % name)
            t = cfg.get(name, 'type')
            _check(t not in UNSUPPORTED, "Layer '%s': layer type '%s' is outside the scope of this build" % (name, t))
            _check(t in PARSERS, "Layer '%s': Unknown layer type: '%s'" % (name, t))
            layers.append(PARSERS[t](name, cfg, layers, ctx))
        layers = _detach_neurons(layers, ctx)
        for l in layers:
            _optimize(l, layers)
        for i, l in enumerate(layers):
            if not l['type'].startswith('cost.'):
                used = any(i in l2['inputs'] for l2 in layers if 'inputs' in l2)
                _check(used, "Layer '%s' of type '%s' is unused" % (l['name'], l['type']))
    pcfg = Cfg(param_cfg_path)
    for l in layers:
        _check(pcfg.has(l['name']) or not l['requiresParams'],
               "Layer '%s' of type '%s' requires extra parameters, but none given in file '%s'." % (l['name'], l['type'], param_cfg_path))
        _add_params(l, pcfg)
        l['conserveMem'] = conserve_mem
    return layers


def layer_index(layers, name):
    return [l['name'] for l in layers].index(name)
