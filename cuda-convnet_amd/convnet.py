"""Python-3 driver over the `_ConvNet` extension.

Counterpart of the reference's convnet.py / gpumodel.py (ConvNet.__init__ / init_model_lib
convnet.py:40-60, IGPUModel.train gpumodel.py:128-163, start_batch / finish_batch
convnet.py:112-119 + gpumodel.py:168-190, print_costs convnet.py:127-135, save_state
gpumodel.py:254-281): same call sequence into the native module, same cost printing,
same checkpoint dict {"model_state": {...}, "op": {...}}.  Data comes from the synthetic
providers of data.py (the reference's datasets are not available).

The native module is mandatory: there is no CPU fallback.
"""
import argparse
import math
import os
import pickle
import sys
import time

import numpy as np

from . import CFG_DIR, layer as layer_mod
from .data import SyntheticDataProvider, shard_for_rank


class _CheckpointUnpickler(pickle.Unpickler):
    _ALLOWED = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'),
                ('numpy', 'dtype'), ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
                ('numpy._core.numeric', '_frombuffer'), ('numpy.core.numeric', '_frombuffer'), ('builtins', 'bytearray'),
                ('collections', 'OrderedDict')}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError('checkpoint refers to %s.%s: not a plain weights checkpoint' % (module, name))


def import_native(hooks=None):
    """The `_ConvNet` extension.  hooks=True (or MCN_USE_TEST_HOOKS=1 in the environment): the TEST build `_ConvNetHooks`
    instead -- same sources plus the shared-memory rehearsal communicator (MCN_COMM=shm) and failNextBackwardAt, which the
    product binaries do not contain."""
    if hooks is None:
        hooks = os.environ.get('MCN_USE_TEST_HOOKS') == '1'
    try:
        if hooks:
            from . import _ConvNetHooks as mod
        else:
            from . import _ConvNet as mod
    except ImportError as e:  # fail loudly: the HIP path is the product
        raise ImportError('cuda_convnet_amd.%s is not built (make -C cuda-convnet_amd/csrc, or '
                          '__graft_entry__.build()): %s' % ('_ConvNetHooks' if hooks else '_ConvNet', e))
    return mod


class ConvNet:
    def __init__(self, layer_def, layer_params, data_provider, test_data_provider=None, minibatch_size=128, gpu=0,
                 seed=0, conserve_mem=0, layers=None, testing_freq=50, save_path=None, rank=0, world=1, comm_id=None):
        self.libmodel = import_native()
        self.train_data_provider = data_provider
        self.test_data_provider = test_data_provider or data_provider
        self.minibatch_size = minibatch_size
        self.testing_freq, self.save_path = testing_freq, save_path
        self.rank, self.world = rank, world
        self.op = dict(layer_def=layer_def, layer_params=layer_params, minibatch_size=minibatch_size, gpu=gpu, seed=seed)
        dims = {i: data_provider.get_data_dims(i) for i in range(len(data_provider.dims))}
        # init_model_state (convnet.py:55-60): a checkpoint's layers only get their hyper-parameters refreshed
        self.layers = layer_mod.parse_layers(layer_def, layer_params, dims, data_provider.get_num_classes(), seed=seed,
                                             conserve_mem=conserve_mem, layers=layers)
        self.model_state = {'layers': self.layers, 'epoch': 1, 'batchnum': data_provider.batch_range[0],
                            'train_outputs': [], 'test_outputs': []}
        self.train_outputs, self.test_outputs = self.model_state['train_outputs'], self.model_state['test_outputs']
        self.epoch, self.batchnum = 1, data_provider.batch_range[0]
        self.libmodel.initModel(self.layers, minibatch_size, gpu)  # convnet.py:52-53
        if world > 1:
            self.libmodel.initDataParallel(rank, world, comm_id)

    # ---- one data batch (gpumodel.py:168-190, convnet.py:112-119) ---------------------------------
    def shard(self, data):
        return shard_for_rank(data, self.rank, self.world, self.minibatch_size)

    def start_batch(self, batch_data, train=True):
        data = batch_data[2]
        for d in data:
            if d.dtype != np.float32:
                raise TypeError('All matrices returned by data provider must consist of single-precision floats.')
        self._keep = self.shard(data)
        self.libmodel.startBatch(self._keep, 0 if train else 1)

    def finish_batch(self):
        return self.libmodel.finishBatch()

    def sync_with_host(self):
        self.libmodel.syncWithHost()

    def check_gradients(self, max_elems=0):
        data = self.train_data_provider.get_next_batch()[2]
        return self.libmodel.checkGradients(data, max_elems)

    def write_features(self, layer_name, data):
        """shownet.py:238-259: returns (numCases, numFeatures) float32."""
        idx = layer_mod.layer_index(self.layers, layer_name)
        ftrs = np.zeros((data[0].shape[1], self.layers[idx]['outputs']), dtype=np.float32)
        self.libmodel.startFeatureWriter(list(data) + [ftrs], idx)
        self.libmodel.finishBatch()
        return ftrs

    def get_test_error(self):
        data = self.test_data_provider.get_next_batch()
        self.start_batch(data, train=False)
        return self.finish_batch()

    # ---- printing (convnet.py:127-157) -----------------------------------------------------------
    @staticmethod
    def format_costs(cost_outputs):
        costs, num_cases = cost_outputs
        parts = []
        for name, vals in costs.items():
            vals = [v / num_cases for v in vals]
            parts.append('%s:  %s' % (name, ', '.join('%6f' % v for v in vals)))
            if any(math.isnan(v) or math.isinf(v) for v in vals):
                print(' '.join(parts) + ' ^ got nan or inf!')
                sys.exit(1)
        return ' '.join(parts)

    def print_test_results(self):
        print('\n======================Test output======================')
        print(self.format_costs(self.test_outputs[-1]))
        print('-------------------------------------------------------')
        for l in self.layers:
            if 'weights' in l:
                for i, (w, wi) in enumerate(zip(l['weights'], l['weightsInc'])):
                    print("Layer '%s' weights[%d]: %e [%e]" % (l['name'], i, np.mean(np.abs(w)), np.mean(np.abs(wi))))
                print("Layer '%s' biases: %e [%e]" % (l['name'], np.mean(np.abs(l['biases'])), np.mean(np.abs(l['biasesInc']))))

    # ---- training loop (gpumodel.py:128-163) ------------------------------------------------------
    def get_num_batches_done(self):
        r = self.train_data_provider.batch_range
        return len(r) * (self.epoch - 1) + self.batchnum - r[0] + 1

    def train(self, num_epochs, quiet=False, prefetch=True):
        """gpumodel.py:128-163, plus: the batch fetched while the current one computes is also UPLOADED meanwhile
        (prefetchBatch: copy stream, own thread), so the next start_batch finds it in HBM."""
        next_data = self.train_data_provider.get_next_batch()
        uploaded = False
        while next_data[0] <= num_epochs:
            data = next_data
            self.epoch, self.batchnum = data[0], data[1]
            t0 = time.time()
            if uploaded:
                self.libmodel.startBatch([], 0)
            else:
                self.start_batch(data)
            next_data = self.train_data_provider.get_next_batch()  # load the next batch while this one computes
            uploaded = prefetch and next_data[0] <= num_epochs
            if uploaded:
                self._keep_next = self.shard(next_data[2])
                self.libmodel.prefetchBatch(self._keep_next)
            out = self.finish_batch()
            self.train_outputs.append(out)
            if not quiet and self.rank == 0:
                print('%d.%d... %s (%.3f sec)' % (self.epoch, self.batchnum, self.format_costs(out), time.time() - t0))
                sys.stdout.flush()
            if self.get_num_batches_done() % self.testing_freq == 0:
                self.sync_with_host()
                self.test_outputs.append(self.get_test_error())
                if not quiet and self.rank == 0:
                    self.print_test_results()
                self.conditional_save()

    # ---- checkpoints (gpumodel.py:254-281) ---------------------------------------------------------
    def conditional_save(self):
        if self.save_path and self.rank == 0:
            self.save_state(os.path.join(self.save_path, '%d.%d' % (self.epoch, self.batchnum)))

    def save_state(self, path):
        self.model_state['epoch'], self.model_state['batchnum'] = self.epoch, self.batchnum
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path, 'wb') as f:
            pickle.dump({'model_state': self.model_state, 'op': self.op}, f, protocol=pickle.HIGHEST_PROTOCOL)

    @staticmethod
    def load_state(path):
        """Reads a checkpoint written by save_state.  A checkpoint is nested dicts / lists / tuples of numbers, strings
        and numpy arrays, so the unpickler admits exactly that: numpy's array and dtype reconstructors and nothing else
        (an unrestricted pickle.load would run whatever a file from an untrusted source asks for)."""
        with open(path, 'rb') as f:
            return _CheckpointUnpickler(f).load()

    def write_features_mat(self, layer_name, feature_path, num_batches=None):
        """shownet.py:238-259 (do_write_features): one MATLAB file `data_batch_<n>` = {'feats': (numCases, numFeatures)} per
        test batch, the next batch fetched while the current one computes.  Returns the paths written."""
        import scipy.io
        os.makedirs(feature_path, exist_ok=True)
        idx = layer_mod.layer_index(self.layers, layer_name)
        num_ftrs = self.layers[idx]['outputs']
        paths = []
        next_data = self.test_data_provider.get_next_batch()
        b1 = next_data[1]
        while True:
            batch, data = next_data[1], next_data[2]
            ftrs = np.zeros((data[0].shape[1], num_ftrs), dtype=np.float32)
            self.libmodel.startFeatureWriter(list(data) + [ftrs], idx)
            next_data = self.test_data_provider.get_next_batch()  # load the next batch while the current one is computing
            self.libmodel.finishBatch()
            path_out = os.path.join(feature_path, 'data_batch_%d.mat' % batch)
            scipy.io.savemat(path_out, {'feats': ftrs})
            paths.append(path_out)
            if next_data[1] == b1 or (num_batches is not None and len(paths) >= num_batches):
                break
        return paths

    def close(self):
        self.libmodel.destroyModel()


PRESETS = {  # name: (layer def, layer params, provider kind, image geometry)
    'deepid': ('deepid_4213.cfg', 'deepid_params.cfg', 'deepid', (55, 55, 3)),
    'deepid-39x31': ('deepid_4213_39x31.cfg', 'deepid_params.cfg', 'deepid', (39, 31, 3)),
    'deepid2': ('deepid2_10575.cfg', 'deepid_params.cfg', 'deepid2', (55, 55, 3)),
    'deepidn': ('deepidn_10575.cfg', 'deepidn_params.cfg', 'deepid2', (55, 55, 3)),  # DeepID2 + cost.fisher2 (deepidn/*.cfg)
    'deepid2plus': ('deepid2plus_10575.cfg', 'deepid_params.cfg', 'deepid2', (55, 55, 3)),
    'deepid-attr': ('deepid_attr_4131.cfg', 'deepid_attr_params.cfg', 'deepid_attr', (55, 55, 3)),
    'cifar10': ('cifar10.cfg', 'cifar10_params.cfg', 'cifar', (32, 32, 3)),
    'cifar10-cnorm': ('cifar10_cnorm.cfg', 'cifar10_params.cfg', 'cifar', (32, 32, 3)),
    'cifar10-imgops': ('cifar10_imgops.cfg', 'cifar10_imgops_params.cfg', 'cifar', (32, 32, 3)),  # colour / resize / blur / nailbed / shift / randSparse
}


def make_preset(name, num_cases=1280, num_batches=1, minibatch=128, **kw):
    ldef, lpar, kind, img = PRESETS[name]
    dp = SyntheticDataProvider(kind, num_cases, num_batches=num_batches, minibatch=minibatch * kw.get('world', 1), img=img)
    return ConvNet(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dp, minibatch_size=minibatch, **kw)


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train a cuda-convnet style net on synthetic data (MI355X)')
    ap.add_argument('--preset', choices=sorted(PRESETS), default='deepid2')
    ap.add_argument('--layer-def')
    ap.add_argument('--layer-params')
    ap.add_argument('--mini', type=int, default=128)
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--batches', type=int, default=4, help='synthetic data batches per epoch')
    ap.add_argument('--cases', type=int, default=8192, help='cases per data batch')
    ap.add_argument('--test-freq', type=int, default=2)
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--save-path')
    ap.add_argument('--check-grads', action='store_true')
    a = ap.parse_args(argv)
    ldef, lpar, kind, img = PRESETS[a.preset]
    dp = SyntheticDataProvider('dummy' if a.check_grads and kind == 'cifar' else kind, 512 if a.check_grads else a.cases,
                               num_batches=a.batches, minibatch=a.mini, img=img)
    net = ConvNet(a.layer_def or os.path.join(CFG_DIR, ldef), a.layer_params or os.path.join(CFG_DIR, lpar), dp,
                  minibatch_size=a.mini, gpu=a.gpu, testing_freq=a.test_freq, save_path=a.save_path)
    if a.check_grads:  # convnet.py:114-115
        res = net.check_gradients(max_elems=25)
        fails = 0
        for name, an, nn, rel, fail in res:
            print('(%s) %s GRADIENT CHECK  analytic norm %e  numeric norm %e  relative error %e' % ('****FAIL****' if fail else 'PASS', name, an, nn, rel))
            fails += fail
        print('%d/%d TESTS FAILED' % (fails, len(res)) if fails else 'ALL %d TESTS PASSED' % len(res))
        net.close()
        return 1 if fails else 0
    net.train(a.epochs)
    net.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
