"""Synthetic data providers.

The reference's providers read face datasets from .mat / HDF5 files that are not
available (convdata.py:754-1117); what the compute core
needs from a provider is only its contract (data.py:32-160 in the reference):

    get_next_batch() -> (epoch, batchnum, [float32 (dims, numCases) C-contiguous arrays])
    get_data_dims(idx), get_num_classes(), batch_range, curr_epoch / curr_batchnum

Every tensor is generated from numpy default_rng seeds as specified in
SURVEY.md section 8(d), so that the HIP path, the CPU oracle and every rank of a
data-parallel job see identical inputs.
"""
import numpy as np


def face_images(seed, dims, num_cases):
    """Mean-subtracted 8-bit pixels (cf. convdata.py:869): N(0,1)*64 clipped to [-128,127]."""
    rng = np.random.default_rng(seed)
    return np.clip(rng.standard_normal((dims, num_cases)) * 64, -128, 127).astype(np.float32)


def random_labels(seed, num_classes, num_cases):
    rng = np.random.default_rng(seed)
    return rng.integers(0, num_classes, size=(1, num_cases)).astype(np.float32)


def paired_labels(seed, num_classes, num_cases, minibatch, p_same=0.5):
    """Identity labels for the Joint identification-verification cost: inside every minibatch
    the pair (j, j + mini/2) shares an identity with probability p_same (layer_kernels.cu:517-529)."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, num_classes, size=num_cases)
    for s in range(0, num_cases, minibatch):
        n = min(minibatch, num_cases - s)
        h = n // 2
        same = rng.random(h) < p_same
        blk = ids[s:s + n]
        blk[h:2 * h][same] = blk[:h][same]
    return ids.astype(np.float32).reshape(1, num_cases)


class SyntheticDataProvider:
    """Serves `num_batches` fixed synthetic batches round-robin (LabeledDummyDataProvider
    data.py:162-184 in the reference, with the shapes of the BASELINE configs)."""

    def __init__(self, kind, num_cases, num_batches=1, minibatch=128, img=(55, 55, 3), num_classes=None, seed=1):
        self.kind, self.num_cases, self.minibatch = kind, num_cases, minibatch
        self.batch_range = list(range(1, num_batches + 1))
        self.curr_epoch, self.curr_batchnum = 1, self.batch_range[0]
        self.seed = seed
        y, x, c = img
        if kind == 'deepid':
            self.num_classes = num_classes or 4213
            self.dims = [y * x * c, 1]
        elif kind == 'deepid2':
            self.num_classes = num_classes or 10575
            self.dims = [y * x * c, 1]
        elif kind == 'deepid_attr':
            self.num_classes = num_classes or 4131
            # the reference's provider declares the three attribute inputs as 1-dimensional although the
            # arrays have 2 / 27 / 23 rows (convdata.py:926-927); get_data_dims keeps that quirk
            self.dims = [y * x * c, 1, 2, 27, 23]
        elif kind == 'cifar':
            self.num_classes = num_classes or 10
            self.dims = [32 * 32 * 3, 1]
        elif kind == 'dummy':
            self.num_classes = num_classes or 10
            self.dims = [y * x * c, 1]
        else:
            raise ValueError('unknown synthetic data kind %r' % kind)
        self._cache = {}

    def get_data_dims(self, idx=0):
        if self.kind == 'deepid_attr' and idx >= 2:
            return 1
        return self.dims[idx]

    def get_num_classes(self):
        return self.num_classes

    def make_batch(self, batchnum):
        if batchnum in self._cache:
            return self._cache[batchnum]
        s, n = self.seed + 100 * batchnum, self.num_cases
        if self.kind == 'cifar':
            rng = np.random.default_rng(s + 6)
            data = [(rng.standard_normal((self.dims[0], n)) * 50).astype(np.float32), random_labels(s + 7, self.num_classes, n)]
        elif self.kind == 'dummy':
            rng = np.random.default_rng(s)
            data = [rng.random((self.dims[0], n)).astype(np.float32), random_labels(s + 1, self.num_classes, n)]
        elif self.kind == 'deepid':
            data = [face_images(s, self.dims[0], n), random_labels(s + 1, self.num_classes, n)]
        else:
            data = [face_images(s, self.dims[0], n), paired_labels(s + 3, self.num_classes, n, self.minibatch)]
            if self.kind == 'deepid_attr':
                rng = np.random.default_rng(s + 5)
                data += [rng.standard_normal((d, n)).astype(np.float32) for d in self.dims[2:]]
        self._cache[batchnum] = data
        return data

    def get_next_batch(self):
        epoch, batchnum = self.curr_epoch, self.curr_batchnum
        data = self.make_batch(batchnum)
        i = self.batch_range.index(batchnum) + 1
        if i == len(self.batch_range):
            i = 0
            self.curr_epoch += 1
        self.curr_batchnum = self.batch_range[i]
        return epoch, batchnum, data


def shard_for_rank(data, rank, world, minibatch):
    """Pair-preserving data-parallel shard.

    `data` holds global minibatches of `minibatch * world` cases.  Rank r takes, from
    every global minibatch, gallery columns [r*h, (r+1)*h) and their partners
    [H + r*h, H + (r+1)*h), h = minibatch/2, H = world*h, so that each L2-SN pair
    (j, j + N/2) lives on one GPU.  The result has `minibatch` cases per local minibatch.
    """
    if world == 1:
        return data
    g = minibatch * world
    h, H = minibatch // 2, (minibatch // 2) * world
    total = data[0].shape[1]
    assert total % g == 0, 'global batch must be a multiple of minibatch*world'
    cols = []
    for s in range(0, total, g):
        cols.extend(range(s + rank * h, s + (rank + 1) * h))
        cols.extend(range(s + H + rank * h, s + H + (rank + 1) * h))
    cols = np.asarray(cols)
    return [np.ascontiguousarray(d[:, cols]) for d in data]
