"""The data-parallel algorithm with more than one replica, on CPU: two processes over torch.distributed
(gloo), each running the whole-net oracle on its pair-preserving shard with gradients summed by
all-reduce and the L2-SN margin searched over the all-gathered pairs, must reproduce ONE process
stepping the concatenated minibatch of 128 x 2 (the equivalence statement).
The product's C++ layer graph implements the same algorithm over RCCL (net.cpp: gradReady /
flushBuckets / waitComm / L2SNCostLayer); its exchange path runs on the GPU in
tests/test_net_gpu.py::test_data_parallel_path_single_rank."""
import copy
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


class GlooComm:
    def __init__(self, dist, torch):
        self.dist, self.torch, self.world = dist, torch, dist.get_world_size()

    def all_reduce(self, a):
        t = self.torch.from_numpy(a)
        self.dist.all_reduce(t)  # sums in place; `a` shares the memory

    def all_gather(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs])


def _setup(cases, mb):
    sys.path.insert(0, ROOT)
    import ccn_loader
    ccn_loader.load()
    from cuda_convnet_amd import CFG_DIR, layer
    from cuda_convnet_amd.convnet import PRESETS
    from cuda_convnet_amd.data import SyntheticDataProvider
    ldef, lpar, kind, img = PRESETS['deepid2']
    dp = SyntheticDataProvider(kind, cases, img=img, minibatch=mb)  # L2-SN pairs are (j, j + mb/2) of the GLOBAL minibatch
    dims = {i: dp.get_data_dims(i) for i in range(len(dp.dims))}
    L = layer.parse_layers(os.path.join(CFG_DIR, ldef), os.path.join(CFG_DIR, lpar), dims, dp.get_num_classes(), seed=3)
    return L, dp.get_next_batch()[2]


def _worker(rank, port, q):
    os.environ['OMP_NUM_THREADS'] = '3'
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=WORLD)
    L, data = _setup(2 * 128 * WORLD, 128 * WORLD)
    from cuda_convnet_amd.data import shard_for_rank
    from oracle.net import OracleNet
    net = OracleNet(L, 128, comm=GlooComm(dist, torch))
    costs, n = net.run_batch(shard_for_rank(data, rank, WORLD, 128))
    net.sync_with_host()
    margins = net.layer('L2-SN').m_used
    q.put((rank, costs, n, margins, {l['name']: [np.array(w) for w in l['weights']] + [np.array(l['biases'])]
                                     for l in L if 'weights' in l}))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_replicas_equal_one_big_minibatch():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = []
    import queue
    import time
    deadline = time.time() + 300
    while len(res) < WORLD:
        try:
            res.append(q.get(timeout=1))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), 'a replica process died'
            assert time.time() < deadline, 'replicas timed out'
    res.sort(key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the single-process reference: same data, minibatch 256
    L, data = _setup(2 * 128 * WORLD, 128 * WORLD)
    L0 = copy.deepcopy(L)
    from oracle.net import OracleNet
    net = OracleNet(L, 128 * WORLD)
    want, n = net.run_batch(data)
    net.sync_with_host()
    assert sum(r[2] for r in res) == n
    assert res[0][3] == res[1][3]  # the data-dependent margin is bit-identical on every replica ...
    assert abs(res[0][3] - net.layer('L2-SN').m_used) <= 1e-5 * net.layer('L2-SN').m_used  # ... and the single-process one up to step-1 rounding
    for k, v in want.items():
        got = [sum(r[1][k][j] for r in res) for j in range(len(v))]
        assert abs(got[0] - v[0]) <= 1e-5 * abs(v[0]), (k, got, v)
        assert got[1:] == v[1:]
    for l, l0 in zip(L, L0):
        if 'weights' not in l:
            continue
        mats = [np.array(w) for w in l['weights']] + [np.array(l['biases'])]
        init = [np.array(w) for w in l0['weights']] + [np.array(l0['biases'])]
        for i, (w, w0) in enumerate(zip(mats, init)):
            step = float(np.max(np.abs(w - w0)))
            assert step > 0
            for r in res:
                assert np.array_equal(r[4][l['name']][i], res[0][4][l['name']][i]), 'replicas diverged'
                err = float(np.max(np.abs(r[4][l['name']][i] - w)))
                assert err <= 1e-3 * step + 1e-9, (l['name'], i, err, step)
