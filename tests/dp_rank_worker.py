"""One rank of a data-parallel rehearsal on a shared GPU (run by tests/test_dp_two_ranks_gpu.py as a fresh child process,
before anything in it has touched the GPU): builds the preset, joins the replica group through the loopback control plane
and the shared-memory rehearsal communicator (MCN_COMM=shm, see mcn_comm_unique_id), trains `steps` minibatches of its
pair-preserving shard with the PRODUCT's exchange path (net.cpp flushBuckets / gatherPairsHere), and leaves weights,
costs and the log of collectives in an .npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world, name, steps, token, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
    import ccn_loader
    ccn_loader.load()
    from cuda_convnet_amd.convnet import import_native
    from cuda_convnet_amd.data import shard_for_rank
    from cuda_convnet_amd.rendezvous import Group
    from test_net_gpu import preset

    mb = 128
    dp, L = preset(name, mb * world * steps, minibatch=mb * world)   # the same seeds as the single-GPU run at minibatch 128*world
    data = shard_for_rank(dp.get_next_batch()[2], rank, world, mb)
    group = Group(rank, world, token=token)
    lib = import_native(hooks=True)  # the test build: the product binaries do not contain the rehearsal communicator
    lib.initModel(L, mb, 0)
    try:
        uid = group.broadcast(lib.commUniqueId() if rank == 0 else None)
        lib.initDataParallel(rank, world, uid)
        assert lib.commInfo() == (world, 0, 0), lib.commInfo()  # the rehearsal transport counts every rank, and says it is not RCCL
        group.barrier()
        lib.startBatch(data, 0)
        costs, num_cases = lib.finishBatch()
        log = lib.getCommLog()
        lib.syncWithHost()
        group.barrier()
    finally:
        lib.destroyModel()
    arrays = {}
    for l in L:
        if 'weights' in l:
            for i, (w, inc) in enumerate(zip(l['weights'], l['weightsInc'])):
                arrays['%s.w%d' % (l['name'], i)] = w
                arrays['%s.inc%d' % (l['name'], i)] = inc
            arrays['%s.b' % l['name']] = l['biases']
            arrays['%s.binc' % l['name']] = l['biasesInc']
    for k, v in costs.items():
        arrays['cost.%s' % k] = np.asarray(v, np.float64)
    arrays['numCases'] = np.asarray([num_cases])
    arrays['commLog'] = np.asarray([(0 if k == 'allreduce' else 1, o, c) for k, o, c in log], np.int64).reshape(-1, 3)
    np.savez(out, **arrays)


if __name__ == '__main__':
    main()
