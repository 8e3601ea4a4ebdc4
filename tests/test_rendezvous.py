"""The control plane of a one-process-per-GPU job (cuda-convnet_amd/rendezvous.py): two real processes over loopback --
broadcast of a unique-id-sized payload, barrier, max -- and the refusals that protect it on a shared machine."""
import multiprocessing as mp
import os
import stat
import struct
import tempfile

import pytest

import ccn_loader

ccn_loader.load()
from cuda_convnet_amd import rendezvous  # noqa: E402


def _rank(rank, token, q):
    g = rendezvous.Group(rank=rank, world=2, token=token, timeout=30)
    uid = g.broadcast(bytes(range(128)) if rank == 0 else None)
    g.barrier()
    m = g.max(1.5 if rank == 0 else 2.5)
    g.close()
    q.put((rank, uid, m))


def test_two_ranks_broadcast_barrier_max():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    token = 'test_%d' % os.getpid()
    ps = [ctx.Process(target=_rank, args=(r, token, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1]
    assert all(g[1] == bytes(range(128)) and g[2] == 2.5 for g in got)
    d = os.path.join(tempfile.gettempdir(), 'mcn_rdzv_%d' % os.getuid())
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700
    assert not os.path.exists(os.path.join(d, token + '.port'))  # rank 0 removed it on close


def test_oversized_message_is_refused():
    import socket
    a, b = socket.socketpair()
    g = rendezvous.Group(rank=0, world=1)
    a.sendall(struct.pack('<q', 1 << 40))
    with pytest.raises(ConnectionError):
        g._recv_msg(b)
    a.close()
    b.close()
