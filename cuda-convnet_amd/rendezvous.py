"""Single-node process-group plumbing for one-process-per-GPU jobs: broadcast of the RCCL unique id,
barrier and max-reduction of a scalar, over loopback TCP.

Reads the launcher's environment (RANK, WORLD_SIZE, MASTER_PORT as set by `python -m torch.distributed.run`)
but deliberately does not import torch in the GPU processes: torch ships its own copies of the HIP runtime
and RCCL, and mixing them with /opt/rocm's in one process makes ncclCommInitRank fail.  The data path
(gradient all-reduce, L2-SN all-gather) is RCCL inside the native module; this is only the control plane.

Rank 0 listens on an ephemeral loopback port and publishes it in a file named after the launcher's pid
and MASTER_PORT (all ranks of one job share both); the other ranks poll for it and connect.
"""
import os
import socket
import struct
import tempfile
import time


class Group:
    MAX_MSG = 1 << 20  # control-plane messages are a unique id (128 bytes) or a scalar

    def __init__(self, rank=None, world=None, token=None, timeout=120.0):
        self.rank = int(os.environ.get('RANK', 0)) if rank is None else rank
        self.world = int(os.environ.get('WORLD_SIZE', 1)) if world is None else world
        self.peers = []
        self.sock = None
        self._file = None
        if self.world == 1:
            return
        token = token or '%d_%s' % (os.getppid(), os.environ.get('MASTER_PORT', '0'))
        # a directory only this user can enter (another local user can neither pre-create the file nor read the port)
        d = os.path.join(tempfile.gettempdir(), 'mcn_rdzv_%d' % os.getuid())
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.stat(d)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError('%s is not a private directory of this user' % d)
        path = os.path.join(d, '%s.port' % token)
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.bind(('127.0.0.1', 0))
            srv.listen(self.world)
            tmp = path + '.%d' % os.getpid()
            with os.fdopen(os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600), 'w') as f:
                f.write('%d' % srv.getsockname()[1])
            os.replace(tmp, path)
            self._file = path
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack('<i', self._recv(c, 4))[0]
                conns[r] = c
            srv.close()
            self.peers = [conns[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    with open(path) as f:
                        port = int(f.read())
                    s = socket.create_connection(('127.0.0.1', port), timeout=5)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    s.sendall(struct.pack('<i', self.rank))
                    self.sock = s
                    break
                except (OSError, ValueError):  # file not there yet, or a stale one from a dead job
                    if time.time() > deadline:
                        raise TimeoutError('rank %d: no rendezvous with rank 0 (%s)' % (self.rank, path))
                    time.sleep(0.02)

    @staticmethod
    def _recv(s, n):
        buf = b''
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise ConnectionError('peer closed the rendezvous connection')
            buf += chunk
        return buf

    def _send_msg(self, s, payload):
        s.sendall(struct.pack('<q', len(payload)) + payload)

    def _recv_msg(self, s):
        n = struct.unpack('<q', self._recv(s, 8))[0]
        if not 0 <= n <= self.MAX_MSG:
            raise ConnectionError('rendezvous message of %d bytes refused' % n)
        return self._recv(s, n)

    def broadcast(self, payload=None):
        """bytes from rank 0 to everyone."""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for p in self.peers:
                self._send_msg(p, payload)
            return payload
        return self._recv_msg(self.sock)

    def gather(self, payload):
        """rank 0 gets [bytes of rank 0, 1, ...]; the others get None."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [self._recv_msg(p) for p in self.peers]
        self._send_msg(self.sock, payload)
        return None

    def barrier(self):
        self.broadcast(b'1' if self.gather(b'0') is not None else None)

    def max(self, value):
        vals = self.gather(struct.pack('<d', value))
        out = struct.pack('<d', max(struct.unpack('<d', v)[0] for v in vals)) if vals is not None else None
        return struct.unpack('<d', self.broadcast(out))[0]

    def close(self):
        for s in self.peers + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        if self._file:
            try:
                os.remove(self._file)
            except OSError:
                pass
