"""Multi-GPU plumbing on the host side: one process per GPU, rows of R split over the ranks for the
U/F sweep and columns for the V/G sweep.  The exchange itself (RCCL all-gather of the freshly drawn
factor block after each half sweep + one all-reduce of a few scalars) happens inside
libbnmtf_hip.so; this module only (a) says which block a rank owns and (b) carries the 128-byte RCCL
id from rank 0 to the other ranks, plus a barrier and a max-reduction for timing, over a plain TCP
control plane (standard library sockets: no PyTorch, no MPI).

Launchers: anything that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (e.g.
`python -m torch.distributed.run`, which is only used as a process launcher here), or
`spawn_local(n, argv)` below, which starts n child processes of the current script on this node.
"""
import ctypes as C
import os
import socket
import struct
import subprocess
import sys
import time

import numpy as np

from . import _lib

_MAGIC = b"BNMTFCTL1"
_PORT_TRIES = 16


def shard_range(n, rank, world):
    """(first, count) of the contiguous block of `n` units owned by `rank` -- the library's own rule."""
    first, count = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().bnmtf_shard_range(int(n), int(rank), int(world), C.byref(first), C.byref(count)))
    return first.value, count.value


def make_comm_id():
    buf = np.zeros(128, dtype=np.uint8)
    _lib.check(_lib.lib().bnmtf_comm_unique_id(_lib.ptr(buf)))
    return bytes(buf)


def _send(sock, payload):
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv(sock):
    def exactly(n):
        out = b""
        while len(out) < n:
            chunk = sock.recv(n - len(out))
            if not chunk:
                raise ConnectionError("control plane: peer closed the connection")
            out += chunk
        return out
    (n,) = struct.unpack("<I", exactly(4))
    return exactly(n)


class ControlPlane(object):
    """Star over TCP: rank 0 listens on one of the ports MASTER_PORT+1 .. MASTER_PORT+16 (BNMTF_CTRL_PORT
    overrides the first candidate), the other ranks connect and identify themselves.  Three operations, all
    collective: broadcast(bytes from rank 0), allreduce_max(float), barrier()."""

    def __init__(self, rank, world, addr=None, port=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: sockets of ranks 1..world-1 (index rank-1)
        self.sock = None         # other ranks: socket to rank 0
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port or os.environ.get("BNMTF_CTRL_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + 1)
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for p in range(base, base + _PORT_TRIES):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr if self._is_local(addr) else "", p))       # the rendezvous address only, not every interface
                    break
                except OSError:
                    srv.close(); srv = None
            if srv is None:
                raise _lib.BnmtfError("control plane: no free port in %d..%d" % (base, base + _PORT_TRIES - 1))
            srv.listen(self.world)
            got = {}
            while len(got) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    srv.close()
                    raise _lib.BnmtfError("control plane: only %d of %d ranks joined within %.0f s" % (len(got) + 1, self.world, timeout))
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                # a stray connection that says nothing (or the wrong thing) is dropped after a short wait; it must not
                # hold up, let alone end, the job
                try:
                    conn.settimeout(5.0)
                    hello = _recv(conn)
                    (r,) = struct.unpack("<I", hello[len(_MAGIC):len(_MAGIC) + 4]) if hello.startswith(_MAGIC) and len(hello) >= len(_MAGIC) + 4 else (None,)
                    if r is None or not (1 <= r < self.world):
                        conn.close(); continue
                    if r in got:        # the rank gave up on its first attempt (its hello timed out while we were held up) and came back:
                        try:            # the earlier socket is dead on its side -- the new one is the peer
                            got[r].close()
                        except OSError:
                            pass
                        del got[r]
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(conn, _MAGIC)
                    got[r] = conn
                except (OSError, ConnectionError, struct.error):
                    conn.close()
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            while True:
                for p in range(base, base + _PORT_TRIES):
                    s = None
                    try:
                        s = socket.create_connection((addr, p), timeout=2.0)
                        s.settimeout(10.0)
                        _send(s, _MAGIC + struct.pack("<I", self.rank))
                        if _recv(s) == _MAGIC:
                            s.settimeout(timeout)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.sock = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        if s is not None:
                            s.close()
                if self.sock is not None:
                    break
                if time.time() > deadline:
                    raise _lib.BnmtfError("control plane: rank %d could not reach rank 0 at %s:%d.." % (self.rank, addr, base))
                time.sleep(0.05)

    @staticmethod
    def _is_local(addr):
        return addr in ("127.0.0.1", "localhost", "::1")

    def broadcast(self, payload=None):
        if self.world == 1:
            return payload
        if self.rank == 0:
            for s in self.peers:
                _send(s, payload)
            return payload
        return _recv(self.sock)

    def allreduce_max(self, value):
        if self.world == 1:
            return float(value)
        if self.rank == 0:
            vals = [float(value)] + [struct.unpack("<d", _recv(s))[0] for s in self.peers]
            m = max(vals)
            for s in self.peers:
                _send(s, struct.pack("<d", m))
            return m
        _send(self.sock, struct.pack("<d", float(value)))
        return struct.unpack("<d", _recv(self.sock))[0]

    def barrier(self):
        self.allreduce_max(0.0)

    def close(self):
        for s in self.peers + ([self.sock] if self.sock is not None else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.sock = [], None


def init_from_env():
    """(rank, world, local_rank, comm_id, control_plane) from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
    MASTER_PORT; rank 0 makes the RCCL id and the control plane carries it to the others."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    cp = ControlPlane(rank, world)
    if world == 1:
        return rank, world, local_rank, None, cp
    cid = cp.broadcast(make_comm_id() if rank == 0 else None)
    return rank, world, local_rank, cid, cp


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_local(n, argv, env=None):
    """Start n child processes of `argv` (one per GPU of this node: RANK = LOCAL_RANK = 0..n-1) and wait for
    them; returns the largest exit code.  The caller must not have touched the GPU."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    # all ranks are polled together: one that dies early takes the others with it instead of leaving them at a barrier
    # until their own timeout
    # (the exit code of a rank that was terminated HERE does not count: the code reported is the failing rank's own)
    rc = 0
    alive = list(procs)
    killed = set()
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if p not in killed:
                rc = max(rc, abs(code))
            if code != 0 and p not in killed:
                for q in alive:
                    if q.poll() is None:
                        killed.add(q)
                        q.terminate()
        if alive:
            time.sleep(0.05)
    return rc
