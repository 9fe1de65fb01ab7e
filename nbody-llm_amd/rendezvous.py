"""rendezvous.py -- the control plane of a multi-rank run, stdlib only (no torch, no MPI).

The engine needs exactly three things from outside to run G ranks: the 128-byte communicator id made by rank 0 reaches
every rank, a barrier around timed regions, and small per-rank records gathered on rank 0.  bench.py, the rank worker
of the tests and tools/ use this instead of `torch.distributed`: a process that imports torch loads torch's bundled
ROCm runtime (HIP 7.0.x, RCCL 2.26) ahead of /opt/rocm's, and libnbody_hip.so would then run on another runtime than
the one it was compiled against.

Rank 0 listens, ranks 1..G-1 connect (retrying until it is there).  Where: a unix-domain socket whose name is derived
from MASTER_PORT (+ TORCHELASTIC_RUN_ID) -- under `python -m torch.distributed.run` the TCP port MASTER_PORT itself
belongs to the launcher's own store, and all ranks of one node see the same values -- or an explicit path.
Messages are length-prefixed JSON; bytes travel as hex strings.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import tempfile
import time


def default_address() -> str:
    port = os.environ.get("MASTER_PORT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    safe = "".join(ch if ch.isalnum() else "_" for ch in f"{port}_{run}")[:60]
    return os.path.join(tempfile.gettempdir(), f"nbody_rdzv_{os.getuid()}_{safe}.sock")


def _send(sock: socket.socket, obj) -> None:
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock: socket.socket):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return json.loads(_recv_exact(sock, n).decode())


class Rendezvous:
    """Star topology around rank 0.  Every collective below is called by all ranks in the same order."""

    def __init__(self, rank: int, world: int, address: str | None = None, timeout: float = 120.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.address = address or default_address()
        self.peers: dict[int, socket.socket] = {}
        self.sock: socket.socket | None = None
        self._listener: socket.socket | None = None
        if self.world == 1:
            return
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            try:
                os.unlink(self.address)   # a socket file left by a run that died
            except FileNotFoundError:
                pass
            ls = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            ls.bind(self.address)
            ls.listen(self.world)
            self._listener = ls
            while len(self.peers) < self.world - 1:
                ls.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = ls.accept()
                except socket.timeout:
                    raise TimeoutError(f"rendezvous: {len(self.peers) + 1} of {self.world} ranks arrived within {self.timeout:.0f} s") from None
                conn.settimeout(self.timeout)
                hello = _recv(conn)
                r = int(hello["rank"])
                if hello.get("world") != self.world or r in self.peers or not (0 < r < self.world):
                    conn.close()
                    raise RuntimeError(f"rendezvous: unexpected hello {hello}")
                self.peers[r] = conn
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.address)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 is not listening on {self.address} after {self.timeout:.0f} s") from None
                    time.sleep(0.02)
            s.settimeout(self.timeout)
            _send(s, {"rank": self.rank, "world": self.world})
            self.sock = s

    # ---- collectives
    def gather(self, obj):
        """Rank 0 gets [obj of rank 0, ..., obj of rank G-1]; the others get None."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [None] * (self.world - 1)
            for r, c in self.peers.items():
                out[r] = _recv(c)
            return out
        _send(self.sock, obj)
        return None

    def bcast(self, obj=None):
        """Everybody gets rank 0's obj."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for c in self.peers.values():
                _send(c, obj)
            return obj
        return _recv(self.sock)

    def allgather(self, obj):
        return self.bcast(self.gather(obj))

    def barrier(self) -> None:
        self.bcast(self.gather(None) and None)

    def bcast_bytes(self, data: bytes | None) -> bytes:
        return bytes.fromhex(self.bcast(data.hex() if self.rank == 0 else None))

    def close(self) -> None:
        for c in self.peers.values():
            c.close()
        self.peers.clear()
        if self.sock is not None:
            self.sock.close()
            self.sock = None
        if self._listener is not None:
            self._listener.close()
            self._listener = None
            try:
                os.unlink(self.address)
            except OSError:
                pass
