"""Multi-GPU communicator of the product path: RCCL through the C ABI, no PyTorch.

One process per GPU.  The only data-path exchange is one SUM all-reduce (fp64, in place in HBM, on the
engine's stream) per chunk of orderings of the packed pending statistics -- the multi-device form of the
reference's merge_sample_mean / merge_sample_cov (cvxgrp/ls-spa ls_spa/ls_spa.py:103-119, :212-216).

    comm = NativeComm.from_env()          # RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT
    res = ls_spa(X_train, X_test, y_train, y_test, method="argsort", device=comm.local_rank, comm=comm)

RCCL needs one 128-byte id made on rank 0 to reach every rank before the communicator exists; it travels over
a plain TCP connection to ``MASTER_ADDR`` (``exchange_unique_id``).  The port is ``LSSPA_RDZV_PORT`` or
``MASTER_PORT + 29`` (``MASTER_PORT`` itself belongs to the launcher's own store when the job is started with
``torch.distributed.run``).
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import time

import numpy as np

from . import _native as N

ID_BYTES = 128
_MAGIC = b"LSSPA-RDZV-1"


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += part
    return buf


def exchange_unique_id(rank, world, addr, port, make_id, timeout=120.0):
    """Rank 0 calls make_id() -> bytes[128] and serves it to the world - 1 other ranks, which fetch it.
    Every rank returns the same bytes.  Peers announce their rank, so a stray connection is refused."""
    if world == 1:
        return make_id()
    deadline = time.monotonic() + timeout
    if rank == 0:
        uid = make_id()
        if len(uid) != ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            seen = set()
            while len(seen) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"rendezvous: only {len(seen)} of {world - 1} ranks reached "
                                       f"{addr}:{port} within {timeout:.0f} s") from None
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, len(_MAGIC) + 4)
                    except (ConnectionError, socket.timeout):
                        continue
                    peer = int.from_bytes(hello[len(_MAGIC):], "little")
                    if hello[:len(_MAGIC)] != _MAGIC or not (0 < peer < world) or peer in seen:
                        continue
                    conn.sendall(uid)
                    seen.add(peer)
        return uid
    last = None
    while time.monotonic() < deadline:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.sendall(_MAGIC + int(rank).to_bytes(4, "little"))
                return _recv_exact(conn, ID_BYTES)
        except (ConnectionRefusedError, ConnectionError, socket.timeout, OSError) as exc:
            last = exc
            time.sleep(0.05)
    raise TimeoutError(f"rendezvous: rank {rank} could not fetch the id from {addr}:{port}: {last}")


class NativeComm:
    """The driver's communicator interface (see ``_driver._Comm``) on ``lsspa_comm_*`` / ``lsspa_*_allreduce``."""

    def __init__(self, rank, world, local_rank=None, addr="127.0.0.1", port=None, force_collective=False):
        if not (0 <= rank < world):
            raise ValueError("need 0 <= rank < world")
        self.rank, self.world = int(rank), int(world)
        self.local_rank = int(rank if local_rank is None else local_rank)
        self._addr, self._port = addr, port
        self._force = force_collective     # tests: run the collective even in a world of one
        self._engine = None

    @classmethod
    def from_env(cls, **kw):
        env = os.environ
        port = env.get("LSSPA_RDZV_PORT")
        if port is None:
            port = int(env.get("MASTER_PORT", "29500")) + 29
        return cls(int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")),
                   local_rank=int(env.get("LOCAL_RANK", env.get("RANK", "0"))),
                   addr=env.get("MASTER_ADDR", "127.0.0.1"), port=int(port), **kw)

    # ---- binding to an engine (its GPU and stream) ------------------------------------------
    def bind(self, engine):
        """Create the RCCL communicator on the engine's context (collective over all ranks)."""
        if self._engine is engine:
            return
        if self._engine is not None:
            raise RuntimeError("this communicator is already bound to another engine")
        lib = engine._lib

        def make_id():
            buf = (C.c_uint8 * ID_BYTES)()
            if lib.lsspa_comm_unique_id(buf) != N.OK:
                msg = lib.lsspa_last_error(None)
                raise N.LSSPANativeError(f"lsspa_comm_unique_id failed: {msg.decode() if msg else ''}")
            return bytes(buf)

        if self._port is None and self.world > 1:
            raise ValueError("NativeComm needs a rendezvous port for world > 1 (use from_env())")
        uid = exchange_unique_id(self.rank, self.world, self._addr, self._port or 0, make_id)
        arr = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        engine._check(lib.lsspa_comm_init(engine._h, arr, self.rank, self.world))
        self._engine = engine

    def _bound(self, engine):
        if self._engine is None:
            self.bind(engine)
        elif engine is not self._engine:
            raise RuntimeError("collective called with an engine this communicator is not bound to")
        return engine

    @property
    def _active(self):
        return self.world > 1 or self._force

    # ---- the driver's interface ---------------------------------------------------------------
    def allreduce_pending(self, engine):
        if self._active:
            e = self._bound(engine)
            e._check(e._lib.lsspa_stats_allreduce(e._h))

    def allreduce_draws(self, engine):
        if self._active:
            e = self._bound(engine)
            e._check(e._lib.lsspa_error_allreduce(e._h))

    def allreduce_reduction(self, engine):
        if self._active:
            e = self._bound(engine)
            e._check(e._lib.lsspa_reduce_allreduce(e._h))

    def sum_ints(self, values):
        vals = np.ascontiguousarray([int(v) for v in values], dtype=np.int64)
        if self._active:
            if self._engine is None:
                raise RuntimeError("bind(engine) before the first integer collective")
            e = self._engine
            e._check(e._lib.lsspa_comm_sum_i64(e._h, vals.ctypes.data_as(C.POINTER(C.c_int64)), len(vals)))
        return [int(v) for v in vals]

    def gather_ints(self, values):
        from ._driver import gather_ints_by_sum
        return gather_ints_by_sum(self, values)

    def gather_lifts(self, local, counts):
        """All ranks' per-sample lift vectors (only for attribution_history / the host low-rank estimator)."""
        if self.world == 1:
            return local
        e = self._engine
        p = local.shape[1]
        width = max(counts)
        mine = np.zeros((width, p))
        mine[: len(local)] = local
        out = np.empty((self.world, width, p))
        e._check(e._lib.lsspa_comm_allgather(e._h, N.dptr(mine), mine.size, N.dptr(out)))
        return [out[r, : counts[r]] for r in range(self.world)]

    def barrier(self):
        self.sum_ints([0])

    def close(self):
        if self._engine is not None and getattr(self._engine, "_h", None):
            self._engine._lib.lsspa_comm_destroy(self._engine._h)
        self._engine = None
