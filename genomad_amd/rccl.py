"""RCCL transport for :mod:`genomad_amd.sharding` — no torch: the collectives are the C ABI's ``gnn_comm_*``
entry points (genomad_amd/csrc/gnn_comm.hip, which dlopens librccl.so), called through ctypes.

One process per GPU, launched e.g. by ``python -m torch.distributed.run`` (only as a process launcher: RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_PORT are read from the environment; the launcher's own store is not used).
Bootstrap of the communicator: rank 0 asks RCCL for a unique id (``ncclGetUniqueId``) and publishes its 128
bytes in a file in the node-local temp directory; the other ranks read it; every rank then runs
``ncclCommInitRank`` on its own device.  (One node, as the bench contract says; a shared directory can be
named with GENOMAD_AMD_RDZV_DIR otherwise.)
"""
import contextlib
import ctypes as C
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

from ._lib import check

ID_BYTES = 128
_SEQ = 0


def world_from_env():
    """(rank, world, local_rank) as the launcher exported them; (0, 1, 0) for a plain process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not (0 <= rank < max(world, 1)):
        raise ValueError(f"RANK={rank} outside WORLD_SIZE={world}")
    return rank, world, local


def prepare_env():
    """Environment the HIP runtime / RCCL need, to be called BEFORE the first HIP call of the process.

    * The reference module exports CUDA_VISIBLE_DEVICES=-1 at import (nn_classification.py:8) to keep
      TensorFlow off the GPU; HIP honours that variable, so it is removed.
    * multi-process device sharing on this driver needs the dmabuf IPC mode."""
    if os.environ.get("CUDA_VISIBLE_DEVICES") == "-1":
        del os.environ["CUDA_VISIBLE_DEVICES"]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _id_file(seq: int) -> Path:
    explicit = os.environ.get("GENOMAD_AMD_RDZV_FILE")
    if explicit:
        return Path(f"{explicit}.{seq}")
    base = Path(os.environ.get("GENOMAD_AMD_RDZV_DIR", tempfile.gettempdir()))
    tag = "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    os.getppid(), seq))
    return base / f"genomad_amd_rccl_{tag}.id"


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """RCCL prints a version banner with C stdio when the first communicator is created; callers such as bench.py
    promise exactly one JSON line on stdout.  File descriptor 1 is pointed at stderr for the duration, and the C
    library's buffers are flushed on both sides of the switch (with stdout redirected to a file, stdio holds the
    banner back until exit otherwise, and it would land behind the JSON line)."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class RcclComm:
    """Transport over ``gnn_comm_*`` for the engine's context (see sharding.py for the interface)."""

    def __init__(self, engine, rank: int, world: int, timeout: float = 300.0):
        global _SEQ
        self.engine, self.lib, self.ctx = engine, engine.lib, engine.ctx
        self.rank, self.world = int(rank), int(world)
        path = _id_file(_SEQ)
        _SEQ += 1
        uid = (C.c_uint8 * ID_BYTES)()
        if self.rank == 0:
            with _c_stdout_to_stderr():
                check(self.lib.gnn_comm_unique_id(uid))
            tmp = path.with_suffix(f".tmp{os.getpid()}")
            tmp.write_bytes(bytes(uid))
            os.replace(tmp, path)
        else:
            deadline = time.time() + timeout
            while not (path.is_file() and path.stat().st_size == ID_BYTES):
                if time.time() > deadline:
                    raise TimeoutError(f"rank {self.rank}: no RCCL unique id at {path} after {timeout:.0f} s")
                time.sleep(0.01)
            C.memmove(uid, path.read_bytes(), ID_BYTES)
        with _c_stdout_to_stderr():
            check(self.lib.gnn_comm_init(self.ctx, self.world, self.rank, uid))
            self.barrier()
        if self.rank == 0:
            try:
                path.unlink()
            except OSError:
                pass

    def close(self):
        if self.ctx is not None and getattr(self.engine, "ctx", None):
            self.lib.gnn_comm_destroy(self.ctx)
        self.ctx = None

    # ---- the sharding.py transport interface
    def barrier(self):
        check(self.lib.gnn_comm_barrier(self.ctx))

    def allgather_i64(self, values):
        send = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
        recv = np.empty((self.world, len(send)), np.int64)
        check(self.lib.gnn_comm_allgather(self.ctx, send.ctypes.data, recv.ctypes.data, send.nbytes))
        return recv

    def gather_array(self, arr, root=0):
        send = np.ascontiguousarray(arr)
        recv = np.empty((self.world,) + send.shape, send.dtype) if self.rank == root else None
        check(self.lib.gnn_comm_gather(self.ctx, send.ctypes.data, recv.ctypes.data if recv is not None else None,
                                       send.nbytes, int(root)))
        return recv

    # ---- extras used by bench.py
    def gather_dev(self, send_ptr: int, recv_ptr: int, nbytes: int, root: int = 0):
        """ncclGather of device buffers, asynchronous on the engine's stream (rccl.h:745)."""
        check(self.lib.gnn_comm_gather_dev(self.ctx, send_ptr, recv_ptr, int(nbytes), int(root)))

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(float(value))
        check(self.lib.gnn_comm_allreduce_max(self.ctx, C.byref(v), 1))
        return v.value


_COMMS = {}


def comm_for(engine):
    """The communicator of ``engine`` for the launcher's world (created on first use; LocalComm-equivalent
    None for a single process)."""
    rank, world, _ = world_from_env()
    if world <= 1:
        return None
    key = id(engine)
    if key not in _COMMS:
        _COMMS[key] = RcclComm(engine, rank, world)
    return _COMMS[key]
