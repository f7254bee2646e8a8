"""RCCL transport for :mod:`genomad_amd.sharding` — no torch: the collectives are the C ABI's ``gnn_comm_*``
entry points (genomad_amd/csrc/gnn_comm.hip, which dlopens librccl.so), called through ctypes.

One process per GPU, launched e.g. by ``python -m torch.distributed.run`` (only as a process launcher: RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_PORT are read from the environment; the launcher's own store is not used).
Bootstrap of the communicator: rank 0 asks RCCL for a unique id (``ncclGetUniqueId``) and publishes its 128
bytes, tagged with a hash of what identifies THIS launch and a timestamp, in a file of a per-user 0700
directory under the node-local temp directory (created exclusively, renamed into place; a leftover of a
crashed run is removed first and is never accepted by the readers: wrong tag or too old); the other ranks
poll for it; every rank then runs ``ncclCommInitRank`` on its own device.  (One node, as the bench contract
says.  Several nodes need a launcher that exports, identically on every node, GENOMAD_AMD_RDZV_DIR (a shared directory owned
by the user who runs the ranks) AND GENOMAD_AMD_RDZV_PARENT or GENOMAD_AMD_RDZV_NONCE: without one of those the launch tag
falls back to the parent pid, which differs between nodes, and the ranks would wait for different file names.)
"""
import contextlib
import ctypes as C
import hashlib
import os
import stat
import struct
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

from ._lib import check

ID_BYTES = 128
_SEQ = 0
_PROCESS_START = time.time()


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


_MAGIC = b"GNNRCCL2"
_STALE_S = 900.0          # a published id older than this can only be a leftover of another run


def _run_tag(seq: int) -> bytes:
    """16 bytes every rank of ONE launch derives identically and another launch does not: the launcher's port, run id,
    restart count and pid (the ranks' common parent), an optional nonce the launcher exports (bench.py's own spawner does),
    and the number of communicators this process has created so far."""
    parts = (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
             os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.environ.get("GENOMAD_AMD_RDZV_NONCE", ""),
             os.environ.get("GENOMAD_AMD_RDZV_PARENT", str(os.getppid())), seq)
    return hashlib.sha256("|".join(str(x) for x in parts).encode()).digest()[:16]


def _rdzv_dir() -> Path:
    """Directory of the id file: GENOMAD_AMD_RDZV_DIR (a shared one for several nodes), else a per-user directory under the
    temp dir, created 0700; a directory that is a symlink, is owned by someone else or is writable by others is refused
    (a predictable name in a world-writable /tmp must not be squattable)."""
    explicit = os.environ.get("GENOMAD_AMD_RDZV_DIR")
    d = Path(explicit) if explicit else Path(tempfile.gettempdir()) / f"genomad_amd_rdzv_{os.getuid()}"
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (not explicit and st.st_mode & 0o022):
        raise PermissionError(f"refusing the RCCL rendezvous directory {d}: not a directory of this user / writable by others")
    return d


def _id_file(seq: int) -> Path:
    explicit = os.environ.get("GENOMAD_AMD_RDZV_FILE")
    if explicit:
        return Path(f"{explicit}.{seq}")
    return _rdzv_dir() / f"rccl_{_run_tag(seq).hex()}.id"


def _publish_id(path: Path, tag: bytes, uid: bytes):
    """rank 0: remove whatever a crashed run left under this name, then create the file exclusively (0600, no symlink
    followed) under a temporary name and rename it into place: a reader sees either nothing or the whole record."""
    record = _MAGIC + tag + bytes(uid) + struct.pack("<d", time.time())
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    tmp = path.with_name(path.name + f".tmp{os.getpid()}")
    try:
        os.unlink(tmp)
    except FileNotFoundError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    try:
        os.write(fd, record)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read_id(path: Path, tag: bytes, started: float):
    """Another rank: the 128 id bytes once a COMPLETE record of THIS launch is there, else None.  A record with another
    tag, or published long before this process started (a leftover rank 0 has not replaced yet), is not accepted."""
    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except OSError:
        return None
    try:
        data = os.read(fd, 4096)
    finally:
        os.close(fd)
    want = len(_MAGIC) + 16 + ID_BYTES + 8
    if len(data) != want or data[:len(_MAGIC)] != _MAGIC or data[len(_MAGIC):len(_MAGIC) + 16] != tag:
        return None
    (published,) = struct.unpack("<d", data[-8:])
    if published < started - _STALE_S:
        return None
    return data[len(_MAGIC) + 16:len(_MAGIC) + 16 + ID_BYTES]


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
        # the rendezvous file exists only for world > 1: a single rank neither creates nor checks the per-user directory
        tag = _run_tag(_SEQ)
        path = _id_file(_SEQ) if self.world > 1 else None
        _SEQ += 1
        uid = (C.c_uint8 * ID_BYTES)()
        if self.rank == 0:
            with _c_stdout_to_stderr():
                check(self.lib.gnn_comm_unique_id(uid))
            if self.world > 1:
                _publish_id(path, tag, bytes(uid))
        else:
            deadline = time.time() + timeout
            while True:
                got = _read_id(path, tag, _PROCESS_START)
                if got is not None:
                    break
                if time.time() > deadline:
                    raise TimeoutError(f"rank {self.rank}: no RCCL unique id of this launch at {path} after {timeout:.0f} s")
                time.sleep(0.01)
            C.memmove(uid, got, ID_BYTES)
        with _c_stdout_to_stderr():
            check(self.lib.gnn_comm_init(self.ctx, self.world, self.rank, uid))
            self.barrier()
        if self.rank == 0 and self.world > 1:
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
