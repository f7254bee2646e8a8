"""CPU stand-in for NNEngine + libgenomad_nn_hip.so's gnn_comm_*, for ONE purpose: running bench.py's multi-rank plumbing
(self-spawn, contiguous shards, the gather, the per-rank fields of the JSON line) on a box without GPUs.

TEST INFRASTRUCTURE ONLY.  bench.py selects it with GENOMAD_AMD_BENCH_FAKE_ENGINE=1 and stamps the line "fake_engine": true;
nothing under genomad_amd/ knows it exists.  "Device memory" is numpy memory (pointers are real host addresses, so bench.py's
pointer arithmetic works); a synthesised window carries its index in its first 8 bytes; the scores of window i are the
committed reference-graph scores of tests/golden/config2_golden.npz for i < 10 000 (so bench.py's golden check runs for
real) and a fixed function of i beyond.  The collectives go through files of the rendezvous directory the launcher made.
"""
import ctypes as C
import os
import time
from pathlib import Path

import numpy as np

_GOLDEN = None


def _golden():
    global _GOLDEN
    if _GOLDEN is None:
        p = Path(__file__).resolve().parent / "golden" / "config2_golden.npz"
        _GOLDEN = np.load(p)["scores_refgraph32"].astype(np.float32)
    return _GOLDEN


def _scores_of(idx: np.ndarray) -> np.ndarray:
    g = _golden()
    out = np.empty((len(idx), 3), np.float32)
    lo = idx < len(g)
    out[lo] = g[idx[lo]]
    hi = idx[~lo].astype(np.float64)
    a, b = 0.2 + 0.5 * ((hi * 0.6180339887) % 1.0), 0.1 + 0.15 * ((hi * 0.3247179572) % 1.0)
    out[~lo] = np.stack([a, b, 1.0 - a - b], axis=1).astype(np.float32)
    return out


class _Buffer:
    def __init__(self, nbytes):
        self.arr = np.zeros(max(int(nbytes), 1), np.uint8)
        self.nbytes, self.ptr = int(nbytes), self.arr.ctypes.data

    def download(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.arr[:n].view(dtype).reshape(shape).copy()

    def upload(self, a):
        a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        self.arr[:len(a)] = a

    def free(self):
        pass


def _view(ptr, nbytes):
    return np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(int(ptr)))


class _FakeLib:
    """gnn_comm_* over files of GENOMAD_AMD_RDZV_DIR (one file per collective call and rank), plus the few other entry points
    bench.py calls on `eng.lib` directly."""

    def __init__(self):
        # One directory per LAUNCH: bench.py's own spawner exports a private GENOMAD_AMD_RDZV_DIR; under another launcher
        # (torch.distributed.run) the directory is named after what identifies the launch - the same tag genomad_amd/rccl.py keys the
        # real unique id on -, so that the files of an earlier run can never stand in for a rank that has not arrived yet (they
        # did, in /tmp: a rank "joined" at once, rank 0 removed the id file, the other rank waited for it until its time-out)
        explicit = os.environ.get("GENOMAD_AMD_RDZV_DIR")
        if explicit:
            self.dir = Path(explicit)
        else:
            import tempfile
            from genomad_amd import rccl
            self.dir = Path(tempfile.gettempdir()) / f"genomad_amd_fake_comm_{os.getuid()}_{rccl._run_tag(0).hex()}"
            self.dir.mkdir(mode=0o700, exist_ok=True)
        self.own_dir = not explicit
        self.seq, self.world, self.rank = 0, 1, 0

    # -- bootstrap
    def gnn_comm_unique_id(self, uid):
        C.memmove(uid, bytes((i * 5 + 1) % 256 for i in range(128)), 128)
        return 0

    def gnn_comm_init(self, ctx, world, rank, uid):
        self.world, self.rank, self.uid = int(world), int(rank), bytes(uid)
        self._exchange(np.frombuffer(self.uid, np.uint8))       # returns once every rank has joined, like ncclCommInitRank
        return 0

    def gnn_comm_destroy(self, ctx):
        if self.world > 1 and self.own_dir:
            self._exchange(np.zeros(1, np.uint8))         # every rank is done reading
            if self.rank == 0:
                import shutil
                time.sleep(0.3)
                shutil.rmtree(self.dir, ignore_errors=True)
        return 0

    def gnn_comm_info(self, ctx, ranks, rank):
        ranks._obj.value, rank._obj.value = self.world, self.rank
        return 0

    def _exchange(self, mine: np.ndarray):
        """every rank contributes one array; returns the list of all ranks' arrays"""
        self.seq += 1
        if self.world == 1:
            return [mine]
        me = self.dir / f"fake_{self.seq}_{self.rank}.npy"
        tmp = self.dir / f"fake_{self.seq}_{self.rank}.tmp.npy"
        np.save(tmp, mine)
        os.replace(tmp, me)
        out, deadline = [], time.time() + 120
        for r in range(self.world):
            p = self.dir / f"fake_{self.seq}_{r}.npy"
            while not p.exists():
                if time.time() > deadline:
                    raise TimeoutError(f"fake comm: rank {r} never reached collective {self.seq}")
                time.sleep(0.002)
            out.append(np.load(p))
        return out

    def gnn_comm_barrier(self, ctx):
        self._exchange(np.zeros(1, np.uint8))
        return 0

    def gnn_comm_allgather(self, ctx, send, recv, nbytes):
        parts = self._exchange(_view(send, nbytes).copy())
        _view(recv, nbytes * self.world)[:] = np.concatenate(parts)
        return 0

    def gnn_comm_gather(self, ctx, send, recv, nbytes, root):
        parts = self._exchange(_view(send, nbytes).copy() if nbytes else np.zeros(0, np.uint8))
        if self.rank == root and nbytes:
            _view(recv, nbytes * self.world)[:] = np.concatenate(parts)
        return 0

    gnn_comm_gather_dev = gnn_comm_gather

    def gnn_comm_allreduce_max(self, ctx, value, n):
        v = value._obj if hasattr(value, "_obj") else value
        v.value = max(float(p[0]) for p in self._exchange(np.array([v.value], np.float64)))
        return 0

    # -- probes bench.py reports beside the roofline
    def gnn_mfma_probe(self, ctx, ms, out):
        out._obj.value = 1800.0
        return 0

    def gnn_mfma_probe_kind(self, ctx, kind, ms, out):
        out._obj.value = 1600.0
        return 0


class FakeEngine:
    def __init__(self, device=0, weights=None, chunk=None):
        self.lib, self.ctx, self.device, self.chunk = _FakeLib(), object(), int(device), int(chunk or 4096)
        self._prof, self._front_ms, self._launches = False, 0.0, 0

    def device_info(self):
        return {"name": "fake engine (CPU)", "cus": 256, "hbm_bytes": 0}

    def alloc(self, nbytes):
        return _Buffer(nbytes)

    def sync(self):
        pass

    flush = sync

    def synth_windows_dev(self, first, n, ptr, seed=1234):
        w = _view(ptr, n * 6000).reshape(n, 6000)
        w[:, :8] = np.arange(first, first + n, dtype=np.int64).view(np.uint8).reshape(n, 8)

    def classify_dev(self, bases_ptr, n, scores_ptr, precision="f16x3"):
        idx = _view(bases_ptr, n * 6000).reshape(n, 6000)[:, :8].copy().view(np.int64)[:, 0]
        _view(scores_ptr, n * 12).view(np.float32).reshape(n, 3)[:] = _scores_of(idx)
        if self._prof:
            launches = -(-n // self.chunk)
            self._launches += launches
            self._front_ms += 1e-3 * n          # 1 us per window: any positive duration will do

    classify_dev_async = classify_dev

    def profile_enable(self, on=True):
        self._prof = bool(on)

    def profile_reset(self):
        self._front_ms, self._launches = 0.0, 0

    def profile_get(self, kernel_id):
        return (self._front_ms, self._launches) if kernel_id in (0, 3) else (0.0, self._launches)
