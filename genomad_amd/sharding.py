"""Data-parallel sharding of windows / contigs across GPUs (one process per GPU).

Windows are independent (every score depends only on its own 6000 bytes and the replicated weights), so
the path shards with no data-path collective: rank r classifies a contiguous range of windows, and the
per-window scores (12 B each) are collected on rank 0 with ONE gather at the end (the loop
being sharded is nn_classification.py:316-320 of the reference).  Contiguous ranges keep every contig on
at most two ranks and preserve window order, so the per-contig segment mean runs on rank 0 over the
gathered array and the result is bit-identical for any number of ranks (no cross-rank reduction).

This module is transport-agnostic and imports neither torch nor the HIP library.  A transport ("comm") is
any object with

    rank, world                        ints
    barrier()
    allgather_i64(values)              list of ints -> int64 array (world, len(values))
    gather_array(arr, root=0)          same-shape/dtype numpy array from every rank -> (world, *shape) on
                                       ``root``, None elsewhere

The product transport is :class:`genomad_amd.rccl.RcclComm` (RCCL over xGMI through the C ABI's
``gnn_comm_*``); ``LocalComm`` below is the one-process case; the CPU test-suite drives the same functions
with a gloo transport (tests/gloo_comm.py, world size 2 and 3).
"""
import json
from typing import List, Optional, Tuple

import numpy as np


def shard_range(n_windows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Half-open window range of ``rank``: ceil(n/G) windows per rank, the tail ranks get fewer
    (possibly zero)."""
    if world_size < 1 or not (0 <= rank < world_size) or n_windows < 0:
        raise ValueError("bad shard arguments")
    per = -(-n_windows // world_size)
    a = min(rank * per, n_windows)
    return a, min(a + per, n_windows)


def shard_counts(n_windows: int, world_size: int) -> List[int]:
    return [b - a for a, b in (shard_range(n_windows, world_size, r) for r in range(world_size))]


class LocalComm:
    """The one-process transport (world size 1)."""
    rank, world = 0, 1

    def barrier(self):
        pass

    def allgather_i64(self, values):
        return np.asarray(values, dtype=np.int64).reshape(1, -1)

    def gather_array(self, arr, root=0):
        return np.asarray(arr)[None]


def gather_scores(comm, local_scores: np.ndarray, n_windows: int, root: int = 0) -> Optional[np.ndarray]:
    """Collect the (n_local, 3) float32 score shards of all ranks on ``root`` in window order.  Shards are
    padded to ceil(n/G) rows so that a single fixed-size gather suffices.  Returns (n_windows, 3) on
    ``root`` and None elsewhere."""
    per = -(-n_windows // comm.world)
    a, b = shard_range(n_windows, comm.world, comm.rank)
    local = np.ascontiguousarray(local_scores, dtype=np.float32).reshape(-1, 3)
    if len(local) != b - a:
        raise ValueError(f"rank {comm.rank}: expected shard of {b - a} windows, got {len(local)}")
    send = np.zeros((per, 3), np.float32)
    send[: b - a] = local
    got = comm.gather_array(send, root)
    if got is None:
        return None
    return got.reshape(comm.world * per, 3)[:n_windows]


def classify_sharded(windows, score_fn, comm=None):
    """Score ``windows`` (n, 6000) with ``score_fn`` on this rank's contiguous shard and collect the (n, 3)
    scores on rank 0 (None on the other ranks).  Without a transport this is just ``score_fn(windows)``."""
    if comm is None or comm.world == 1:
        return np.ascontiguousarray(score_fn(windows), dtype=np.float32)
    a, b = shard_range(len(windows), comm.world, comm.rank)
    return gather_scores(comm, score_fn(windows[a:b]), len(windows))


def gather_bytes(comm, payload: bytes, root: int = 0) -> Optional[List[bytes]]:
    """Variable-length byte strings of every rank on ``root`` (None elsewhere): sizes travel in one small
    all-gather, the payloads padded to the longest in one gather."""
    sizes = comm.allgather_i64([len(payload)])[:, 0]
    width = max(int(sizes.max()), 1)
    send = np.zeros(width, np.uint8)
    send[: len(payload)] = np.frombuffer(payload, np.uint8)
    got = comm.gather_array(send, root)
    if got is None:
        return None
    return [got[r, : int(sizes[r])].tobytes() for r in range(comm.world)]


def contig_subset(offsets, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous contig range [a, b) of ``rank`` when every rank holds the WHOLE contig table (compressed
    inputs, which cannot be read by byte range): ranges are balanced by sequence length (windows are
    proportional to it) and never split a contig."""
    off = np.asarray(offsets, dtype=np.int64)
    n = len(off) - 1
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    total = int(off[-1])
    cut = lambda r: int(np.searchsorted(off[:-1], total * r // world, side="left")) if r < world else n  # noqa: E731
    return (0 if rank == 0 else cut(rank)), cut(rank + 1)


def gather_contig_parts(comm, parts, root: int = 0):
    """Contig-sharded results -> ``root``.  ``parts`` = this rank's list of (order_key, names (k,),
    predictions (k, 3) float32, window_ids) — one entry per piece of the input it classified, window ids
    counting contigs from 0 inside the piece.  A contig never straddles pieces, so every per-contig mean is
    local and the assembled table is bit-identical to a single-process run.  The scores travel in ONE
    gather; names, window ids and the piece table are small host objects and go as bytes.  ``root`` puts the
    pieces of all ranks in ``order_key`` order (= file order) and returns (names, predictions, window_ids,
    total_windows); the other ranks get (None, None, None, total_windows) — the window total is known
    everywhere so that all ranks take the same exit."""
    comm = comm or LocalComm()
    parts = list(parts)
    table = np.array([[int(key), len(nm), len(ids)] for key, nm, _, ids in parts], dtype="<i8").reshape(-1, 3)
    k = int(table[:, 1].sum()) if len(table) else 0
    meta = comm.allgather_i64([k, int(table[:, 2].sum()) if len(table) else 0])
    total_windows = int(meta[:, 1].sum())
    per = max(int(meta[:, 0].max()), 1)
    send = np.zeros((per, 3), np.float32)
    if k:
        send[:k] = np.concatenate([np.ascontiguousarray(pr, dtype=np.float32).reshape(-1, 3) for _, _, pr, _ in parts])
    preds = comm.gather_array(send, root)
    names = [str(n) for _, nm, _, _ in parts for n in nm]
    ids = np.concatenate([np.asarray(i, dtype="<i8") for _, _, _, i in parts]) if parts else np.zeros(0, "<i8")
    table_blobs = gather_bytes(comm, table.tobytes(), root)
    name_blobs = gather_bytes(comm, json.dumps(names).encode("utf-8", "surrogateescape"), root)
    id_blobs = gather_bytes(comm, ids.tobytes(), root)
    if comm.rank != root:
        return None, None, None, total_windows
    pieces = []                      # (order_key, rank, contig slice, window slice)
    r_names, r_ids = [], []
    for r in range(comm.world):
        t = np.frombuffer(table_blobs[r], dtype="<i8").reshape(-1, 3)
        r_names.append(json.loads(name_blobs[r].decode("utf-8", "surrogateescape")))
        r_ids.append(np.frombuffer(id_blobs[r], dtype="<i8").astype(np.int64))
        c0 = w0 = 0
        for key, nc, nw in t:
            pieces.append((int(key), r, c0, c0 + int(nc), w0, w0 + int(nw)))
            c0, w0 = c0 + int(nc), w0 + int(nw)
    pieces.sort()
    if len({p[0] for p in pieces}) != len(pieces):
        raise ValueError("duplicate piece keys in gather_contig_parts")
    out_names, out_preds, out_ids, base = [], [], [], 0
    for _, r, c0, c1, w0, w1 in pieces:
        out_names.extend(r_names[r][c0:c1])
        out_preds.append(preds[r][c0:c1])
        out_ids.append(r_ids[r][w0:w1] + base)
        base += c1 - c0
    return (np.array(out_names) if out_names else np.zeros(0, dtype="<U1"),
            np.concatenate(out_preds, axis=0) if out_preds else np.zeros((0, 3), np.float32),
            np.concatenate(out_ids) if out_ids else np.zeros(0, np.int64), total_windows)


def gather_contig_results(comm, names, predictions, window_ids, root: int = 0):
    """One piece per rank, in rank order (see :func:`gather_contig_parts`)."""
    comm = comm or LocalComm()
    return gather_contig_parts(comm, [(comm.rank, names, predictions, window_ids)], root)


def broadcast_flags(comm, flags, root: int = 0):
    """Small ints decided on ``root`` (e.g. "skip this stage") made known to every rank."""
    got = comm.allgather_i64([int(f) for f in flags])
    return [int(v) for v in got[root]]


def accession_digests(accessions) -> np.ndarray:
    """64-bit digests of record accessions, as a uint64 array: what travels instead of the strings (the same function as the
    native one-pass :func:`genomad_amd.sequence.accession_digests_of_text`)."""
    from .sequence import _digest_of_accession
    return np.array([_digest_of_accession(a) for a in accessions], dtype="<u8")


def fasta_verdict(comm, accessions, exact_check, root: int = 0) -> bool:
    """check_fasta (genomad/sequence.py:124-131: False for a file without records or with two records of one accession) when
    every rank has seen only ITS records: ``accessions`` = the accessions of all records of this rank's share of the file (no N
    stripping: records that the classification pass drops count) - or, as the product path passes them, their uint64 digests
    already (``sequence.accession_digests_of_text``).  ONE gather of 64-bit digests to ``root``; equal digests, inside a share or
    across shares, are either a true duplicate or a collision (about 1e-7 for a million records): ``exact_check()`` - the
    sequential whole-file check - decides, on ``root`` only.  Every rank returns the same verdict."""
    comm = comm or LocalComm()
    if isinstance(accessions, np.ndarray) and accessions.dtype.kind == "u":
        mine = np.ascontiguousarray(accessions, dtype="<u8")        # digests already (sequence.accession_digests_of_text: one native pass)
    else:
        mine = accession_digests(list(accessions))
    local_dup = len(np.unique(mine)) != len(mine)                   # equal digests inside a share: a duplicate, or a collision
    meta = comm.allgather_i64([len(mine), int(local_dup)])
    blobs = gather_bytes(comm, mine.tobytes() if not meta[:, 1].any() else b"", root)
    ok = True
    if comm.rank == root:
        if int(meta[:, 0].sum()) == 0:
            ok = False
        elif meta[:, 1].any():
            ok = bool(exact_check())
        else:
            d = np.sort(np.concatenate([np.frombuffer(b, dtype="<u8") for b in blobs]))
            if len(d) > 1 and bool((d[1:] == d[:-1]).any()):
                ok = bool(exact_check())
    return bool(broadcast_flags(comm, [ok], root)[0])
