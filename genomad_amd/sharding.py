"""Data-parallel sharding of windows across GPUs (one process per GPU).

Windows are independent (every score depends only on its own 6000 bytes and the replicated
weights), so the path shards with no data-path collective: rank r classifies a contiguous range
of windows, and the per-window scores (12 B each) are collected on rank 0 with ONE gather at the
end — ``torch.distributed`` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
Contiguous ranges keep every contig on at most two ranks and preserve window order, so the
per-contig segment mean (nn_classification.py:320) runs on rank 0 over the gathered array and the
result is bit-identical for any number of ranks (no cross-rank reduction is involved).

torch is imported lazily and only here: it is plumbing (process group, gather), not arithmetic.
"""
from typing import List, Tuple


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


def gather_scores(local_scores, n_windows: int, group=None, dst: int = 0):
    """Collect the (n_local, 3) float32 score shards of all ranks on ``dst`` in window order.

    ``local_scores`` is a torch tensor (CUDA for nccl, CPU for gloo) holding this rank's shard as
    produced for :func:`shard_range`.  Shards are padded to ceil(n/G) rows so that a single
    fixed-size gather suffices.  Returns the (n_windows, 3) tensor on ``dst`` and None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-n_windows // world)
    a, b = shard_range(n_windows, world, rank)
    if tuple(local_scores.shape) != (b - a, 3):
        raise ValueError(f"rank {rank}: expected shard of shape {(b - a, 3)}, got {tuple(local_scores.shape)}")
    send = local_scores
    if b - a != per:
        send = torch.zeros((per, 3), dtype=local_scores.dtype, device=local_scores.device)
        send[: b - a] = local_scores
    send = send.contiguous()
    if world == 1:
        return send[:n_windows]
    recv = None
    if rank == dst:
        recv = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(recv, dim=0)[:n_windows]


def classify_sharded(windows, score_fn):
    """Score ``windows`` (n, 6000) with ``score_fn`` on this rank's contiguous shard and collect the
    (n, 3) scores on rank 0 (returns None on the other ranks).  Without an initialised process group
    this is just ``score_fn(windows)``."""
    import numpy as np
    try:
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except ImportError:
        distributed = False
    if not distributed:
        return np.ascontiguousarray(score_fn(windows), dtype=np.float32)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    a, b = shard_range(len(windows), world, rank)
    local = torch.from_numpy(np.ascontiguousarray(score_fn(windows[a:b]), dtype=np.float32).reshape(b - a, 3))
    if dist.get_backend() == "nccl":
        local = local.cuda()
    out = gather_scores(local, len(windows))
    return None if out is None else out.cpu().numpy()


def ensure_process_group():
    """Under ``python -m torch.distributed.run`` (WORLD_SIZE > 1) make sure the default process
    group exists: "nccl" (= RCCL) with the GPU of LOCAL_RANK, "gloo" without a GPU.  torch is
    imported here BEFORE the HIP library is loaded so that both share one HIP runtime.  Returns
    (rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("GENOMAD_AMD_DIST_BACKEND")      # "gloo": e.g. several ranks sharing one GPU
        if backend == "gloo":
            dist.init_process_group("gloo")
        elif torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    return dist.get_rank(), dist.get_world_size()


def contig_subset(offsets, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous contig range [a, b) of ``rank`` when every rank holds the WHOLE contig table
    (compressed inputs, which cannot be read by byte range): ranges are balanced by sequence length
    (windows are proportional to it) and never split a contig."""
    import numpy as np
    off = np.asarray(offsets, dtype=np.int64)
    n = len(off) - 1
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    total = int(off[-1])
    cut = lambda r: int(np.searchsorted(off[:-1], total * r // world, side="left")) if r < world else n  # noqa: E731
    return (0 if rank == 0 else cut(rank)), cut(rank + 1)


def gather_contig_results(names, predictions, window_ids, dst: int = 0):
    """Contig-sharded results -> ``dst``.  Every rank holds the per-contig scores of its OWN
    contiguous run of contigs (a contig never straddles ranks, so the per-contig mean is local and
    the gathered table is bit-identical to a single-process run): ``names`` (k,), ``predictions``
    (k,3) float32, ``window_ids`` (local contig id of every kept window).  The scores travel in ONE
    tensor gather (RCCL on GPUs); names / window ids are small host objects.  Returns
    (names, predictions, window_ids, total_windows) on ``dst`` and (None, None, None, total_windows)
    elsewhere — the window total is known everywhere so that all ranks take the same exit."""
    import numpy as np
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    k = len(names)
    meta = [None] * world
    dist.all_gather_object(meta, (k, int(len(window_ids))))
    counts = [m[0] for m in meta]
    total_windows = sum(m[1] for m in meta)
    per = max(max(counts), 1)
    send = torch.zeros((per, 3), dtype=torch.float32, device=dev)
    if k:
        send[:k] = torch.from_numpy(np.ascontiguousarray(predictions, dtype=np.float32).reshape(k, 3)).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    host = [None] * world if rank == dst else None
    dist.gather_object((list(names), np.asarray(window_ids, dtype=np.int64)), host, dst=dst)
    if rank != dst:
        return None, None, None, total_windows
    all_names, all_ids, base = [], [], 0
    for r in range(world):
        nm, ids = host[r]
        all_names.extend(nm)
        all_ids.append(ids + base)
        base += counts[r]
    preds = torch.cat([recv[r][:counts[r]] for r in range(world)], dim=0).cpu().numpy()
    return np.array(all_names), preds, (np.concatenate(all_ids) if all_ids else np.zeros(0, np.int64)), total_windows
