"""One rank of the multi-GPU tests (tests/test_multi_gpu.py), started D times by bench.spawn_ranks - i.e. the way a launcher
starts bench.py's ranks: RANK / LOCAL_RANK / WORLD_SIZE and a private rendezvous directory in the environment.

    multi_gpu_worker.py comm OUT.json            RcclComm over the rank's engine: every collective of genomad_amd/rccl.py
    multi_gpu_worker.py main FASTA OUTDIR        genomad_amd.nn_classification.main under the launcher's world

With GENOMAD_AMD_BENCH_FAKE_ENGINE=1 the engine is tests/fake_engine.py (CPU): the same code path over a fake comm library."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_comm(out_path):
    import numpy as np
    import bench
    from genomad_amd import rccl
    rccl.prepare_env()
    rank, world, local = rccl.world_from_env()
    eng, n_dev, local = bench.make_engine(local, None, 4096)
    comm = rccl.RcclComm(eng, rank, world, timeout=300.0)
    res = {"world": world, "devices": n_dev}
    # allgather of small int64 rows (control messages of sharding.py)
    got = comm.allgather_i64([rank, rank * rank, 7])
    res["allgather_ok"] = bool(np.array_equal(got, np.array([[r, r * r, 7] for r in range(world)], np.int64)))
    # gather of host arrays to a non-zero root as well
    for root in sorted({0, world - 1}):
        a = comm.gather_array(np.full((5, 3), rank + 0.25, np.float32), root=root)
        ok = (a is None) if rank != root else bool(np.array_equal(a, np.stack([np.full((5, 3), r + 0.25, np.float32) for r in range(world)])))
        res[f"gather_root{root}_ok"] = ok
    # ncclGather of DEVICE buffers (the one collective of the data path: 12 B per window to rank 0)
    n = 3000
    send, recv = eng.alloc(n * 12), (eng.alloc(world * n * 12) if rank == 0 else None)
    mine = (np.arange(n * 3, dtype=np.float32).reshape(n, 3) + 1000 * rank)
    send.upload(mine)
    comm.gather_dev(send.ptr, recv.ptr if recv is not None else None, n * 12, 0)
    eng.sync()
    if rank == 0:
        all_ = recv.download((world, n, 3), np.float32)
        res["gather_dev_ok"] = bool(all(np.array_equal(all_[r], np.arange(n * 3, dtype=np.float32).reshape(n, 3) + 1000 * r) for r in range(world)))
    res["allreduce_max_ok"] = comm.allreduce_max(float(rank) + 0.5) == world - 0.5
    comm.barrier()
    oks = comm.allgather_i64([int(all(v for k, v in res.items() if k.endswith("_ok")))])
    res["all_ranks_ok"] = bool(oks.all())
    comm.close()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)


def run_main(fasta, out_dir):
    from genomad_amd import nn_classification as nnc
    nnc.main(fasta, out_dir, False, 128, False, 1, False, False)


if __name__ == "__main__":
    if sys.argv[1] == "comm":
        run_comm(sys.argv[2])
    elif sys.argv[1] == "main":
        run_main(sys.argv[2], sys.argv[3])
    else:
        sys.exit(f"unknown mode {sys.argv[1]}")
