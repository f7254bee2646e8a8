"""Try the RCCL transport with TWO ranks on the ONE GPU a test box has (RCCL normally refuses duplicate devices; this
only tells whether the refusal is the only obstacle).  Usage: rccl_two_ranks_one_gpu.py"""
import multiprocessing as mp
import os
import sys

sys.path.insert(0, '.')


def worker(rank, world, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_PORT="29701", TORCHELASTIC_RUN_ID="x")
    import numpy as np
    from genomad_amd import rccl, sharding, synthetic
    from genomad_amd.engine import NNEngine
    rccl.prepare_env()
    try:
        eng = NNEngine(0, synthetic.synth_weights())
        comm = rccl.RcclComm(eng, rank, world, timeout=60)
        x = np.full((5, 3), rank + 1, np.float32)
        got = comm.gather_array(x)
        tot = comm.allgather_i64([rank * 10 + 1])
        scores = sharding.gather_scores(comm, np.full((4 if rank == 0 else 3, 3), rank, np.float32), 7)
        q.put((rank, None if got is None else got[:, 0, 0].tolist(), tot[:, 0].tolist(), None if scores is None else scores[:, 0].tolist()))
        comm.close()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, "ERROR", str(exc)[:300], None))


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, q)) for r in range(2)]
    for p in ps:
        p.start()
    for _ in ps:
        try:
            print(q.get(timeout=120))
        except Exception as exc:  # noqa: BLE001
            print("no answer:", exc)
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
