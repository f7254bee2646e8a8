"""gloo transport for genomad_amd.sharding — TEST INFRASTRUCTURE: lets the CPU suite run the multi-rank
logic (shard ranges, padded gathers, contig-result assembly, main()'s rank gating) with world size 2/3
without GPUs.  The product transport is genomad_amd.rccl.RcclComm."""
import numpy as np


class GlooComm:
    def __init__(self, rank, world, port):
        import os
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.rank, self.world = dist, rank, world

    def close(self):
        self.dist.destroy_process_group()

    def barrier(self):
        self.dist.barrier()

    def _allgather(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy())
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.numpy() for o in out])

    def allgather_i64(self, values):
        send = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
        return self._allgather(send).view(np.int64).reshape(self.world, len(send))

    def gather_array(self, arr, root=0):
        send = np.ascontiguousarray(arr)
        got = self._allgather(send)
        if self.rank != root:
            return None
        return got.view(send.dtype).reshape((self.world,) + send.shape)
