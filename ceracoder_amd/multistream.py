"""N independent streams, one per GPU/rank (the path does not shard, so there is
no data-path collective).  The only cross-rank traffic is the barrier pair that brackets a timed
region and the MAX of the per-rank wall times -- host tensors over gloo."""
import os
import time


class Ranks:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0])

    def timed(self, fn, sync=None):
        """barrier + device sync, run fn(), device sync + barrier; returns (max wall seconds over ranks, fn's result)."""
        self.barrier()
        if sync:
            sync()
        t0 = time.perf_counter()
        out = fn()
        if sync:
            sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), out

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
