"""Process-group plumbing for the sharded benchmark: one process per GPU, independent frames,
no collective on the data path.  Only the timing needs communication (a barrier and a MAX
reduce); `backend="nccl"` is RCCL on ROCm, `"gloo"` is used by the CPU tests."""
import os


class Group:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        # AGX_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a single GPU)
        if self.world > 1 or os.environ.get("AGX_FORCE_DIST") == "1":
            import torch.distributed as dist

            kwargs = {}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend=backend, **kwargs)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """MAX of a python float over all ranks (every rank gets the result)."""
        if not self.dist:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if not self.dist:
            return int(value)
        import torch

        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_rows(self, row):
        """every rank's list of floats, in rank order (all ranks get the table): the per-rank timing / power / clock figures of
        the benchmark, so that a slow or throttled rank is visible beside the MAX (None entries travel as NaN)"""
        vals = [float("nan") if v is None else float(v) for v in row]
        if not self.dist:
            return [vals]
        import torch

        dev = self.device if self.device is not None else "cpu"
        mine = torch.tensor(vals, dtype=torch.float64, device=dev)
        table = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(table, mine)
        return [[float(v) for v in t.cpu().tolist()] for t in table]

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
            self.dist = None


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` polynomials owned by `rank` of `world`."""
    return total * rank // world, total * (rank + 1) // world


def aggregate_throughput(units_per_rank_per_step, steps, world, max_elapsed_s):
    """whole-job units/s: all ranks' units over the slowest rank's time"""
    return units_per_rank_per_step * steps * world / max_elapsed_s
