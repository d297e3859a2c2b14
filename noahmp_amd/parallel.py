"""One process per GPU (= one MPI rank per GPU, mpp/module_mpp_land.F90) over torch.distributed.

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
The column physics needs no data-path collective: these helpers only carry the tile
assignment, the timing barrier and the metric reductions.  The ZWTXY halo of the MMF lateral-flow
stencil is the only physics exchange and lives with that kernel.
"""
import os

from .partition import partition, neighbours


class Comm:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend

    # ---- tile assignment (mpp_land_partition_calc, mpp:227-288)
    def my_tile(self, global_nx, global_ny):
        return partition(global_nx, global_ny, self.world)[self.rank]

    def my_neighbours(self):
        return neighbours(self.rank, self.world)

    # ---- timing / metric plumbing
    def _dev(self):
        import torch
        return torch.device("cuda", self.local_rank) if (self.dist and self.backend == "nccl") else torch.device("cpu")

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce_max(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_to_root(self, array):
        """numpy array -> list of arrays on rank 0 (None elsewhere); test helper for tile reassembly."""
        if not self.dist:
            return [array]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(array, out, dst=0)
        return out

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
