"""Multi-GPU plumbing for the throughput run: one process per GPU, one independent video stream per
rank (tracker state is per stream, reference deep_sort/deep_sort.py:41-44 ``clone()``), so the data path
has NO collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU
tests) is used only for the rendezvous, the timing barriers and the max-over-ranks reduction."""

from __future__ import annotations

import os


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch
            import torch.distributed as dist
            backend = backend or "nccl"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)
            self.dist = dist
            self.backend = backend

    def stream_seed(self, base=0):
        """Each rank synthesises its own stream: seed = base + rank (seeds 0-7)."""
        return base + self.rank

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (small host objects: device ids, per-rank rates)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def total_frames(self, steps, frames_per_step):
        """Whole-job work: every rank processes steps x frames_per_step frames (weak scaling)."""
        return steps * frames_per_step * self.world

    def shutdown(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
