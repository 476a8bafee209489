"""Multi-GPU plumbing of the stream-sharded run: one process per GPU, one independent video stream per rank
(tracker state is per stream, reference deep_sort/deep_sort.py:41-44 ``clone()``), weights replicated.

The exchange step - rank 0 collects every stream's int32 tracker rows once per frame batch - the timing barriers and the
max / sum reductions run on RCCL DIRECTLY through libydsort's ``yds_comm_*`` group (csrc/comm.cpp: ncclCommInitRank from
an id created by rank 0, ncclAllGather of the fixed block {count, rows[256][6]} per frame, ncclAllReduce) - backend "nccl".
torch.distributed is only the host-side rendezvous that hands the 128-byte RCCL id to the other ranks (a gloo group over
127.0.0.1: the launcher's MASTER_ADDR / MASTER_PORT contract); no torch CUDA call is made.  Backend "gloo" keeps
everything on that host group: CPU tests, and the 2-ranks-on-one-GPU test (RCCL refuses two ranks on one device)."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

MAX_ROWS = 256                       # YDS_COMM_MAX_ROWS
BLOCK = 1 + MAX_ROWS * 6             # int32 per frame: count, rows[256][6]


def pack_rows(outs):
    """[rows or None per frame] -> int32 [batch, BLOCK] (count -1 = the detector returned None)."""
    blk = np.zeros((len(outs), BLOCK), np.int32)
    for b, o in enumerate(outs):
        if o is None:
            blk[b, 0] = -1
            continue
        o = np.asarray(o, np.int32).reshape(-1, 6)
        if o.shape[0] > MAX_ROWS:
            raise ValueError(f"{o.shape[0]} tracker rows in one frame exceed the exchange block ({MAX_ROWS})")
        blk[b, 0] = o.shape[0]
        blk[b, 1:1 + o.size] = o.reshape(-1)
    return blk


def unpack_rows(blk):
    """int32 [batch, BLOCK] -> [rows or None per frame]"""
    out = []
    for row in np.asarray(blk, np.int32).reshape(-1, BLOCK):
        n = int(row[0])
        out.append(None if n < 0 else row[1:1 + n * 6].reshape(n, 6).copy())
    return out


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend or "nccl"
        self.dist = None                 # host-side group (gloo)
        self.comm = None                 # yds_comm handle (RCCL) once connect() ran
        self.fallback_reason = None      # why connect() stayed on the host group although "nccl" was asked for
        if self.world > 1:
            # libydsort (and with it the HIP runtime it is linked against) is mapped BEFORE torch brings its own ROCm libraries
            from . import _lib
            _lib.load()
            import torch.distributed as dist
            dist.init_process_group("gloo")
            self.dist = dist

    def connect(self):
        """Create the RCCL communicator on the device this process is bound to (after _lib.init()).  No-op for one rank or
        the gloo backend."""
        if self.world == 1 or self.backend != "nccl" or self.comm is not None:
            return self
        from . import _lib
        lib = _lib.load()
        _lib.init()
        import torch
        ident = (C.c_char * 128)()
        err = None
        if self.rank == 0:
            try:
                _lib.check(lib.yds_comm_unique_id(ident))
            except _lib.YdsError as e:
                err = str(e)
        box = [bytes(ident.raw), err]
        self.dist.broadcast_object_list(box, src=0)
        comm = None
        if box[1] is None:
            try:
                comm = _lib.check_ptr(lib.yds_comm_create(C.create_string_buffer(box[0], 128), self.world, self.rank))
            except _lib.YdsError as e:
                err = str(e)
        else:
            err = box[1]
        # every rank must end up on the same transport: if RCCL could not be brought up anywhere, all of them stay on the host group
        ok = torch.tensor([1 if comm else 0], dtype=torch.int32)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            self.comm = comm
        else:
            if comm:
                lib.yds_comm_destroy(comm)
            self.fallback_reason = err or "RCCL initialisation failed on another rank"
            self.backend = "gloo"
            if self.rank == 0:
                import sys
                print(f"[yolo_deepsort_amd.dist] RCCL communicator not available ({self.fallback_reason}); exchange step falls back to gloo", file=sys.stderr)
        return self

    def stream_seed(self, base=0):
        """Each rank synthesises its own stream: seed = base + rank (seeds 0-7)."""
        return base + self.rank

    def barrier(self):
        if self.comm is not None:
            from . import _lib
            _lib.check(_lib.load().yds_comm_barrier(self.comm))
        elif self.dist is not None:
            self.dist.barrier()

    def _reduce(self, value, op):
        if self.comm is not None:
            from . import _lib
            v = (C.c_double * 1)(float(value))
            _lib.check(_lib.load().yds_comm_allreduce_f64(self.comm, v, 1, op))
            return float(v[0])
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM)
        return float(t.item())

    def max_over_ranks(self, value):
        return self._reduce(value, 1)

    def sum_over_ranks(self, value):
        return self._reduce(value, 0)

    def gather_rows(self, outs):
        """The exchange step: every rank hands in the rows of its stream for one frame batch ([rows or None] per frame) and
        receives [stream 0's list, stream 1's list, ...] - what rank 0 emits for the whole job."""
        if self.world == 1:
            return [outs]
        blk = pack_rows(outs)
        allb = np.zeros((self.world,) + blk.shape, np.int32)
        if self.comm is not None:
            from . import _lib
            _lib.check(_lib.load().yds_comm_allgather(self.comm, _lib.ptr(blk), blk.nbytes, _lib.ptr(allb)))
        else:
            import torch
            parts = [torch.zeros(blk.shape, dtype=torch.int32) for _ in range(self.world)]
            self.dist.all_gather(parts, torch.from_numpy(blk))
            allb = np.stack([p.numpy() for p in parts], 0)
        return [unpack_rows(allb[r]) for r in range(self.world)]

    def gather_array(self, arr):
        """All-gather of one fixed-size numpy array per rank: [world, *arr.shape] on every rank (RCCL through yds_comm_allgather,
        or the host group)."""
        arr = np.ascontiguousarray(arr)
        if self.world == 1:
            return arr[None].copy()
        out = np.zeros((self.world,) + arr.shape, arr.dtype)
        if self.comm is not None:
            from . import _lib
            _lib.check(_lib.load().yds_comm_allgather(self.comm, _lib.ptr(arr), arr.nbytes, _lib.ptr(out)))
        else:
            import torch
            flat = torch.from_numpy(arr.reshape(-1).view(np.uint8).copy())
            parts = [torch.zeros_like(flat) for _ in range(self.world)]
            self.dist.all_gather(parts, flat)
            out = np.stack([p.numpy().view(arr.dtype).reshape(arr.shape) for p in parts], 0)
        return out

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
        if self.comm is not None:
            from . import _lib
            _lib.load().yds_comm_destroy(self.comm)
            self.comm = None
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
