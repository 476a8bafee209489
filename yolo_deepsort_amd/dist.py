"""Multi-GPU plumbing of the stream-sharded run: one process per GPU, one independent video stream per rank
(tracker state is per stream, reference deep_sort/deep_sort.py:41-44 ``clone()``), weights replicated.

The exchange step - rank 0 collects every stream's int32 tracker rows once per frame batch - the timing barriers and the
max / sum reductions run on RCCL DIRECTLY through libydsort's ``yds_comm_*`` group (csrc/comm.cpp: ncclCommInitRank from
an id created by rank 0, ncclAllGather of the block {count, rows[R][6]} per frame - R starts at 64 and grows when a frame has
more rows - ncclAllReduce) - backend "nccl".
torch.distributed is only the host-side rendezvous that hands the 128-byte RCCL id to the other ranks (a gloo group over
127.0.0.1: the launcher's MASTER_ADDR / MASTER_PORT contract); no torch CUDA call is made.  Backend "gloo" keeps
everything on that host group: CPU tests, and the 2-ranks-on-one-GPU test (RCCL refuses two ranks on one device)."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

MIN_ROWS = 64                        # rows per frame the exchange block starts with; it grows in steps of 64 (never shrinks)


def rows_for(n):
    """Block capacity (rows per frame) that holds n rows: a multiple of MIN_ROWS."""
    return max(MIN_ROWS, (int(n) + MIN_ROWS - 1) // MIN_ROWS * MIN_ROWS)


def block_ints(rows):
    return 1 + rows * 6              # int32 per frame: count, rows[rows][6]


def pack_rows(outs, rows=MIN_ROWS):
    """[rows or None per frame] -> int32 [batch, 1 + rows*6].  Header: count; -1 = the detector returned None; -(2 + n) = this
    frame has n > `rows` rows and did not fit (the receivers grow the block to hold n and the exchange is repeated)."""
    blk = np.zeros((len(outs), block_ints(rows)), np.int32)
    for b, o in enumerate(outs):
        if o is None:
            blk[b, 0] = -1
            continue
        o = np.asarray(o, np.int32).reshape(-1, 6)
        if o.shape[0] > rows:
            blk[b, 0] = -(2 + o.shape[0])
            continue
        blk[b, 0] = o.shape[0]
        blk[b, 1:1 + o.size] = o.reshape(-1)
    return blk


def rows_needed(blk):
    """Largest row count any header of `blk` (any leading shape, last axis = one block) announces."""
    h = np.asarray(blk, np.int32)[..., 0]
    return int(np.where(h <= -2, -2 - h, np.maximum(h, 0)).max(initial=0))


def unpack_rows(blk):
    """int32 [batch, 1 + rows*6] -> [rows or None per frame]"""
    blk = np.asarray(blk, np.int32)
    out = []
    for row in blk.reshape(-1, blk.shape[-1]):
        n = int(row[0])
        if n <= -2:
            raise ValueError(f"exchange block of {(blk.shape[-1] - 1) // 6} rows cannot hold a frame of {-2 - n} rows")
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
        self.requested = self.backend    # what the caller asked for (bench.py refuses a silent downgrade)
        # rows per frame of the exchange block (grows, same value on every rank); YDS_EXCHANGE_ROWS: a smaller start, so that tests reach
        # the growth path with ordinary scenes (must be the same on every rank, like every launcher-provided variable)
        self.rows = int(os.environ.get("YDS_EXCHANGE_ROWS", MIN_ROWS))
        if self.world > 1:
            # libydsort (and with it the HIP runtime it is linked against) is mapped BEFORE torch brings its own ROCm libraries
            from . import _lib
            _lib.load()
            import torch.distributed as dist
            dist.init_process_group("gloo")
            self.dist = dist

    def connect(self):
        """Create the RCCL communicator on the device this process is bound to (after _lib.init()).  No-op for one rank or
        the gloo backend.  Two votes on the host group keep a one-sided failure from hanging the job: (1) a LOCAL preflight
        (librccl found, its symbols resolved, a device bound, a stream created - yds_comm_preflight, no collective) - only if
        every rank passed does any rank enter ncclCommInitRank, which is itself a collective; (2) the result of the init."""
        if self.world == 1 or self.backend != "nccl" or self.comm is not None:
            return self
        from . import _lib
        lib = _lib.load()
        import torch

        def all_ok(flag):
            ok = torch.tensor([1 if flag else 0], dtype=torch.int32)
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            return int(ok.item()) == 1
        err = None
        try:
            _lib.init()
            _lib.check(lib.yds_comm_preflight())
        except Exception as e:                                  # noqa: BLE001 - whatever went wrong locally, the others must learn of it
            err = f"rank {self.rank}: {e}"
        comm = None
        if all_ok(err is None):
            ident = (C.c_char * 128)()
            if self.rank == 0:
                try:
                    _lib.check(lib.yds_comm_unique_id(ident))
                except _lib.YdsError as e:
                    err = str(e)
            box = [bytes(ident.raw), err]
            self.dist.broadcast_object_list(box, src=0)
            if box[1] is None:
                try:
                    comm = _lib.check_ptr(lib.yds_comm_create(C.create_string_buffer(box[0], 128), self.world, self.rank))
                except _lib.YdsError as e:
                    err = str(e)
            else:
                err = box[1]
        # every rank must end up on the same transport: if RCCL could not be brought up anywhere, all of them stay on the host group
        if all_ok(comm is not None):
            self.comm = comm
        else:
            if comm:
                lib.yds_comm_destroy(comm)
            reasons = [r for r in self.gather_objects(err) if r]
            self.fallback_reason = "; ".join(reasons) or "RCCL initialisation failed"
            self.backend = "gloo"
            if self.rank == 0:
                import sys
                print(f"[yolo_deepsort_amd.dist] RCCL communicator not available ({self.fallback_reason}); exchange step falls back to gloo", file=sys.stderr)
        return self

    @property
    def transport(self):
        """What the collectives of this job really run on: "rccl", "gloo" or "none" (one rank)."""
        return "none" if self.world == 1 else ("rccl" if self.comm is not None else "gloo")

    def describe(self):
        """Machine-readable record of the exchange transport (bench.py puts it into the JSON line)."""
        from . import _lib
        lib = _lib.load()
        rec = dict(transport=self.transport, requested=self.requested, rccl_world=0, rccl_version=None, fallback_reason=self.fallback_reason,
                   exchange_block_rows=self.rows)
        if self.comm is not None:
            rec["rccl_world"] = int(lib.yds_comm_world(self.comm))
            v = int(lib.yds_comm_rccl_version())
            rec["rccl_version"] = v if v > 0 else None
        return rec

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
        while True:
            blk = pack_rows(outs, self.rows)
            allb = self.gather_array(blk)
            need = rows_needed(allb)
            if need <= self.rows:
                break
            self.rows = rows_for(need)                 # every rank sees the same headers, so every rank grows alike and repeats
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
