#!/usr/bin/env python3
"""Timed leg of the single-stream mode (SURVEY 8e optional; yolo_deepsort_amd/single_stream.py): ONE cfg2 stream (yolov3-608 +
DeepSORT, 30 persons per frame, boxes injected as head logits like bench.py), frames detected + embedded on rank f // B % N, the
tracker on rank 0.  Run on the GPU box:

    python tools/single_stream_bench.py [--frames 192] [--per-rank 1 4] [--host]          # one rank (a world-of-one RCCL communicator)
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/single_stream_bench.py   # N GPUs

Prints one JSON line per (form, frames per rank): frames/s of the whole stream on rank 0's clock."""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import _lib, cfgs  # noqa: E402
from yolo_deepsort_amd.dist import Ranks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--per-rank", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--host", action="store_true", help="also time the host-block form (embeddings copied to the host and gathered there)")
    args = ap.parse_args()
    ranks = Ranks(os.environ.get("YDS_DIST_BACKEND", "nccl"))
    _lib.init()
    ranks.connect()
    lib = _lib.load()
    own_comm = None
    if ranks.world == 1 and ranks.comm is None:                    # one rank: a world-of-one communicator exercises the device exchange
        ident = (C.c_char * 128)()
        _lib.check(lib.yds_comm_unique_id(ident))
        own_comm = ranks.comm = _lib.check_ptr(lib.yds_comm_create(ident, 1, 0))
    from yolo_deepsort_amd.detect import ImageDetector
    from yolo_deepsort_amd.single_stream import SingleStream
    from yolo_deepsort_amd.workload import CLASS_MASK, CONF_THRES, NMS_THRES, Workload
    n_all = args.warmup + args.frames
    wl = Workload("cfg2", 1, seed=0, n_distinct=(n_all + 1) // 2 + 1)     # every rank holds the SAME stream
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
        names = f.name
    real = ImageDetector(wl.net, names, thres=CONF_THRES, nms_thres=NMS_THRES)
    frames = [wl.ring[i] for i in range(n_all)]
    index = {id(fr): i for i, fr in enumerate(frames)}

    class Det:                                                     # ImageDetector + the bench's logit injection for the frame at hand
        model = wl.net

        def detect(self, frame):
            wl._pl.select_injection_set(wl.net, index[id(frame)] % wl.n_sets)
            return real.detect(frame)

    forms = [("device", True)] + ([("host", False)] if args.host else [])
    for name, dev in forms:
        for B in args.per_rank:
            from yolo_deepsort_amd.deep_sort import DeepSort
            from yolo_deepsort_amd.workload import DS_PARAMS
            ds = DeepSort(wl.reid_sd, use_cuda=True, **DS_PARAMS)
            S = SingleStream.from_components(ranks, Det(), ds, class_mask=CLASS_MASK, frames_per_rank=B, device=dev)
            S.run(frames[:args.warmup])
            ranks.barrier()
            t0 = time.perf_counter()
            out = S.run(frames[args.warmup:])
            _lib.check(lib.yds_device_sync())
            ranks.barrier()
            dt = ranks.max_over_ranks(time.perf_counter() - t0)
            if ranks.rank == 0:
                rows = sum(len(o) for o in out if o is not None)
                print(json.dumps({"mode": "single stream on %d GPU(s)" % ranks.world, "form": name if S.on_device == dev else "host", "frames_per_rank": B,
                                  "frames": args.frames, "frames_per_s": round(args.frames / dt, 1), "tracker_rows": rows, "exchanges": S.exchanges,
                                  "block_detections": S.cap, **ranks.describe()}))
    if own_comm is not None:
        lib.yds_comm_destroy(own_comm)
        ranks.comm = None
    os.unlink(names)
    ranks.shutdown()


if __name__ == "__main__":
    main()
