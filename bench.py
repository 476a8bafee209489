#!/usr/bin/env python3
"""Headline benchmark: end-to-end frames/sec (detect + ReID + association), 608x608 model input.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1 via torch.distributed.run)
One rank per GPU; every rank runs its own independent synthetic 1080p stream (weak scaling, no
data-path collective: tracker state is per stream).  A step = one pass of the hot path over
one batch of `--batch` consecutive frames that are already resident in HBM: stretch-resize + /255,
Darknet, decode, NMS, class mask, crop + ReID CNN, Kalman / cost / Hungarian association, int32 outputs.
Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant conv kernel vs the fp32 MFMA peak) and
`cpu_baseline` (the numpy/C oracle on the host cores over a bounded sample of the same stream).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1], [2], [4]
    "cfg2": dict(net="yolov3", persons=30, visible=None, workload="yolov3.cfg 608x608 + DeepSORT, synthetic 1080p stream, 30 persons/frame"),
    "cfg3": dict(net="yolov4", persons=30, visible=None, workload="yolov4.cfg 608x608 + DeepSORT, synthetic 1080p stream, 30 persons/frame"),
    "cfg5": dict(net="yolov4", persons=200, visible=150, workload="yolov4.cfg 608x608 + DeepSORT, crowd stream 200 tracks / 150 detections per frame"),
}
DS_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)   # video_deepsort.py:18-25
MEASURED_F16_MFMA_TFLOPS = 1824.0      # tools/probes/mfma_probe, random operands, 1.3 s launch at the power limit
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32-in MFMA = 64 FLOP/clk/SIMD
PEAK_F16_MFMA_TFLOPS = 2500.0         # dense fp16/bf16 MFMA (not the 2:1-sparse headline)


def build_stream(cfg, seed, n_frames, img_size, net):
    from yolo_deepsort_amd import synth
    scene = synth.PersonScene(cfg["persons"], seed=seed, n_visible=cfg["visible"])
    frames = np.stack([scene.frame(t) for t in range(n_frames)], 0)
    heads = net.yolo_heads()
    inj = [synth.head_injection(scene.boxes(t)[1], (scene.H, scene.W), img_size, heads, cls=0) for t in range(n_frames)]
    return scene, frames, inj


def cpu_baseline(cfg, img_size, frames, inj_rows, blob, reid_sd, n_frames):
    """Oracle pipeline on the host cores over `n_frames` frames of the same stream (reported, not the target)."""
    from oracle.darknet import DarknetOracle
    from oracle.pipeline import run_stream
    from yolo_deepsort_amd import cfgs
    net = DarknetOracle(cfgs.cfg_text(cfg["net"], img_size, img_size), img_size, is_text=True)
    net.load_weights_array(np.frombuffer(blob, dtype=np.float32, offset=20))
    t0 = time.perf_counter()
    run_stream(net, reid_sd, DS_PARAMS, frames[:n_frames], inj_rows[:n_frames])
    dt = time.perf_counter() - t0
    return n_frames / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="frames per step (detector batch)")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own launches the N ranks itself (one process per GPU, rendezvous on 127.0.0.1);
    # under torch.distributed.run (the driver's launch line) WORLD_SIZE is already set and this is one of the ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}; launch one rank per GPU")

    import torch
    from yolo_deepsort_amd.dist import Ranks
    ranks = Ranks("nccl")
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world

    from yolo_deepsort_amd import _lib, cfgs, synth
    from yolo_deepsort_amd.models import Darknet
    from yolo_deepsort_amd.deep_sort import DeepSort
    from yolo_deepsort_amd import pipeline as pl
    _lib.init(local_rank)

    cfg = CONFIGS[args.config]
    S, B, K, W = 608, args.batch, args.steps, args.warmup
    cfg_text = cfgs.cfg_text(cfg["net"], S, S)
    blob = synth.darknet_weights_blob(cfg_text, seed=0)
    net = Darknet(None, img_size=(S, S), batch_max=B, cfg_text=cfg_text)
    net.load_darknet_weights(None, blob=blob)
    reid_sd = synth.reid_state_dict(0)
    from yolo_deepsort_amd.deep_sort import Extractor
    per_frame = cfg["visible"] or cfg["persons"]
    ds = DeepSort(Extractor(reid_sd, max_crops=B * (per_frame + 8)), use_cuda=True, **DS_PARAMS)

    # ping-pong ring of frames so that the stream stays continuous when it wraps
    n_distinct = max(4 * B, 32)
    scene, frames, inj = build_stream(cfg, ranks.stream_seed(), n_distinct, (S, S), net)
    order = list(range(n_distinct)) + list(range(n_distinct - 1, -1, -1))
    n_sets = len(order) // B
    pl.load_injection_sets(net, [[inj[order[s * B + b]] for b in range(B)] for s in range(n_sets)])
    H, Wf = frames.shape[1:3]
    ring = np.ascontiguousarray(frames[order])
    dev = _lib.DeviceBuffer.from_array(ring)
    frame_bytes = H * Wf * 3
    pipe = pl.Pipeline(net, ds, conf_thres=0.5, nms_thres=0.4, class_mask=[0, 2, 4], cap=512)

    state = {"sel": None}

    def run_step(i, prefetch=True):
        """Step i.  The injection set of the detector pass that is enqueued inside this call must be selected
        before it: that is step i's own pass when nothing was prefetched, else step i+1's."""
        s, s_next = i % n_sets, (i + 1) % n_sets
        if state["sel"] != s:
            pl.select_injection_set(net, s)
            state["sel"] = s
        nxt = None
        if prefetch:
            nxt = dev.offset(s_next * B * frame_bytes)
        out = pipe.step(dev.offset(s * B * frame_bytes), H, Wf, B, nxt, select_next=(s_next if prefetch else None))
        if prefetch:
            state["sel"] = s_next
        return out

    def sync():
        _lib.check(_lib.load().yds_device_sync())
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for i in range(W):
        run_step(i, prefetch=i + 1 < W)       # nothing of the timed region is enqueued before the clock starts
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    n_out = 0
    for i in range(W, W + K):
        outs = run_step(i, prefetch=i + 1 < W + K)   # exactly K detector passes inside the timed region
        n_out += sum(0 if o is None else len(o) for o in outs)
    sync()
    ranks.barrier()
    sync()
    dt = ranks.max_over_ranks(time.perf_counter() - t0)
    n_out = int(ranks.sum_over_ranks(n_out))
    rank_devices = ranks.gather_objects((_lib.current_device(), _lib.pci_bus_id()))
    stage = pipe.stage_us()

    roofline, variants = None, None
    if rank == 0 and not args.no_roofline:
        # HIP events recorded around every conv launch on the detector's stream, over extra steps of the
        # same workload (kept out of the throughput region because each pair forces a host sync)
        pl.conv_timing(net, 1)
        for i in range(W + K, W + K + 2):
            run_step(i, prefetch=False)
        variants = pl.conv_timing(net, 2)
        dom = max(variants, key=lambda v: v["us"])
        if dom["launches"]:
            avg_us = dom["us"] / dom["launches"]
            achieved = dom["flops"] / dom["us"] / 1e6        # TFLOP/s of algorithmic (fp32-equivalent) conv work
            # f16x3 spends 3 fp16 MFMAs per multiply-accumulate, so its matrix-pipe bound is the dense fp16 peak / 3
            f16 = "f16x3" in dom["name"]
            peak = PEAK_F16_MFMA_TFLOPS / 3 if f16 else PEAK_F32_MFMA_TFLOPS
            roofline = dict(bound="mfma", kernel=dom["name"], achieved=round(achieved, 2), peak=round(peak, 1),
                            unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=None,
                            peak_note=("dense fp16 MFMA 2500 TFLOP/s / 3 MFMAs per fp32-equivalent MAC" if f16
                                       else "fp32-input MFMA, 64 FLOP/clk/SIMD"),
                            frac_of_fp32_mfma_peak=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                            # pure-MFMA loop on random fp16 data, sustained (power limited): profiles/r01_ablation_dma.txt #6
                            frac_of_measured_mfma_ceiling=round(achieved / (MEASURED_F16_MFMA_TFLOPS / 3 if f16 else PEAK_F32_MFMA_TFLOPS), 4),
                            avg_launch_us=round(avg_us, 2), launches=int(dom["launches"]),
                            flops_per_launch=dom["flops"] / dom["launches"])
            # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from
            # the committed rocprofv3 --pmc passes over this same command (tools/profile_bench.sh, PMC=1)
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r01_traffic_{args.config}.json")
            if os.path.exists(tpath) and B == 16:
                rec = json.load(open(tpath))["kernels"].get(dom["name"])
                if rec:
                    roofline["traffic"] = round(rec["hbm_bytes_per_launch"])
                    roofline["traffic_unit"] = "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/" + os.path.basename(tpath) + ")"

    cpu = None
    if rank == 0 and args.cpu_frames > 0:
        fps_cpu, secs = cpu_baseline(cfg, S, ring, [inj[o] for o in order], blob, reid_sd, args.cpu_frames)
        cpu = dict(value=round(fps_cpu, 4), unit="frames/s", cores=os.cpu_count(), kind="port",
                   sample=f"first {args.cpu_frames} frames of the same stream through oracle/ (numpy+BLAS, {secs:.1f} s)")

    if rank == 0:
        frames_total = ranks.total_frames(K, B)
        flops_frame = net.conv_flops() + 30 * 2242904064 if cfg["visible"] is None else net.conv_flops() + 150 * 2242904064
        line = {
            "metric": "end-to-end frames/sec (detect+ReID+assoc), 608x608",
            "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3" if _lib.load().yds_get_conv_math() == 1 else "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"], "frames_per_step": B, "streams": world, "frame": "1920x1080x3 u8",
                       "tracker_rows_out": n_out, "parallelism": f"stream-per-gpu x{world}",
                       "rank_devices": [d for d, _ in rank_devices], "rank_pci_bus_ids": [b for _, b in rank_devices]},
            "stage_us_last_step": {k: round(v, 1) for k, v in stage.items()},
            "algorithmic_gflop_per_frame": round(flops_frame / 1e9, 2),
            "roofline": roofline, "conv_variants": variants, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    ranks.shutdown()


if __name__ == "__main__":
    main()
