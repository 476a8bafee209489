#!/usr/bin/env python3
"""Headline benchmark: end-to-end frames/sec (detect + ReID + association), 608x608 model input.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: one rank per GPU under torch.distributed.run;
`python bench.py --gpus N` on its own launches the N ranks itself).
Every rank runs its own independent synthetic 1080p stream (weak scaling, no data-path collective: tracker state is per
stream).  A step = one pass of the hot path over one batch of `--batch` consecutive frames: stretch-resize
+ /255, Darknet, decode, NMS, class mask, crop + ReID CNN, Kalman / cost / Hungarian association, int32 outputs.
Rank 0 prints ONE JSON line:
  value               frames/s with the frames already resident in HBM when the timed region starts (the contract's metric)
  value_with_upload   the same K steps with the frames handed over as pinned HOST memory and uploaded inside the step
                      (yds_pipeline_step_host: copy stream, double buffered) - the PCIe-inclusive rate
  value_f32_math      a short run of the same workload on the exact-fp32 MFMA kernels (dtype of `value` is f16x3)
  roofline            dominant conv kernel vs its MFMA bound, HIP-event timed on the detector's stream
  cpu_baseline        the oracle pipeline (C + OpenMP / BLAS) on the host cores over a bounded sample of the same stream
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MEASURED_F16_MFMA_TFLOPS = 1824.0      # tools/probes/mfma_probe, random operands, 1.3 s launch at the power limit
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32-in MFMA = 64 FLOP/clk/SIMD
PEAK_F16_MFMA_TFLOPS = 2500.0         # dense fp16/bf16 MFMA (not the 2:1-sparse headline)
REFERENCE_8CORE_FPS = "1.4-1.5"       # the real reference (PyTorch CPU) on 8 cores, cfg2-like load


def cpu_baseline(config, n_frames):
    """oracle/cpu_baseline.py in its own process (thread pools configured before any library loads)."""
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--config", config, "--frames", str(n_frames)],
                         cwd=ROOT, capture_output=True, text=True, timeout=1200)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return dict(value=None, unit="frames/s", cores=os.cpu_count(), kind="port", sample="failed: " + out.stderr[-300:])
    rec = json.loads(lines[-1])
    rec["reference_pytorch_cpu_8_cores_fps"] = REFERENCE_8CORE_FPS + " (BASELINE.md, measured in the build container, not on this host)"
    return rec


def timed_steps(wl, ranks, sync, K, W, first, host_frames):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync; max over ranks."""
    for i in range(first, first + W):
        wl.step(i, prefetch=i + 1 < first + W, host_frames=host_frames)   # nothing of the timed region is enqueued before the clock starts
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    n_out = 0
    for i in range(first + W, first + W + K):
        outs = wl.step(i, prefetch=i + 1 < first + W + K, host_frames=host_frames)   # exactly K detector passes inside the timed region
        n_out += sum(0 if o is None else len(o) for o in outs)
    sync()
    ranks.barrier()
    sync()
    return ranks.max_over_ranks(time.perf_counter() - t0), n_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="frames per step (detector batch)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the upload-inclusive and f32-math legs")
    ap.add_argument("--half", action="store_true", help="Darknet.half(): single-term fp16 kernels (reported under dtype f16)")
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
    # nccl = RCCL over xGMI (the real multi-GPU run).  YDS_DIST_BACKEND=gloo lets the N-rank path be exercised on a box whose
    # ranks share one GPU (tests: YDS_DEVICE=0 for every rank) - RCCL refuses two ranks on one device
    ranks = Ranks(os.environ.get("YDS_DIST_BACKEND", "nccl"))
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world

    from yolo_deepsort_amd import _lib, pipeline as pl
    from yolo_deepsort_amd.workload import Workload
    _lib.init(None if os.environ.get("YDS_DEVICE") else local_rank)
    lib = _lib.load()

    B, K, W = args.batch, args.steps, args.warmup
    wl = Workload(args.config, B, seed=ranks.stream_seed(), half=args.half)
    cfg = wl.cfg
    wl.to_device()

    def sync():
        _lib.check(lib.yds_device_sync())
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    # ---- the metric: frames resident in HBM
    dt, n_out = timed_steps(wl, ranks, sync, K, W, 0, host_frames=False)
    n_out = int(ranks.sum_over_ranks(n_out))
    rank_devices = ranks.gather_objects((_lib.current_device(), _lib.pci_bus_id()))
    stage = wl.pipe.stage_us()
    math_name = {0: "f32", 1: "f16x3", 2: "f16"}[lib.yds_get_conv_math()] if not args.half else "f16"

    # ---- the same steps with the frames coming from pinned host memory (PCIe inside the timed region)
    dt_up = None
    if not args.no_extras:
        dt_up, _ = timed_steps(wl, ranks, sync, K, W, W + K, host_frames=True)

    roofline, variants = None, None
    if rank == 0 and not args.no_roofline:
        # HIP events recorded around every conv launch on the detector's stream, over extra steps of the
        # same workload (kept out of the throughput region because each pair forces a host sync)
        pl.conv_timing(wl.net, 1)
        base = 2 * (W + K)
        for i in range(base, base + 2):
            wl.step(i, prefetch=False)
        variants = pl.conv_timing(wl.net, 2)
        dom = max(variants, key=lambda v: v["us"])
        if dom["launches"]:
            avg_us = dom["us"] / dom["launches"]
            achieved = dom["flops"] / dom["us"] / 1e6        # TFLOP/s of algorithmic (fp32-equivalent) conv work
            # f16x3 spends 3 fp16 MFMAs per multiply-accumulate, so its matrix-pipe bound is the dense fp16 peak / 3
            f16x3 = "f16x3" in dom["name"] and not args.half
            peak = PEAK_F16_MFMA_TFLOPS / 3 if f16x3 else (PEAK_F16_MFMA_TFLOPS if args.half else PEAK_F32_MFMA_TFLOPS)
            tot_us = sum(v["us"] for v in variants)
            tot_fl = sum(v["flops"] for v in variants)
            roofline = dict(bound="mfma", kernel=dom["name"], achieved=round(achieved, 2), peak=round(peak, 1),
                            unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=None,
                            peak_note=("dense fp16 MFMA 2500 TFLOP/s / 3 MFMAs per fp32-equivalent MAC" if f16x3
                                       else ("dense fp16 MFMA" if args.half else "fp32-input MFMA, 64 FLOP/clk/SIMD")),
                            frac_of_fp32_mfma_peak=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                            # pure-MFMA loop on random fp16 data, sustained (power limited): profiles/r01_ablation_dma.txt #6
                            frac_of_measured_mfma_ceiling=round(achieved / (MEASURED_F16_MFMA_TFLOPS / 3 if f16x3 else PEAK_F32_MFMA_TFLOPS), 4),
                            avg_launch_us=round(avg_us, 2), launches=int(dom["launches"]),
                            flops_per_launch=dom["flops"] / dom["launches"],
                            share_of_conv_time=round(dom["us"] / tot_us, 4),
                            all_conv_kernels=dict(achieved=round(tot_fl / tot_us / 1e6, 2), frac=round(tot_fl / tot_us / 1e6 / peak, 4),
                                                  us_per_frame=round(tot_us / (2 * B), 1)))
            # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from
            # the committed rocprofv3 --pmc passes over this same command (tools/profile_bench.sh, PMC=1)
            for rnd in ("r02", "r01"):
                tpath = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{args.config}.json")
                if os.path.exists(tpath) and B == 16:
                    rec = json.load(open(tpath))["kernels"].get(dom["name"])
                    if rec:
                        roofline["traffic"] = round(rec["hbm_bytes_per_launch"])
                        roofline["traffic_unit"] = "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/" + os.path.basename(tpath) + ")"
                        break

    # ---- exact-fp32 kernels, short run (the network is re-planned: tensor formats depend on the conv math)
    f32_fps = None
    if not args.no_extras and not args.half:
        del wl
        sync()
        lib.yds_set_conv_math(0)
        wl32 = Workload(args.config, B, seed=ranks.stream_seed())
        wl32.to_device()
        k32 = max(3, min(K, 6))
        dt32, _ = timed_steps(wl32, ranks, sync, k32, 2, 0, host_frames=False)
        f32_fps = ranks.total_frames(k32, B) / dt32
        del wl32
        lib.yds_set_conv_math(1)

    cpu = None
    if rank == 0 and args.cpu_frames > 0:
        cpu = cpu_baseline(args.config, args.cpu_frames)

    if rank == 0:
        frames_total = ranks.total_frames(K, B)
        per_frame = cfg["visible"] or cfg["persons"]
        flops_frame = None
        line = {
            "metric": "end-to-end frames/sec (detect+ReID+assoc), 608x608",
            "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": math_name, "data": "synthetic",
            "config": {"workload": cfg["workload"], "frames_per_step": B, "streams": world, "frame": "1920x1080x3 u8",
                       "frames_in": "resident in HBM", "tracker_rows_out": n_out, "parallelism": f"stream-per-gpu x{world}",
                       "rank_devices": [d for d, _ in rank_devices], "rank_pci_bus_ids": [b for _, b in rank_devices]},
            "value_with_upload": None if dt_up is None else round(frames_total / dt_up, 2),
            "value_with_upload_note": "same K steps, frames handed over as pinned host memory and uploaded inside the step (copy stream, double buffered)",
            "value_f32_math": None if f32_fps is None else round(f32_fps, 2),
            "stage_us_last_step": {k: round(v, 1) for k, v in stage.items()},
            "algorithmic_gflop_per_frame": round((140.692 if cfg["net"] == "yolov3" else 128.389) + per_frame * 2.2429, 2),
            "roofline": roofline, "conv_variants": variants, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    ranks.shutdown()


if __name__ == "__main__":
    main()
