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
  value_other_schedule  the same K steps under the other stream schedule (config.schedule names the one `value` ran under:
                      "serialized" = the ReID pass of batch i is enqueued on the detector's stream between the first layers of pass
                      i+1 and the rest, every conv kernel has the chip to itself; "two-stream" = both passes share the CUs;
                      results are identical; the pipeline times both on its first steady-state steps and keeps the faster one -
                      config.schedule_trial carries what it measured; those steps run as set-up before the W warm-up steps)
  value_f32_math      a short run of the same workload on the exact-fp32 MFMA kernels (dtype of `value` is f16x3)
  value_real_weights  only with --weights FILE [--ckpt FILE]: the same K steps on real files, head logits NOT injected
  value_half_mode     a short run with the detector in Darknet.half() (fp16 operands, 2-byte activations; not the metric)
  value_frame_by_frame  one frame in, one result out (batch_frames = 1, nothing enqueued ahead: the reference's own loop,
                      video_detect.py:124-157); value_frame_by_frame_lookahead1 hands the next frame over one step early
  roofline            dominant conv kernel vs its MFMA bound: HIP event pairs around every conv launch on the detector's
                      stream, recorded (without host synchronisation) over a repeat of the SAME K timed steps - ReID and
                      association streams live, next pass prefetched - under the SERIALIZED schedule (a diagnostic leg since
                      round 5: every conv launch has the chip to itself there, so `achieved`, `avg_launch_us` and `frac` are the
                      kernel's own in-pipeline figures whatever schedule the product picked for `value`; tools/profile_bench.sh
                      profiles the same leg with rocprofv3: profiles/); `frac_isolated` / `achieved_isolated` come from two non-prefetched steps (conv
                      stream alone).  `all_conv_kernels` adds the per-launch attainable bound max(flops / MFMA peak,
                      bytes / 6.29 TB/s) so HBM-bound 1x1 layers are priced against the right roof.
  roofline_f32        the same record for the value_f32_math leg (exact fp32 MFMA kernels, the reference's own arithmetic)
                      against the fp32-input MFMA peak of 157.3 TFLOP/s
  cpu_baseline        the oracle pipeline (C + OpenMP / BLAS) on the host cores over a bounded sample of the same stream
N > 1: every rank runs its own stream (cfg4 = cfg3 per rank, seeds = rank); after each step the ranks all-gather their
result rows ({count, rows[256][6]} per frame, RCCL through libydsort's yds_comm_*) so rank 0 holds every stream's output.
"""

import argparse
import gc
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


def timed_steps(wl, ranks, sync, K, W, first, host_frames, lookahead=True, keep=None, own=None):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync; max over ranks.
    Every step ends with the exchange step of the multi-GPU run: the all-gather of this batch's result rows (no-op for one rank)."""
    for i in range(first, first + W):
        wl.step(i, prefetch=lookahead and i + 1 < first + W, host_frames=host_frames, prefetch2=i + 2 < first + W)   # nothing of the timed region is enqueued before the clock starts
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    n_out = 0
    for i in range(first + W, first + W + K):
        outs = wl.step(i, prefetch=lookahead and i + 1 < first + W + K, host_frames=host_frames, prefetch2=i + 2 < first + W + K)   # exactly K detector passes (and K uploads) inside the timed region
        streams = ranks.gather_rows(outs)
        n_out += sum(0 if o is None else len(o) for st in streams for o in st)
        if keep is not None:
            keep.append(streams)
    sync()
    if own is not None:
        own["dt"] = time.perf_counter() - t0        # this rank's own K steps (before the closing barrier)
    ranks.barrier()
    sync()
    return ranks.max_over_ranks(time.perf_counter() - t0), n_out


class _ClipCapture:
    """cv2.VideoCapture protocol over frames held in memory (BGR, like a decoder's output): what the demo's video file is to
    VideoDetector.detect.  read() hands out views - the decode cost of a real file is not modelled (and not the subject)."""

    def __init__(self, frames_bgr, n, fps=25.0):
        self.frames, self.n, self.fps, self.pos = frames_bgr, n, fps, 0

    def isOpened(self):
        return True

    def get(self, prop):
        _, h, w = self.frames.shape[:3]
        return {5: self.fps, 6: 0.0, 3: float(w), 4: float(h), 7: float(self.n), 1: float(self.pos)}[prop]

    def set(self, prop, value):
        self.pos = int(value)

    def read(self):
        if self.pos >= self.n:
            return False, None
        f = self.frames[self.pos % len(self.frames)]
        self.pos += 1
        return True, f

    def release(self):
        pass


def video_detector_leg(config, B, seed, n_frames, device_overlay=True):
    """The path the UNCHANGED demo takes (video_deepsort.py:13-45): VideoDetector / DeepSort through the reference's import paths
    (the shim packages yolo3/, deep_sort/), no batch_frames keyword, a file-like source, `for image, detections, actions in
    vd.detect(source)` - frames decoded (here: served from memory as BGR), staged, uploaded, detector + ReID + association,
    overlay + BGR output, every result image handed to the consumer.  Returns (frames/s, per-frame host breakdown)."""
    import tempfile
    import numpy as np
    from yolo3.detect.video_detect import VideoDetector              # the shim import paths of the drop-in contract
    from yolo_deepsort_amd import cfgs, pipeline as pl
    from yolo_deepsort_amd.workload import Workload, CONF_THRES, NMS_THRES, CLASS_MASK
    wl = Workload(config, B, seed=seed, n_distinct=4 * B)
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
    vd = VideoDetector(wl.net, f.name, thres=CONF_THRES, nms_thres=NMS_THRES, class_mask=CLASS_MASK, tracker=wl.ds, device_overlay=device_overlay)
    os.unlink(f.name)
    assert vd.batch_frames is None
    vd.AUTO_BATCH = B                                                 # (the generator's default read-ahead is the bench step of this config)

    class InjectingPipeline(pl.Pipeline):                             # bench-only head-logit injection, set i for the i-th batch of the ring
        i, sel = 0, None

        def step(self, frames_dev, h, w, batch, next_frames_dev=None, select_next=None):
            s_cur, s_next = self.i % wl.n_sets, (self.i + 1) % wl.n_sets
            if self.sel != s_cur:
                pl.select_injection_set(wl.net, s_cur)
            nxt = s_next if next_frames_dev is not None else None
            out = super().step(frames_dev, h, w, batch, next_frames_dev, select_next=nxt)
            self.sel, self.i = nxt, self.i + 1
            return out
    vd._pipe = InjectingPipeline(wl.net, wl.ds, CONF_THRES, NMS_THRES, class_mask=CLASS_MASK, cap=512)
    bgr = np.ascontiguousarray(wl.ring[..., ::-1])                    # what a decoder delivers
    warm = 24 * B                                                     # schedule trial (20 steady-state steps) + ramp, untimed
    cap = _ClipCapture(bgr, warm + n_frames)
    rows = n = 0
    t0 = None
    for image, detections, actions in vd.detect(cap, show_fps=True):
        n += 1
        if n == warm:
            t0 = time.perf_counter()
            vd.host_us.update({k: 0.0 for k in vd.host_us}, frames=0)
        elif n > warm:
            rows += 0 if detections is None else len(detections)
            assert image.shape == bgr.shape[1:]
            t_last = time.perf_counter()                              # the clock stops when the LAST result image is delivered: the generator's
    dt = t_last - t0                                                  # teardown (thread joins with 0.1 s polling hand-overs) is not throughput
    u = vd.host_us
    per = {"engine: " + k: round(u[k] / max(u["frames"], 1), 1) for k in ("wait_frames", "step", "wait_consumer")}
    per.update({"consumer: " + k: round(u[k] / max(u["frames"], 1), 1) for k in ("wait_engine", "overlay")})
    rec_schedule, rec_batch, rec_overlay = vd._pipe.last_schedule(), vd._batch_now, vd.device_overlay
    del vd, wl, cap
    gc.collect()                                                      # (this leg's workload off the device before the next leg builds its own)
    return (n - warm) / dt, dict(us_per_frame_by_thread=per, frames=n - warm, tracker_rows=rows, batch_frames=rec_batch,
                                 output_stage="device (csrc/overlay.hip)" if rec_overlay else "host (numpy LabelDrawer)",
                                 schedule=rec_schedule)


def conv_roofline(variants, peak):
    """Per-variant totals -> the dominant variant's record + the all-conv record (achieved, frac, frac_of_attainable)."""
    live = [v for v in variants if v["launches"]]
    if not live:
        return None, None
    dom = max(live, key=lambda v: v["us"])
    tot_us = sum(v["us"] for v in live)
    tot_fl = sum(v["flops"] for v in live)
    tot_by = sum(v["bytes"] for v in live)
    tot_at = sum(v["attainable_us"] for v in live)
    rec = dict(kernel=dom["name"], achieved=dom["flops"] / dom["us"] / 1e6, avg_launch_us=dom["us"] / dom["launches"], launches=int(dom["launches"]),
               flops_per_launch=dom["flops"] / dom["launches"], bytes_per_launch=dom["bytes"] / dom["launches"],
               share_of_conv_time=dom["us"] / tot_us, frac_of_attainable=dom["attainable_us"] / dom["us"])
    rec["frac"] = rec["achieved"] / peak
    allc = dict(achieved=tot_fl / tot_us / 1e6, frac=tot_fl / tot_us / 1e6 / peak, hbm_tb_s=tot_by / tot_us / 1e6,
                attainable_us=tot_at, measured_us=tot_us, frac_of_attainable=tot_at / tot_us)
    return rec, allc


def measure_roofline(wl, ranks, sync, pl, K, W, first, peak, peak_of=None):
    """Event-timed conv launches of the detector stream (yds_conv_timing_ex: a HIP event pair around every launch, resolved when the
    counters are read - no host synchronisation inside a pass).
      in-pipeline: a repeat of the K timed steps (prefetched, ReID / association streams live) - what `value` ran under;
      isolated:    two non-prefetched steps (the conv stream alone, launches back to back).
    Returns (in-pipeline variants, record of the kernel that dominates IN the pipeline, all-conv record, seconds of the repeat);
    the record carries the same kernel's isolated figures.  peak_of(kernel name) -> that kernel's own bound, if it differs."""
    pl.conv_timing(wl.net, 1)
    dt_ev, _ = timed_steps(wl, ranks, sync, K, W, first, host_frames=False)
    variants = pl.conv_timing(wl.net, 2)
    dom, allc = conv_roofline(variants, peak)
    pl.conv_timing(wl.net, 1)
    base = first + W + K
    for i in range(base, base + 2):
        wl.step(i, prefetch=False)
    iso_variants = pl.conv_timing(wl.net, 2)
    _, iso_all = conv_roofline(iso_variants, peak)
    if dom is None:
        return variants, None, None, dt_ev
    dom_peak = peak_of(dom["kernel"]) if peak_of else peak
    dom["peak"] = dom_peak
    dom["frac"] = dom["achieved"] / dom_peak
    iso = next((v for v in iso_variants if v["name"] == dom["kernel"] and v["launches"]), None)
    dom["achieved_isolated"] = None if iso is None else iso["flops"] / iso["us"] / 1e6
    dom["avg_launch_us_isolated"] = None if iso is None else iso["us"] / iso["launches"]
    dom["frac_isolated"] = None if iso is None else dom["achieved_isolated"] / dom_peak
    allc["isolated"] = iso_all
    allc["steps_in_pipeline"] = K + W
    return variants, dom, allc, dt_ev


def roofline_json(dom, allc, B, peak_note, traffic=None):
    """The `roofline` object of the JSON line (bench contract: bound / achieved / peak / unit / frac / traffic + detail)."""
    r = lambda v, n=4: None if v is None else round(v, n)
    iso = allc["isolated"]
    rec = dict(bound="mfma", kernel=dom["kernel"], achieved=r(dom["achieved"], 2), peak=r(dom["peak"], 1), unit="TFLOP/s", frac=r(dom["frac"]),
               traffic=None,
               timing="HIP event pairs around every conv launch on the detector stream, no host synchronisation inside a pass.  achieved / "
                      "avg_launch_us / frac = IN THE PIPELINE: a repeat of the K timed steps (prefetched, ReID pass and association live, under "
                      "the schedule named in `schedule`) - the conditions `value` was measured under and what a rocprofv3 kernel-trace summary "
                      "of this command shows; *_isolated = two non-prefetched steps (conv stream alone)",
               avg_launch_us=r(dom["avg_launch_us"], 2), launches=dom["launches"],
               achieved_isolated=r(dom["achieved_isolated"], 2), frac_isolated=r(dom["frac_isolated"]), avg_launch_us_isolated=r(dom["avg_launch_us_isolated"], 2),
               peak_note=peak_note, frac_of_attainable=r(dom["frac_of_attainable"]),
               flops_per_launch=dom["flops_per_launch"], algorithmic_bytes_per_launch=round(dom["bytes_per_launch"]),
               share_of_conv_time=r(dom["share_of_conv_time"]),
               all_conv_kernels=dict(achieved=r(allc["achieved"], 2), frac=r(allc["frac"]), frac_of_attainable=r(allc["frac_of_attainable"]),
                                     us_per_frame=r(allc["measured_us"] / (allc["steps_in_pipeline"] * B), 1),
                                     achieved_isolated=r(iso["achieved"], 2), frac_isolated=r(iso["frac"]),
                                     frac_of_attainable_isolated=r(iso["frac_of_attainable"]), us_per_frame_isolated=r(iso["measured_us"] / (2 * B), 1),
                                     algorithmic_hbm_tb_s_isolated=r(iso["hbm_tb_s"], 3),
                                     attainable="sum over launches of max(flops / MFMA bound, algorithmic bytes / 6.29 TB/s)"))
    if traffic:
        rec.update(traffic)
        if rec.get("traffic") and dom["avg_launch_us"]:
            rec["hbm_gb_s"] = round(rec["traffic"] / dom["avg_launch_us"] / 1e3, 1)      # PMC bytes per launch / event-timed launch duration
            rec["hbm_frac_of_8tb_s"] = round(rec["hbm_gb_s"] / 8000.0, 4)
    return rec


def committed_traffic(config, kernel, B, mode=""):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes over this command (tools/profile_bench.sh PMC=1): PMC
    counters cannot be read from inside this process."""
    cfg_key = "cfg3" if config == "cfg4" else config
    out = None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{cfg_key}{mode}.json")
        if os.path.exists(tpath) and json.load(open(tpath)).get("frames_per_step", 16) == B:
            rec = json.load(open(tpath))["kernels"].get(kernel)
            if rec:
                out = dict(traffic=round(rec["hbm_bytes_per_launch"]),
                           traffic_unit="bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/" + os.path.basename(tpath) + ")")
                break
    # matrix-pipe utilisation of the same kernel from the committed SQ_VALU_MFMA_BUSY_CYCLES pass (tools/profile_bench.sh MFMA=1)
    for rnd in ("r06", "r05"):
        mpath = os.path.join(ROOT, "profiles", f"{rnd}_mfma_busy_{cfg_key}{mode}.json")
        if os.path.exists(mpath):
            rec = json.load(open(mpath))["kernels"].get(kernel)
            if rec:
                out = dict(out or {}, mfma_busy=rec["mfma_busy"],
                           mfma_busy_note="SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the launch), rocprofv3 --pmc pass over this "
                                          "command, profiles/" + os.path.basename(mpath))
                break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x 32 frames = 2 s of the pipeline: long enough for external samplers such as rocm-smi to land inside the timed region)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None, help="frames per step (detector batch).  Default (workload.DEFAULT_BATCH): 68 for every config since round 6 - "
                                                              "the 19x19 layers of a 32-frame batch are 368 window tiles for 256 CUs (1.44 rounds, a quarter of the second "
                                                              "round idle), 768 = 3.0 rounds at 68 frames: +4..5 %% end to end, +2.3 %% on the crowd stream (profiles/r06_batch_sweep.txt); "
                                                              "32 = rounds 4-5, 16 = rounds 1-3")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 = BASELINE configs[1] (the metric's configuration); cfg4 = configs[3]: yolov4 + DeepSORT, one stream per GPU (seeds = rank)")
    ap.add_argument("--seed-base", type=int, default=0, help="stream seed of rank r = seed-base + r")
    ap.add_argument("--dump-rows", default=None, help="rank 0 saves every stream's tracker rows of the timed steps (npz: s<stream>_k<step>_f<frame>)")
    ap.add_argument("--latency-steps", type=int, default=150, help="frames of the frame-by-frame legs (0 = skip)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--schedule", default="policy", choices=["policy", "serialized", "two-stream"],
                    help="ReID pass of batch i vs detector pass of batch i+1: serialized on one stream or sharing the CUs from two (yds_pipeline_set_schedule); "
                         "policy = the library's own choice: both timed on this box at set-up, the faster one kept (pipeline.cpp Trial)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the upload-inclusive and f32-math legs")
    ap.add_argument("--half", action="store_true", help="Darknet.half(): single-term fp16 kernels (reported under dtype f16)")
    ap.add_argument("--math", default="f16x3", choices=["f16x3", "f32"], help="conv arithmetic of the WHOLE run: f16x3 (default: split-fp16 operands, exact products, "
                                                                              "fp32 accumulate) or f32 (exact fp32 MFMA kernels; profiles of the value_f32_math leg)")
    ap.add_argument("--weights", default=os.environ.get("YDS_WEIGHTS"), help="a real Darknet .weights file of the config's network (weights/yolov3.weights ...): adds the "
                                                                            "UN-INJECTED leg value_real_weights (SURVEY 8d 'Weights'): same steps, the detector sees what it sees")
    ap.add_argument("--ckpt", default=os.environ.get("YDS_CKPT"), help="a real DeepSORT ckpt.t7 for that leg (default: the synthetic state dict)")
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

    from yolo_deepsort_amd.dist import Ranks
    # nccl = RCCL over xGMI through libydsort's yds_comm_* (the real multi-GPU run).  YDS_DIST_BACKEND=gloo lets the N-rank path be
    # exercised on a box whose ranks share one GPU (tests: YDS_DEVICE=0 for every rank) - RCCL refuses two ranks on one device
    ranks = Ranks(os.environ.get("YDS_DIST_BACKEND", "nccl"))
    rank, world = ranks.rank, ranks.world

    from yolo_deepsort_amd import _lib, pipeline as pl
    from yolo_deepsort_amd.workload import Workload
    _lib.init()                    # YDS_DEVICE, else LOCAL_RANK (ordinal 0 when the launcher masks one device per rank)
    lib = _lib.load()

    from yolo_deepsort_amd.workload import DEFAULT_BATCH
    B, K, W = args.batch or DEFAULT_BATCH[args.config], args.steps, args.warmup
    if args.math == "f32":
        lib.yds_set_conv_math(0)
    wl = Workload(args.config, B, seed=ranks.stream_seed(args.seed_base), half=args.half)
    cfg = wl.cfg
    wl.to_device()
    sched_arg = {"policy": None, "serialized": 0, "two-stream": -1}[args.schedule]
    wl.pipe.set_schedule(sched_arg)

    def sync():
        _lib.check(lib.yds_device_sync())

    def settle_schedule(w, first, host_frames):
        """Set-up, like the conv autotuner's launches at plan time: the pipeline picks its stream schedule BY MEASUREMENT on the caller's
        first 20 steady-state steps (pipeline.cpp Trial: groups of 5 alternating serialized / two-stream, the faster one kept).  These steps run here,
        before the W warm-up steps, so that warm-up and timed region both run the settled schedule.  Returns (steps used, record)."""
        n = 0
        if args.schedule != "policy" or w.per_frame * w.batch < 256:     # (smaller ReID passes keep two streams without a trial)
            return 0, None
        while w.pipe.schedule_trial(host_frames)["decided"] is None and n < 32:
            w.step(first + n, prefetch=True, host_frames=host_frames, prefetch2=True)
            n += 1
        if n:                                                     # (leave nothing of the trial in flight across the clock start)
            w.step(first + n, prefetch=False, host_frames=host_frames)
            n += 1
        rec = w.pipe.schedule_trial(host_frames)
        tr = ranks.gather_objects(rec)
        return n, tr

    # ---- the metric: frames resident in HBM
    n_set, trial = settle_schedule(wl, 0, False)

    # ---- N > 1: rank 0 ALONE first (the other ranks' GPUs idle, no communicator yet, no exchange step), so that the N-rank line explains
    #      itself: efficiency_vs_rank0_single = value / (N x this rate).  It runs on a workload OF ITS OWN (same construction, same seed):
    #      the job's stream and tracker state must start untouched - every stream's rows equal a single-rank run of its seed bit for bit.
    solo = None
    if world > 1:
        if rank == 0:
            w1 = Workload(args.config, B, seed=ranks.stream_seed(args.seed_base), half=args.half)
            w1.to_device()
            w1.pipe.set_schedule(sched_arg)
            n1 = 0
            if args.schedule == "policy" and w1.per_frame * w1.batch >= 256:
                while w1.pipe.schedule_trial(False)["decided"] is None and n1 < 32:
                    w1.step(n1, prefetch=True)
                    n1 += 1
            k1 = max(3, min(K, 20))
            for i in range(n1, n1 + W):
                w1.step(i, prefetch=i + 1 < n1 + W)
            sync()
            t0 = time.perf_counter()
            for i in range(n1 + W, n1 + W + k1):
                w1.step(i, prefetch=i + 1 < n1 + W + k1)
            sync()
            dt1 = time.perf_counter() - t0
            solo = dict(value=round(k1 * B / dt1, 2), ms_per_step=round(dt1 / k1 * 1e3, 3), steps=k1, schedule=w1.pipe.last_schedule(),
                        note="rank 0 alone before the communicator formed: same workload construction and steps on a workload of its own, "
                             "no exchange step, the other GPUs idle")
            del w1
            gc.collect()                                               # (the handles sit in reference cycles: free the device buffers NOW - a crowd workload is ~80 GB)
            sync()
        ranks.barrier()            # (host group: the communicator does not exist yet)
    ranks.connect()                # RCCL communicator on the bound device (N > 1)
    # A multi-GPU run must not go green on the host transport by accident: RCCL is what the N > 1 numbers are about.
    if world > 1 and ranks.requested == "nccl" and ranks.transport != "rccl":
        if rank == 0:
            print(json.dumps({"error": "bench.py --gpus %d: the RCCL communicator could not be formed (%s); refusing to measure over gloo - "
                                       "set YDS_DIST_BACKEND=gloo to run the host transport on purpose" % (world, ranks.fallback_reason)}))
        ranks.shutdown()
        raise SystemExit(3)
    kept = [] if args.dump_rows else None
    own = {}
    dt, n_out = timed_steps(wl, ranks, sync, K, W, n_set, host_frames=False, keep=kept, own=own)     # n_out: rows of ALL streams (gathered on every rank)
    dt_own = own["dt"]
    clock_ghz, clock_ms = 0.0, 0.0
    if kept is not None and rank == 0:
        import numpy as np
        arrays = {}
        for k, streams in enumerate(kept):
            for st, frames in enumerate(streams):
                for f, o in enumerate(frames):
                    arrays[f"s{st}_k{k}_f{f}"] = np.full((1, 6), -1, np.int32) if o is None else np.asarray(o, np.int32).reshape(-1, 6)
        np.savez(args.dump_rows, **arrays)
    rank_devices = ranks.gather_objects((_lib.current_device(), _lib.pci_bus_id()))
    rank_values = ranks.gather_objects(round(K * B / dt_own, 2))        # every rank's own frames/s over its own clock around the K steps
    rank_ms = ranks.gather_objects(round(dt_own / K * 1e3, 3))          # and its own ms per step
    flops_frame = wl.flops_per_frame()
    stage = wl.pipe.stage_us()
    schedule = wl.pipe.last_schedule()
    math_name = {0: "f32", 1: "f16x3", 2: "f16"}[lib.yds_get_conv_math()] if not args.half else "f16"
    # dtype: the arithmetic type the conv path computes in.  f16x3 = both operands as two-term fp16 expansions (22 significant bits), exact
    # fp16 products, fp32 accumulation - the reference's fp32 class; f32 = v_mfma_f32_32x32x2_f32

    # ---- the same steps with the frames coming from pinned host memory (PCIe inside the timed region)
    dt_up, trial_up = None, None
    base = n_set + W + K
    if not args.no_extras:
        n_up, trial_up = settle_schedule(wl, base, True)
        base += n_up
        dt_up, _ = timed_steps(wl, ranks, sync, K, W, base, host_frames=True)
        schedule_up = wl.pipe.last_schedule()
        base += W + K

    # ---- the same K steps under the OTHER schedule (results are identical; the line carries both rates)
    other = None
    if not args.no_extras:
        wl.pipe.set_schedule(-1 if schedule == "serialized" else 0)
        dt_o, _ = timed_steps(wl, ranks, sync, K, W, base, host_frames=False)
        other = {"schedule": wl.pipe.last_schedule(), "value": round(ranks.total_frames(K, B) / dt_o, 2)}
        wl.pipe.set_schedule(sched_arg)
        base += W + K

    roofline, variants = None, None
    if not args.no_roofline:
        # (every rank runs the leg - its steps contain the exchange step - rank 0 reports)
        f16x3 = lib.yds_get_conv_math() == 1 and not args.half
        peak = PEAK_F16_MFMA_TFLOPS / 3 if f16x3 else (PEAK_F16_MFMA_TFLOPS if args.half else PEAK_F32_MFMA_TFLOPS)
        # The per-kernel record is a DIAGNOSTIC leg under the serialized schedule whatever `value` ran under (round 5): there every conv
        # launch has the chip to itself, so a launch's duration is the kernel's own; under two streams the ReID network's launches
        # share the CUs and stretch every detector launch without the chip doing less (config.schedule / schedule_trial say what the
        # product picked on this box and why).
        wl.pipe.set_schedule(0)
        # the driver's sclk sampled on a host thread beside the diagnostic leg (never beside `value`); in-kernel sampling only exists in a
        # -DYDS_CLOCK_PROBE build (tools/) - the product kernels carry none since round 6
        pl.conv_clock(reset=True)
        with pl.SclkSampler() as sclk:
            variants, dom, allc, dt_ev = measure_roofline(wl, ranks, sync, pl, K, W, base, peak)
        clock_ghz, clock_ms = pl.conv_clock(reset=False)
        clock_src = "s_memtime / s_memrealtime inside the window kernels (-DYDS_CLOCK_PROBE build)"
        if not clock_ghz:
            clock_ghz, clock_ms, clock_src = sclk.ghz() or 0.0, len(sclk.samples) * sclk.period * 1e3, "driver sclk (sysfs pp_dpm_sclk, %d samples on a host thread beside the diagnostic leg)" % len(sclk.samples)
        wl.pipe.set_schedule(sched_arg)
        if rank == 0 and dom is not None:
            note = ("dense fp16 MFMA 2500 TFLOP/s / 3 MFMAs per fp32-equivalent MAC" if f16x3
                    else ("dense fp16 MFMA" if args.half else "fp32-input MFMA, 64 FLOP/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz"))
            roofline = roofline_json(dom, allc, B, note, committed_traffic(args.config, dom["kernel"], B, "_f32" if math_name == "f32" else ""))
            roofline.update(
                frac_of_fp32_mfma_peak=round(dom["achieved"] / PEAK_F32_MFMA_TFLOPS, 4),
                # pure-MFMA loop on random fp16 data, sustained (power limited): profiles/r01_ablation_dma.txt #6
                frac_of_measured_mfma_ceiling=round(dom["achieved"] / (MEASURED_F16_MFMA_TFLOPS / 3 if f16x3 else PEAK_F32_MFMA_TFLOPS), 4),
                fps_with_conv_events=round(ranks.total_frames(K, B) / dt_ev, 2),
                schedule="serialized (diagnostic leg: every conv launch has the chip to itself; `value` ran under config.schedule = %s)" % schedule,
                # every convolution of the step (detector + ReID network) against the step's wall time: a floor of the
                # conv efficiency that charges every non-conv kernel, gap and host stall to the convolutions
                pipeline_conv_frac=round(flops_frame * K * B / dt / 1e12 / peak, 4),
                # the peak assumes 2.4 GHz; the chip is power limited under this load
                sustained_clock_ghz=round(clock_ghz, 3) if clock_ghz else None, sustained_clock_source=clock_src, nominal_clock_ghz=2.4,
                peak_at_sustained_clock=round(dom["peak"] * clock_ghz / 2.4, 1) if clock_ghz else None,
                frac_at_sustained_clock=round(dom["achieved"] / (dom["peak"] * clock_ghz / 2.4), 4) if clock_ghz else None)

    # ---- power experiment: the dominant layer alone, random operands vs operands that never toggle (same binary, same instruction
    #      stream; the chip is power limited)
    power = None
    if rank == 0 and not args.no_roofline and not args.half and math_name != "f32":
        import ctypes as C
        shape = (76, 76, 128, 256, 3, 1, 1, 0)
        rec = {}
        for kind in ("random", "zero"):
            os.environ["YDS_BENCH_DATA"] = "zero" if kind == "zero" else "rand"
            us, var = C.c_double(), C.c_int()
            _lib.check(lib.yds_conv_bench(B, *shape, 30, C.byref(us), C.byref(var)))
            fl = 2.0 * B * shape[0] * shape[1] * shape[3] * 9 * shape[2]
            rec[kind] = dict(us=round(us.value, 1), tflops=round(fl / us.value / 1e6, 1), frac=round(fl / us.value / 1e6 / (PEAK_F16_MFMA_TFLOPS / 3), 4),
                             kernel=lib.yds_conv_variant_name(var.value).decode())
        os.environ.pop("YDS_BENCH_DATA", None)
        power = dict(layer="3x3 s1 128->256 @76x76, batch %d, 30 back-to-back launches" % B, random_operands=rec["random"], zero_operands=rec["zero"],
                     note="same kernel binary and instruction stream; operands that never toggle let the chip hold a higher clock under its power limit")

    # ---- frame by frame: batch_frames = 1 (the reference's loop: one frame in, one result out)
    fbf, fbf_ahead, fbf_stage = None, None, None
    if args.latency_steps > 0 and not args.half:
        wl1 = Workload(args.config, 1, seed=ranks.stream_seed(args.seed_base), n_distinct=64)
        wl1.to_device()
        n1 = args.latency_steps
        dt1, _ = timed_steps(wl1, ranks, sync, n1, 20, 0, host_frames=False, lookahead=False)
        fbf_stage = wl1.pipe.stage_us()
        dt1a, _ = timed_steps(wl1, ranks, sync, n1, 20, n1 + 20, host_frames=False, lookahead=True)
        fbf, fbf_ahead = ranks.total_frames(n1, 1) / dt1, ranks.total_frames(n1, 1) / dt1a
        del wl1
        gc.collect()                                               # (the handles sit in reference cycles: free the device buffers NOW - a crowd workload is ~80 GB)
        sync()

    # ---- the generator: VideoDetector.detect as the unchanged demo drives it (default arguments, shim import paths, BGR result images)
    vd_fps, vd_rec, vd_host_fps, vd_host_rec = None, None, None, None
    if args.latency_steps > 0 and not args.half and world == 1 and math_name != "f32":
        try:
            vd_fps, vd_rec = video_detector_leg(args.config, B, ranks.stream_seed(args.seed_base), 40 * B, device_overlay=True)
            vd_host_fps, vd_host_rec = video_detector_leg(args.config, B, ranks.stream_seed(args.seed_base), 4 * B, device_overlay=False)
        except Exception as e:                                     # noqa: BLE001 (a side leg, one rank only: no collective inside)
            vd_rec = {"error": f"{type(e).__name__}: {e}"[:300]}
        sync()

    # ---- exact-fp32 kernels (the reference's own arithmetic), short run; the network is re-planned: tensor formats depend on the
    #      conv math.  Its roofline block is measured like the default one, against the fp32-input MFMA peak.
    f32_fps, roofline_f32 = None, None
    if not args.no_extras and not args.half and math_name != "f32":
        del wl
        gc.collect()                                               # (the handles sit in reference cycles: free the device buffers NOW - a crowd workload is ~80 GB)
        sync()
        lib.yds_set_conv_math(0)
        wl32 = Workload(args.config, B, seed=ranks.stream_seed(args.seed_base))
        wl32.to_device()
        wl32.pipe.set_schedule(sched_arg)
        n32, trial32 = settle_schedule(wl32, 0, False)
        k32 = max(3, min(K, 10))                                  # (shorter runs under-read this mode: its first steps still ramp)
        dt32, _ = timed_steps(wl32, ranks, sync, k32, 3, n32, host_frames=False)
        f32_fps = ranks.total_frames(k32, B) / dt32
        sched32 = wl32.pipe.last_schedule()
        if not args.no_roofline:
            wl32.pipe.set_schedule(0)
            _, dom32, all32, _ = measure_roofline(wl32, ranks, sync, pl, k32, 3, n32 + k32 + 3, PEAK_F32_MFMA_TFLOPS)
            if rank == 0 and dom32 is not None:
                roofline_f32 = roofline_json(dom32, all32, B, "fp32-input MFMA (v_mfma_f32_32x32x2_f32), 64 FLOP/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz",
                                             committed_traffic(args.config, dom32["kernel"], B, "_f32"))
                roofline_f32["pipeline_conv_frac"] = round(flops_frame * k32 * B / dt32 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                roofline_f32["value_f32_math"] = round(f32_fps, 2)
                roofline_f32["schedule"] = "serialized (diagnostic leg); value_f32_math ran under " + sched32
                roofline_f32["schedule_trial"] = trial32
        del wl32
        gc.collect()                                               # (the handles sit in reference cycles: free the device buffers NOW - a crowd workload is ~80 GB)
        lib.yds_set_conv_math(1)

    # ---- Darknet.half() (ImageDetector(half=True), img_detect.py:49-50): single-term fp16 operands AND 2-byte activations in the
    #      detector; fp16-class accuracy, never the metric - reported so that the half / default ratio comes from one box and one run
    half_fps, half_err = None, None
    if not args.no_extras and not args.half and math_name != "f32":
        # A side leg: its failure must not cost the line its metric.  The part that can fail per rank (building the half-mode network: memory,
        # a format check) runs first; the ranks then VOTE, and only if every rank built its workload do they enter timed_steps, whose
        # barriers and gathers are collectives - a rank skipping them alone would leave its peers blocked.
        wlh = None
        try:
            wlh = Workload(args.config, B, seed=ranks.stream_seed(args.seed_base), half=True)
            wlh.to_device()
            wlh.pipe.set_schedule(sched_arg)
        except Exception as e:                                     # noqa: BLE001
            half_err = f"{type(e).__name__}: {e}"[:300]
            wlh = None
        votes = ranks.gather_objects(half_err)
        if all(v is None for v in votes):
            nh, _ = settle_schedule(wlh, 0, False)
            kh = max(3, min(K, 20))
            dth, _ = timed_steps(wlh, ranks, sync, kh, 3, nh, host_frames=False)
            half_fps = ranks.total_frames(kh, B) / dth
        else:
            half_err = "; ".join(f"rank {r}: {v}" for r, v in enumerate(votes) if v is not None)[:300]
        del wlh
        gc.collect()                                               # (the handles sit in reference cycles: free the device buffers NOW - a crowd workload is ~80 GB)
        sync()

    # ---- the un-injected leg (SURVEY 8d "Weights"): real .weights / ckpt.t7 files, nothing written into the head tensors
    real, real_err = None, None
    if args.weights:
        wlr, dtr = None, None
        try:
            if not os.path.exists(args.weights) or (args.ckpt and not os.path.exists(args.ckpt)):
                raise FileNotFoundError(args.weights if not os.path.exists(args.weights) else args.ckpt)
            wlr = Workload(args.config, B, seed=ranks.stream_seed(args.seed_base), weights=args.weights, ckpt=args.ckpt, half=args.half)
            wlr.to_device()
            wlr.pipe.set_schedule(sched_arg)
        except Exception as e:                                     # noqa: BLE001 (per-rank failure: vote before any collective, like the half leg)
            real_err = f"{type(e).__name__}: {e}"[:300]
            wlr = None
        votes = ranks.gather_objects(real_err)
        if all(v is None for v in votes):
            nr, _ = settle_schedule(wlr, 0, False)
            try:
                dtr, rows_r = timed_steps(wlr, ranks, sync, K, W, nr, host_frames=False)
            except Exception as e:                                 # noqa: BLE001 - e.g. weights whose detections include empty crops: the reference's
                real_err = f"{type(e).__name__}: {e}"[:300]        # cv2.resize raises there too (one rank only: a multi-rank job would stop here)
                if world > 1:
                    raise
                dtr = None
        if real_err is None and dtr is not None:
            real = dict(value=round(ranks.total_frames(K, B) / dtr, 2), tracker_rows_out=rows_r, weights=os.path.basename(args.weights),
                        ckpt=os.path.basename(args.ckpt) if args.ckpt else "synthetic state dict", schedule=wlr.pipe.last_schedule(),
                        note="no head-logit injection: detections are whatever these weights find in the synthetic frames")
        elif real_err is None or any(v is not None for v in votes):
            real_err = "; ".join(f"rank {r}: {v}" for r, v in enumerate(votes) if v is not None)[:300]
        del wlr
        gc.collect()                                               # (the handles sit in reference cycles: free the device buffers NOW - a crowd workload is ~80 GB)
        sync()

    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:      # rank 0 at N = 1 only (bench contract)
        cpu = cpu_baseline("cfg3" if args.config == "cfg4" else args.config, args.cpu_frames)

    if rank == 0:
        frames_total = ranks.total_frames(K, B)
        line = {
            "metric": "end-to-end frames/sec (detect+ReID+assoc), 608x608",
            "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": math_name, "data": "synthetic",
            "config": {"workload": cfg["workload"], "frames_per_step": B, "streams": world, "frame": "1920x1080x3 u8",
                       "frames_in": "resident in HBM", "tracker_rows_out": n_out, "schedule": schedule, "parallelism": f"stream-per-gpu x{world}",
                       "rank_devices": [d for d, _ in rank_devices], "rank_pci_bus_ids": [b for _, b in rank_devices],
                       "rank_values": rank_values, "rank_ms_per_step": rank_ms,
                       **({} if solo is None else {"rank0_single": solo,
                                                   "efficiency_vs_rank0_single": round(frames_total / dt / (world * solo["value"]), 4)}),
                       "schedule_trial": None if trial is None else dict(
                           note="stream schedule picked by measurement at set-up (pipeline.cpp Trial): seconds of three measured steps (the better of two groups, wall clock "
                                "between the returns of the step call) under either schedule on this box, per rank; `schedule` is what the timed steps then ran under",
                           per_rank=trial, with_upload=trial_up, set_up_steps=n_set),
                       **ranks.describe()},
            "value_definition": "frames resident in HBM when the timed region starts (bench contract); value_with_upload is the PCIe-inclusive rate",
            "value_with_upload": None if dt_up is None else round(frames_total / dt_up, 2),
            "value_with_upload_schedule": None if dt_up is None else schedule_up,
            "value_with_upload_note": "same K steps, frames handed over as pinned host memory and uploaded inside the timed region (copy stream, three staging buffers, each batch announced two steps ahead like a decoder queue)",
            "value_other_schedule": other,
            "value_f32_math": None if f32_fps is None else round(f32_fps, 2),
            "value_half_mode": None if half_fps is None else round(half_fps, 2),
            **({"value_half_mode_error": half_err} if half_err else {}),
            "value_half_mode_note": "Darknet.half(): detector on single-term fp16 operands with 2-byte activations (the reference's ImageDetector(half=True)); fp16-class accuracy, not the metric",
            **({"value_real_weights": real} if args.weights else {}), **({"value_real_weights_error": real_err} if real_err else {}),
            "value_frame_by_frame": None if fbf is None else round(fbf, 2),
            "value_frame_by_frame_lookahead1": None if fbf_ahead is None else round(fbf_ahead, 2),
            "value_frame_by_frame_note": "batch_frames = 1: one frame in, one result out, nothing enqueued ahead (video_detect.py:124-157); "
                                         "lookahead1 = the next frame is handed over one step early (its detector pass overlaps this frame's association)",
            "stage_us_frame_by_frame": None if fbf_stage is None else {k: round(v, 1) for k, v in fbf_stage.items()},
            "value_video_detector": None if vd_fps is None else round(vd_fps, 2),
            "value_video_detector_note": "VideoDetector.detect exactly as the unchanged demo calls it (video_deepsort.py:13-45: reference import paths, default "
                                         "arguments, no batch_frames keyword) on a file-like source serving 1080p BGR frames from memory: read-ahead "
                                         "batching by default (the reference decodes 128 frames ahead, video_detect.py:86), frames uploaded inside, "
                                         "overlay + RGB->BGR + FPS text on the device, every BGR result image delivered to the consumer",
            "video_detector": vd_rec,
            "value_video_detector_host_overlay": None if vd_host_fps is None else round(vd_host_fps, 2),
            "video_detector_host_overlay": vd_host_rec,
            "exchange": None if world == 1 else "all-gather of {count, rows[256][6]} per frame per rank after every step (%s)" % (
                "RCCL via yds_comm_*" if ranks.comm is not None else ("gloo" + (f"; RCCL unavailable: {ranks.fallback_reason}" if ranks.fallback_reason else ""))),
            "stage_us_last_step": {k: round(v, 1) for k, v in stage.items()},
            "algorithmic_gflop_per_frame": round(flops_frame / 1e9, 2),
            "roofline": roofline, "roofline_f32": roofline_f32, "power_experiment": power, "conv_variants": variants, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    ranks.shutdown()


if __name__ == "__main__":
    main()
