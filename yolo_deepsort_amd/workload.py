"""The benchmark workloads (BASELINE.json configs[1], [2], [4]) as one construction shared by bench.py and
the parity tests at the benchmarked shape, so that what is measured is what is checked.

A workload = Darknet 608x608 (seeded synthetic weights in the real .weights layout) + DeepSORT (synthetic ckpt.t7
state dict, the demo's parameters video_deepsort.py:18-25) + a synthetic 1080p stream of scripted persons whose boxes
are injected as head logits (synthetic weights cannot see them; compute is unchanged)."""

from __future__ import annotations

import numpy as np

from . import _lib, cfgs, synth

CONFIGS = {
    "cfg2": dict(net="yolov3", persons=30, visible=None, workload="yolov3.cfg 608x608 + DeepSORT, synthetic 1080p stream, 30 persons/frame"),
    "cfg3": dict(net="yolov4", persons=30, visible=None, workload="yolov4.cfg 608x608 + DeepSORT, synthetic 1080p stream, 30 persons/frame"),
    # BASELINE configs[3]: cfg3 on every rank, one independent stream per GPU (stream seed = rank; bench.py --gpus N)
    "cfg4": dict(net="yolov4", persons=30, visible=None, workload="yolov4.cfg 608x608 + DeepSORT, independent synthetic 1080p streams (seed = rank), one per GPU, result rows all-gathered after every step (transport: `exchange`)"),
    "cfg5": dict(net="yolov4", persons=200, visible=150, workload="yolov4.cfg 608x608 + DeepSORT, crowd stream 200 tracks / 150 detections per frame"),
}
# frames per step of bench.py (profiles/r06_batch_sweep.txt).  68 for the 608 x 608 detectors: one workgroup per CU and 256-pixel tiles make a
# layer's time a step function of its tile count, and 68 x 361 = 24 548 pixels of a 19 x 19 map are 95.9 -> 96 tiles of 256: x 8 filter tiles
# = 768 = exactly 3 rounds on 256 CUs (38 x 38: 384 x 4 = 6.0 rounds, 76 x 76: 1535 x 2 = 11.99) - at 64 frames they are 2.84 / 5.64 / 11.3,
# at 32 (rounds 4-5) 1.44 / 2.83 / 5.64.  cfg5 (4800 crops per 32 frames: its ReID pass alone fills the chip) gains less: 64 frames -1.4 %,
# 34 equal, 68 +2.3 % (667 -> 682 frames/s on one box).
DEFAULT_BATCH = {"cfg2": 68, "cfg3": 68, "cfg4": 68, "cfg5": 68}
DS_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)   # video_deepsort.py:18-25
IMG = 608
CONF_THRES, NMS_THRES, CLASS_MASK = 0.5, 0.4, [0, 2, 4]


class Workload:
    def __init__(self, config, batch, seed=0, n_distinct=None, half=False, long_occlude=None, pingpong=True, weights=None, ckpt=None):
        """long_occlude / pingpong=False: the long-stream parity form (tests/test_gpu_long_stream.py) - `n_distinct` frames of
        the stream with synth.PersonScene's long occlusion windows, played once front to back.
        weights / ckpt (SURVEY 8d "Weights"): paths of a real Darknet .weights file / a ckpt.t7; with `weights` the head logits are
        NOT injected - the detector sees what it sees (bench.py --weights/--ckpt, the un-injected leg)."""
        from .deep_sort import DeepSort, Extractor
        from .models import Darknet
        from . import pipeline as pl
        self.cfg = CONFIGS[config]
        self.batch = B = int(batch)
        self.cfg_text = cfgs.cfg_text(self.cfg["net"], IMG, IMG)
        self.injected = weights is None
        self.net = Darknet(None, img_size=(IMG, IMG), batch_max=B, cfg_text=self.cfg_text)
        if half:
            self.net.half()
        if weights is None:
            self.blob = synth.darknet_weights_blob(self.cfg_text, seed=0)
            self.net.load_darknet_weights(None, blob=self.blob)
        else:
            self.net.load_darknet_weights(weights)
        self.reid_sd = synth.reid_state_dict(0) if ckpt is None else ckpt                  # (Extractor takes a state dict or a path)
        per_frame = self.cfg["visible"] or self.cfg["persons"]
        self.per_frame = per_frame
        self.ds = DeepSort(Extractor(self.reid_sd, max_crops=B * (per_frame + 8)), use_cuda=True, **DS_PARAMS)
        # ping-pong ring of frames so that the stream stays continuous when it wraps
        n_distinct = n_distinct or max(4 * B, 32)
        self.scene = synth.PersonScene(self.cfg["persons"], seed=seed, n_visible=self.cfg["visible"], long_occlude=long_occlude)
        frames = np.stack([self.scene.frame(t) for t in range(n_distinct)], 0)
        self.order = list(range(n_distinct)) + (list(range(n_distinct - 1, -1, -1)) if pingpong else [])
        self.n_sets = len(self.order) // B
        if self.injected:
            heads = self.net.yolo_heads()
            self.inj = [synth.head_injection(self.scene.boxes(t)[1], (self.scene.H, self.scene.W), (IMG, IMG), heads, cls=0)
                        for t in range(n_distinct)]
            pl.load_injection_sets(self.net, [[self.inj[self.order[s * B + b]] for b in range(B)] for s in range(self.n_sets)])
        self.H, self.W = frames.shape[1:3]
        self.frame_bytes = self.H * self.W * 3
        self._pinned = _lib.PinnedArray((len(self.order),) + frames.shape[1:], np.uint8)     # host copy of the stream, in play order,
        np.take(frames, self.order, axis=0, out=self._pinned.array)                          # in pinned memory (where a decoder writes)
        self.ring = self._pinned.array
        self.pipe = pl.Pipeline(self.net, self.ds, conf_thres=CONF_THRES, nms_thres=NMS_THRES, class_mask=CLASS_MASK, cap=512)
        self._pl = pl
        self._sel = None
        self.dev = None

    def to_device(self):
        """Keep the whole stream resident in HBM (the 'hbm' mode of the bench)."""
        if self.dev is None:
            self.dev = _lib.DeviceBuffer.from_array(self.ring)
        return self.dev

    def step(self, i, prefetch=True, host_frames=False, prefetch2=False):
        """Step i: one pass over `batch` consecutive frames.  The injection set of the detector pass that is enqueued
        inside this call must be selected before it: step i's own pass when nothing was prefetched, else step i+1's.
        host_frames: the frames are handed over as HOST memory (pinned ring) and uploaded inside the step; prefetch2: step i+2
        will follow, its frames are announced to the pipeline now (yds_pipeline_prefetch_host)."""
        B = self.batch
        s, s_next = i % self.n_sets, (i + 1) % self.n_sets
        if self.injected and self._sel != s:
            self._pl.select_injection_set(self.net, s)
            self._sel = s
        if host_frames:
            if prefetch and prefetch2:             # the batch after next starts its upload now (a decoder two batches ahead)
                s2 = (i + 2) % self.n_sets
                self.pipe.prefetch_host(self.ring[s2 * B:(s2 + 1) * B])
            cur = self.ring[s * B:(s + 1) * B]
            nxt = self.ring[s_next * B:(s_next + 1) * B] if prefetch else None
            out = self.pipe.step_host(cur, nxt, select_next=(s_next if prefetch and self.injected else None))
        else:
            dev = self.to_device()
            nxt = dev.offset(s_next * B * self.frame_bytes) if prefetch else None
            out = self.pipe.step(dev.offset(s * B * self.frame_bytes), self.H, self.W, B, nxt, select_next=(s_next if prefetch and self.injected else None))
        if prefetch:
            self._sel = s_next
        return out

    def flops_per_frame(self):
        return self.net.conv_flops() + self.per_frame * 2242904064
