"""Whole-path driver used by bench.py and smoke(): frames resident in HBM -> detector over a batch ->
per frame NMS, class mask, ReID, tracker (csrc/pipeline.cpp; reference yolo3/detect/video_detect.py:134-157)."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class Pipeline:
    def __init__(self, net, deepsort, conf_thres=0.5, nms_thres=0.4, class_mask=None, cap=512):
        self.net, self.ds, self.cap = net, deepsort, int(cap)
        # the batched path hands the tracker handle to the C pipeline and never passes through DeepSort.update: the
        # tracker-side NMS (deep_sort.py:52-57, a host-ordered reordering of the detections) is not part of it
        if getattr(deepsort, "nms_max_overlap", 1) != 1:
            raise ValueError("Pipeline: DeepSort(nms_max_overlap=%r) needs the frame-by-frame path (DeepSort.update / "
                             "VideoDetector(batch_frames=1)); the batched pipeline has no tracker-side NMS" % (deepsort.nms_max_overlap,))
        mask = np.ascontiguousarray(class_mask if class_mask is not None else [], dtype=np.int32)
        self._h = _lib.check_ptr(_lib.load().yds_pipeline_create(net._h, deepsort.extractor._h, deepsort.tracker._h,
                                                                 conf_thres, nms_thres,
                                                                 _lib.ptr(mask) if mask.size else None, int(mask.size)))

    def step(self, frames_dev, h, w, batch, next_frames_dev=None, select_next=None):
        """frames_dev: device pointer to uint8 [batch,h,w,3]; next_frames_dev (optional): the frames of the next
        call, whose detector pass is enqueued early.  Returns a list of int32 [m,6] (None when the detector found
        nothing and the tracker was not called)."""
        out = np.zeros((batch, self.cap, 6), np.int32)
        counts = np.zeros(batch, np.int32)
        if select_next is not None:
            _lib.check(_lib.load().yds_pipeline_set_next_injection(self._h, int(select_next)))
        _lib.check(_lib.load().yds_pipeline_step(self._h, frames_dev, next_frames_dev, h, w, batch, _lib.ptr(out), self.cap,
                                                 _lib.ptr(counts)))
        return [None if counts[b] < 0 else out[b, :counts[b]].copy() for b in range(batch)]

    def step_host(self, frames, next_frames=None, select_next=None):
        """frames / next_frames: uint8 [batch,h,w,3] HOST arrays (C-contiguous; views of a _lib.PinnedArray upload
        asynchronously).  The upload of next_frames overlaps this call's work; the next call must pass the same array
        object's memory as `frames`."""
        assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.flags["C_CONTIGUOUS"]
        batch, h, w, _ = frames.shape
        if next_frames is not None:
            assert next_frames.shape == frames.shape and next_frames.dtype == np.uint8 and next_frames.flags["C_CONTIGUOUS"]
        out = np.zeros((batch, self.cap, 6), np.int32)
        counts = np.zeros(batch, np.int32)
        if select_next is not None:
            _lib.check(_lib.load().yds_pipeline_set_next_injection(self._h, int(select_next)))
        _lib.check(_lib.load().yds_pipeline_step_host(self._h, _lib.ptr(frames), _lib.ptr(next_frames), h, w, batch, _lib.ptr(out),
                                                      self.cap, _lib.ptr(counts)))
        return [None if counts[b] < 0 else out[b, :counts[b]].copy() for b in range(batch)]

    def set_frame_order(self, bgr):
        """bgr=True: the frames handed to step / step_host are B, G, R as a decoder delivers them (read in place, no reversed copy)."""
        _lib.check(_lib.load().yds_pipeline_set_frame_order(self._h, 1 if bgr else 0))

    def prefetch_host(self, frames):
        """Start uploading a batch that a LATER step_host call will receive (one per step; the array must stay alive and
        unchanged until the next step_host call returns)."""
        assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.flags["C_CONTIGUOUS"]
        batch, h, w, _ = frames.shape
        _lib.check(_lib.load().yds_pipeline_prefetch_host(self._h, _lib.ptr(frames), h, w, batch))

    def set_schedule(self, min_crops=None):
        """Crops per batch from which the ReID pass is serialized with the detector passes (own stream otherwise);
        -1 = always two streams, None = the library's policy (both schedules timed on the first steady-state steps, the
        faster one kept: schedule_trial()).  Results do not depend on it."""
        _lib.check(_lib.load().yds_pipeline_set_schedule(self._h, -2 if min_crops is None else int(min_crops)))

    def last_schedule(self):
        return "serialized" if _lib.load().yds_pipeline_last_schedule(self._h) else "two-stream"

    def schedule_trial(self, uploaded=False):
        """What the pipeline's schedule trial measured for an entry (frames resident in HBM / uploaded inside the step):
        dict(decided="serialized" | "two-stream" | None while measuring, serialized_s, two_stream_s: seconds of three measured steps, the better of each schedule's two groups, wall clock between returns of step())."""
        d, a, b = C.c_int(0), C.c_double(0), C.c_double(0)
        _lib.check(_lib.load().yds_pipeline_schedule_trial(self._h, 1 if uploaded else 0, C.byref(d), C.byref(a), C.byref(b)))
        return dict(decided={1: "serialized", -1: "two-stream"}.get(d.value), serialized_s=a.value, two_stream_s=b.value)

    def stage_us(self):
        us = np.zeros(5, np.float32)
        _lib.check(_lib.load().yds_pipeline_stage_us(self._h, _lib.ptr(us)))
        return dict(zip(("resize_dev", "detector_dev", "wait_nms_host", "reid_host", "assoc_host"), us.tolist()))

    def __del__(self):
        try:
            if self._h:
                _lib.load().yds_pipeline_destroy(self._h)
                self._h = None
        except Exception:
            pass


def conv_timing(net, mode=0):
    """Per tile-variant (total_us, launches, flops, name) of the conv kernel; mode 1 resets+starts, 2 stops."""
    lib = _lib.load()
    nv = lib.yds_conv_num_variants()
    us, fl, by, at = ((C.c_double * nv)() for _ in range(4))
    n = (C.c_int64 * nv)()
    _lib.check(lib.yds_conv_timing_ex(net._h, mode, us, n, fl, by, at))
    return [dict(name=lib.yds_conv_variant_name(v).decode(), us=us[v], launches=n[v], flops=fl[v], bytes=by[v], attainable_us=at[v])
            for v in range(nv)]


def conv_clock(reset=True):
    """(GHz, sampled ms) of the in-kernel clock sampling of a -DYDS_CLOCK_PROBE build since the last reset; (0, 0) from the product
    library, whose kernels carry no sampling code (bench.py then reads the driver's sclk instead: SclkSampler)."""
    ghz, ms = C.c_double(0), C.c_double(0)
    _lib.check(_lib.load().yds_conv_clock(C.byref(ghz), C.byref(ms), 1 if reset else 0))
    return ghz.value, ms.value


class SclkSampler:
    """Samples the driver's shader clock of the bound device (sysfs /sys/bus/pci/devices/<bdf>/pp_dpm_sclk: the line marked '*') on a
    host thread while a workload runs: `with SclkSampler() as s: ...; s.ghz()`.  None when the file is absent or unreadable."""

    def __init__(self, period_s=0.01):
        import threading
        self.period, self.samples, self._stop = period_s, [], threading.Event()
        bdf = _lib.pci_bus_id()
        self.path = "/sys/bus/pci/devices/%s/pp_dpm_sclk" % bdf.lower() if bdf else None
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            with open(self.path) as f:
                for line in f:
                    if line.rstrip().endswith("*"):
                        return float(line.split(":")[1].lower().split("mhz")[0])
        except Exception:                                   # noqa: BLE001 - no sysfs in this container, another driver version ...
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v is not None:
                self.samples.append(v)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.path:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=1)

    def ghz(self):
        busy = [v for v in self.samples if v > 300]          # (drop the idle / sleep state between legs)
        return round(sum(busy) / len(busy) / 1e3, 3) if busy else None


def load_injection_sets(net, sets, logit=6.0):
    """sets: list (per step) of lists (per batch slot) of [n,9] arrays."""
    bm = net.batch_max
    rows, offsets = [], [0]
    for s in sets:
        assert len(s) == bm, "every set needs one table per batch slot"
        for r in s:
            r = np.asarray(r, np.float32).reshape(-1, 9)
            rows.append(r)
            offsets.append(offsets[-1] + r.shape[0])
    rows = np.ascontiguousarray(np.concatenate(rows, 0) if rows else np.zeros((0, 9), np.float32))
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    _lib.check(_lib.load().yds_darknet_load_injection_sets(net._h, _lib.ptr(rows), _lib.ptr(off), len(sets), logit))


def select_injection_set(net, i):
    _lib.check(_lib.load().yds_darknet_select_injection_set(net._h, int(i)))
