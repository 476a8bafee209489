"""Drop-in ``ImageDetector`` / ``VideoDetector`` (reference yolo3/detect/img_detect.py:36-153,
yolo3/detect/video_detect.py:39-208).  Only the per-frame hot glue is accelerated; video decode,
drawing and display stay host side and are optional (cv2 / imutils are not required to import this)."""

from __future__ import annotations

import logging
import time
from functools import reduce

import numpy as np

from .loaders import load_classes


def p1p2Toxywh(x):
    """model_build.py:326-332"""
    y = np.empty_like(x)
    y[..., 0] = x[..., 0]
    y[..., 1] = x[..., 1]
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def _as_tensor(a):
    try:
        import torch
        return torch.from_numpy(a)
    except ImportError:
        return a


class ImageDetector:
    def __init__(self, model, class_path, thickness=2, thres=0.5, nms_thres=0.4, win_size=None, overlap=0.15,
                 half=False):
        self.model = model
        self.model.eval()
        self.device = next(self.model.parameters()).device
        if half:
            self.model.half()
        self.classes = load_classes(class_path)
        self.num_classes = len(self.classes)
        self.thickness = thickness
        self.thres = thres
        self.nms_thres = nms_thres
        self.half = half
        self.win_size = win_size
        self.overlap = overlap

    def detect(self, img):
        """img: RGB uint8 [H,W,3] -> Tensor[n,6] (x1,y1,x2,y2,conf,cls) in frame pixels, or None."""
        h, w, _ = img.shape
        prev_time = time.time()
        if self.win_size is not None:
            win_width, win_height = self.win_size
            if not (w < win_width and h < win_height):
                # img_detect.py:97-151: windows on a win_size grid, each extended by the overlap and clipped to the frame
                overlap_x, overlap_y = int(win_width * self.overlap), int(win_height * self.overlap)
                tiles = [(x, y, min(y + win_height + overlap_y, h) - y, min(x + win_width + overlap_x, w) - x)
                         for x in range(0, w, win_width) for y in range(0, h, win_height)]
                det = self.model.detect_tiled(img, tiles, self.thres, self.nms_thres)
                logging.info("\t Inference time: %.6f s" % (time.time() - prev_time))
                return _as_tensor(det) if det.shape[0] else None
        self.model.forward_u8(img, want_output=False)
        det = self.model.nms(0, self.thres, self.nms_thres, frame_hw=(h, w))
        logging.info("\t Inference time: %.6f s" % (time.time() - prev_time))
        if det.shape[0] == 0:
            return None
        return _as_tensor(det)


class _NullDrawer:
    """Stand-in for LabelDrawer (host-side rendering is out of scope)."""

    def draw_labels(self, frame, detections, only_rect=False):
        return frame, None, None

    draw_labels_by_trackers = draw_labels


class VideoDetector:
    def __init__(self, model, class_path, thickness=2, font_path=None, font_size=10, thres=0.7, nms_thres=0.4,
                 skip_frames=-1, fourcc="XVID", class_mask=None, win_size=None, overlap=0.15, tracker=None,
                 action_id=None, half=False, batch_frames=1):
        # batch_frames (not in the reference): with a tracker, read that many frames ahead and run them through the
        # batched device pipeline (csrc/pipeline.cpp) - same results per frame, yielded in order, ~10x the frame rate of
        # the frame-by-frame path; 1 keeps the reference's latency (one frame in, one result out)
        self.batch_frames = max(1, int(batch_frames))
        self._pipe = None
        self.thickness = thickness
        self.skip_frames = skip_frames
        self.class_names = load_classes(class_path)
        self.fourcc = fourcc
        self.class_mask = class_mask
        self.tracker = tracker
        self.action_id = action_id
        self.label_drawer = _NullDrawer()
        self.image_detector = ImageDetector(model, class_path, thickness=thickness, thres=thres, nms_thres=nms_thres,
                                            win_size=win_size, overlap=overlap, half=half)

    def _frames(self, video_path):
        if hasattr(video_path, "__iter__") and not isinstance(video_path, (str, bytes)):
            for f in video_path:           # already-decoded RGB frames
                yield f
            return
        try:
            import cv2
        except ImportError:
            raise IOError("Couldn't open webcam or video")
        cap = cv2.VideoCapture(video_path)
        if not cap.isOpened():
            raise IOError("Couldn't open webcam or video")
        while True:
            ok, frame = cap.read()
            if not ok:
                return
            yield frame[:, :, ::-1]        # BGR -> RGB like video_detect.py:33-36

    def process(self, frame):
        """The hot glue of video_detect.py:134-157 for one frame: returns hold_detections."""
        detections = self.image_detector.detect(frame)
        if detections is not None and self.tracker is not None:
            det = detections.numpy() if hasattr(detections, "numpy") else detections
            boxs = p1p2Toxywh(det[:, :4])
            class_ids = det[:, -1]
            confidences = det[:, 4]
            if self.class_mask is not None:
                mask = reduce(lambda a, b: a | b, [class_ids == m for m in self.class_mask])
                boxs, confidences, class_ids = boxs[mask], confidences[mask], class_ids[mask]
            if hasattr(self.tracker, "extractor"):
                self.tracker.frame_source = self.image_detector.model       # lets update() crop from the detector's device copy
            detections = self.tracker.update(boxs.astype(np.float32), confidences, frame, class_ids)
        return detections

    def _processed_batches(self, video_path):
        """Groups of consecutive frames holding up to batch_frames frames that pass the skip_frames gate."""
        group, n_proc, frames = [], 0, 0
        for frame in self._frames(video_path):
            if frame is None:
                break
            proc = frames % self.skip_frames == 0
            if proc:
                frames = 0
            frames += 1
            group.append((frame, proc))
            n_proc += proc
            if n_proc == self.batch_frames:
                yield group
                group, n_proc = [], 0
        if group:
            yield group

    def _detect_batched(self, video_path):
        """detect() through the batched pipeline: identical per-frame results, frames are read batch_frames ahead."""
        from . import _lib, pipeline as pl
        det = self.image_detector
        if self._pipe is None:
            if det.model.batch_max < self.batch_frames:
                det.model.set_batch_max(self.batch_frames)
            self._pipe = pl.Pipeline(det.model, self.tracker, det.thres, det.nms_thres, class_mask=self.class_mask)
        hold_detections = None

        def upload(group):
            fr = [f for f, proc in group if proc]
            return (_lib.DeviceBuffer.from_array(np.stack(fr, 0)), fr[0].shape[0], fr[0].shape[1], len(fr)) if fr else None

        groups = self._processed_batches(video_path)
        cur = next(groups, None)
        cur_dev = upload(cur) if cur is not None else None
        while cur is not None:
            nxt = next(groups, None)
            nxt_dev = upload(nxt) if nxt is not None else None
            outs = []
            if cur_dev is not None:
                buf, h, w, n = cur_dev
                ahead = nxt_dev[0].offset(0) if nxt_dev is not None and nxt_dev[1:] == (h, w, n) else None
                outs = self._pipe.step(buf.offset(0), h, w, n, ahead)
            k = 0
            for frame, proc in cur:
                actions = []
                if proc:
                    o = outs[k]
                    k += 1
                    hold_detections = None if o is None else (o if len(o) else [])
                    if self.action_id is not None and hold_detections is not None:
                        actions = self.action_id.update(hold_detections)
                yield np.ascontiguousarray(frame[:, :, ::-1]), hold_detections, actions
            if cur_dev is not None:
                cur_dev[0].free()
            cur, cur_dev = nxt, nxt_dev

    def detect(self, video_path, output_path=None, skip_secs=0, real_show=False, show_fps=True):
        # (the tracker-side NMS option reorders detections on the host, so it keeps the frame-by-frame path)
        if (self.batch_frames > 1 and self.tracker is not None and self.image_detector.win_size is None
                and getattr(self.tracker, "nms_max_overlap", 1) == 1):
            yield from self._detect_batched(video_path)
            return
        hold_detections, actions, frames = None, [], 0
        for frame in self._frames(video_path):
            if frame is None:
                break
            if frames % self.skip_frames == 0:
                hold_detections = self.process(frame)
                if self.action_id is not None and hold_detections is not None and self.tracker is not None:
                    actions = self.action_id.update(hold_detections)
                else:
                    actions = []
                frames = 0
            else:
                actions = []
            if hold_detections is not None:
                if self.tracker is not None:
                    image, _, _ = self.label_drawer.draw_labels_by_trackers(frame, hold_detections, only_rect=False)
                else:
                    image, _, _ = self.label_drawer.draw_labels(frame, hold_detections, only_rect=False)
            else:
                image = frame
            result = np.ascontiguousarray(image[:, :, ::-1])       # RGB -> BGR
            frames += 1
            yield result, hold_detections, actions
