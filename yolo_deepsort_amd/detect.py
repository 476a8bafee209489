"""Drop-in ``ImageDetector`` / ``VideoDetector`` (reference yolo3/detect/img_detect.py:36-153,
yolo3/detect/video_detect.py:39-208).  Only the per-frame hot glue is accelerated; video decode,
drawing and display stay host side and are optional (cv2 / imutils are not required to import this)."""

from __future__ import annotations

import logging
import os
import time
from functools import reduce

import numpy as np

from .label_draw import LabelDrawer, put_text
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


class FileVideoStream:
    """Threaded frame reader with a bounded queue and a per-frame transform - the role of ``imutils.video.FileVideoStream``
    in the reference (yolo3/detect/video_detect.py:12,86,112-126: ``FileVideoStream(path, _transform).start()``, ``more()``,
    ``read()``).  Sources: a video file / camera index through cv2.VideoCapture when cv2 is installed, or an ``.npy`` file
    holding uint8 [N,H,W,3] frames (what the tests and synthetic streams use), or any object with cv2.VideoCapture's
    ``isOpened / read / get / set / release`` (an already-open capture, an in-memory clip).  Frames are BGR like cv2's;
    ``transform`` is applied on the reader thread (the reference passes BGR->RGB)."""

    # cv2.CAP_PROP_* ordinals (stable across OpenCV 3.x / 4.x), so that capture-like sources work without cv2
    CAP_PROP_POS_FRAMES, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT = 1, 5, 7

    def __init__(self, path, transform=None, queue_size=128):
        import queue
        self.transform = transform
        self.stopped = False
        self.Q = queue.Queue(maxsize=queue_size)
        self._frames = None
        self._cap = None
        if isinstance(path, str) and path.endswith(".npy"):
            self._frames = iter(np.load(path, mmap_mode="r"))
        elif all(hasattr(path, a) for a in ("isOpened", "read", "get", "set")):
            self._cap = path
            if not self._cap.isOpened():
                raise IOError("Couldn't open webcam or video")
        else:
            try:
                import cv2
            except ImportError:
                raise IOError("Couldn't open webcam or video")
            self._cap = cv2.VideoCapture(path)
            if not self._cap.isOpened():
                raise IOError("Couldn't open webcam or video")
        self._thread = None

    def start(self):
        import threading
        self._thread = threading.Thread(target=self._update, daemon=True)
        self._thread.start()
        return self

    def _next(self):
        if self._frames is not None:
            f = next(self._frames, None)
            return None if f is None else np.array(f)
        ok, f = self._cap.read()
        return f if ok else None

    def _update(self):
        import queue
        try:
            while not self.stopped:
                frame = self._next()
                if frame is None:
                    break
                if self.transform is not None:
                    frame = self.transform(frame)
                while not self.stopped:            # a full queue must not pin the thread once the consumer has gone
                    try:
                        self.Q.put(frame, timeout=0.05)
                        break
                    except queue.Full:
                        pass
        finally:
            self.stopped = True
            if self._cap is not None and hasattr(self._cap, "release"):   # released on the thread that reads it (never concurrently with read())
                self._cap.release()
            try:
                self.Q.put_nowait(None)            # end marker wakes a blocked reader
            except queue.Full:
                pass

    def more(self):
        return not (self.stopped and self.Q.empty())

    def read(self):
        import queue
        while True:
            try:
                return self.Q.get(timeout=0.05)
            except queue.Empty:
                if self.stopped and self.Q.empty():
                    return None

    def fps(self):
        """video_detect.py:92: int(CAP_PROP_FPS) of a capture - the reference truncates the rate before it uses it for the seek
        and for the writer (29.97 -> 29).  None for .npy sources (no rate stored with them)."""
        if self._cap is None:
            return None
        v = self._cap.get(self.CAP_PROP_FPS)
        return int(v) if v and v > 0 else None

    def seek_secs(self, skip_secs):
        """video_detect.py:97-101: skip_frames = int(skip_secs) * int(fps); the seek is refused (with the reference's message)
        when skip_secs exceeds the frame count; before start().  .npy sources count 25 frames per second."""
        if self._thread is not None:
            raise RuntimeError("seek_secs must be called before start()")
        if self._cap is not None:
            total = int(self._cap.get(self.CAP_PROP_FRAME_COUNT))
            if skip_secs > total:
                print("Can't skip over total video!")
            else:
                self._cap.set(self.CAP_PROP_POS_FRAMES, int(skip_secs) * (self.fps() or 0))
        else:
            for _ in range(int(skip_secs) * 25):
                if next(self._frames, None) is None:
                    break

    def stop(self):
        """Ends the reader thread (drains the queue so a blocked put() returns, joins) - safe when the consumer stops early."""
        import queue
        self.stopped = True
        t = self._thread
        if t is not None and t.is_alive():
            while t.is_alive():
                try:
                    while True:
                        self.Q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)
        elif t is None and self._cap is not None and hasattr(self._cap, "release"):
            self._cap.release()
        try:
            while True:
                self.Q.get_nowait()
        except queue.Empty:
            pass


def _transform(frame):
    """video_detect.py:33-36: BGR -> RGB"""
    return None if frame is None else np.ascontiguousarray(frame[:, :, ::-1])


class VideoDetector:
    def __init__(self, model, class_path, thickness=2, font_path=None, font_size=10, thres=0.7, nms_thres=0.4,
                 skip_frames=-1, fourcc="mp4v", class_mask=None, win_size=None, overlap=0.15, tracker=None,
                 action_id=None, half=False, batch_frames=None, device_overlay=True):
        # batch_frames (not in the reference): with a tracker, read that many frames ahead and run them through the batched device
        # pipeline (csrc/pipeline.cpp) - same results per frame, yielded in order, several times the frame rate of the frame-by-frame
        # path.  None (default, round 5) = by source: AUTO_BATCH frames for a file / an .npy / an iterable of frames - the reference
        # itself decodes up to 128 frames ahead of the detector for those (FileVideoStream queue, video_detect.py:86) - and 1 for a
        # live source (a camera index, a stream URL), where reading ahead would be waiting.  1 = the reference's latency (one frame
        # in, one result out).  device_overlay: the output stage (overlay, RGB -> BGR, FPS text) of the batched path on the device.
        self.batch_frames = None if batch_frames is None else max(1, int(batch_frames))
        self.device_overlay = bool(device_overlay) and not os.environ.get("YDS_HOST_OVERLAY")
        self._pipe = None
        self.thickness = thickness
        self.skip_frames = skip_frames
        self.class_names = load_classes(class_path)
        self.fourcc = fourcc
        self.class_mask = class_mask
        self.tracker = tracker
        self.action_id = action_id
        self._fps_prev, self._fps_acc, self._fps_cnt, self._fps_text = time.time(), 0.0, 0, "FPS: ??"
        self.label_drawer = LabelDrawer(self.class_names, font_path=font_path, font_size=font_size, thickness=thickness,
                                        img_size=getattr(model, "img_size", None))
        self.image_detector = ImageDetector(model, class_path, thickness=thickness, thres=thres, nms_thres=nms_thres,
                                            win_size=win_size, overlap=overlap, half=half)

    AUTO_BATCH = 68          # frames per step of the batched path when batch_frames is left to the source: bench.py's step size for the 608 x 608 nets
    #                          (workload.DEFAULT_BATCH says why 68; the reference itself reads 128 frames ahead); a shorter clip is read whole

    @staticmethod
    def _is_live(video_path):
        """A source that cannot be read ahead of real time: a camera index (cv2.VideoCapture(0), video_detect.py:86 passes the path
        through) or a network stream URL."""
        if isinstance(video_path, int):
            return True
        if hasattr(video_path, "isOpened"):                          # an already-open capture: live unless it knows its frame count
            try:
                return float(video_path.get(7)) <= 0                 # cv2.CAP_PROP_FRAME_COUNT (0 / -1 for cameras and streams)
            except Exception:
                return True
        return isinstance(video_path, str) and (video_path.isdigit() or "://" in video_path)

    def _batchable(self):
        """The batched device pipeline drives the package's own DeepSort (its Extractor and device tracker handles); a tracker that
        only offers update() - a DeepSort built around a user callable, any custom tracker - keeps the frame-by-frame loop."""
        from .deep_sort import DeepSort, Extractor
        t = self.tracker
        return (isinstance(t, DeepSort) and isinstance(getattr(t, "extractor", None), Extractor)
                and hasattr(t.extractor, "_h") and hasattr(getattr(t, "tracker", None), "_h")
                and self.image_detector.win_size is None and getattr(t, "nms_max_overlap", 1) == 1)

    def _frames(self, video_path, skip_secs=0, transform=True):
        """RGB frames of the source (transform=False: a capture / file source's frames as decoded, BGR)."""
        self._source_fps = None
        if hasattr(video_path, "__iter__") and not isinstance(video_path, (str, bytes)) and not hasattr(video_path, "isOpened"):
            if skip_secs:
                raise ValueError("skip_secs needs a seekable source (a video file or an .npy path), not an iterable of frames")
            for f in video_path:           # already-decoded RGB frames
                yield f
            return
        fvs = FileVideoStream(video_path, _transform if transform else None)   # video_detect.py:86: decode thread + BGR -> RGB
        self._source_fps = fvs.fps()
        fvs.seek_secs(skip_secs)                                     # video_detect.py:99-101
        fvs.start()
        try:
            while fvs.more():
                frame = fvs.read()
                if frame is None:
                    return
                yield frame
        finally:
            fvs.stop()

    def process(self, frame):
        """The hot glue of video_detect.py:134-157 for one frame: returns hold_detections (self._tracked: the tracker branch
        :137-154 ran, i.e. the detector returned rows and a tracker is set)."""
        detections = self.image_detector.detect(frame)
        self._tracked = detections is not None and self.tracker is not None
        if self._tracked:
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

    def _processed_batches(self, video_path, skip_secs=0, transform=True):
        """Groups of consecutive frames holding up to batch_frames frames that pass the skip_frames gate."""
        group, n_proc, frames = [], 0, 0
        bf = getattr(self, "_batch_now", None) or self.batch_frames or 1
        for frame in self._frames(video_path, skip_secs, transform):
            if frame is None:
                break
            proc = frames % self.skip_frames == 0
            if proc:
                frames = 0
            frames += 1
            group.append((frame, proc))
            n_proc += proc
            if n_proc == bf:
                yield group
                group, n_proc = [], 0
        if group:
            yield group

    def _render_host(self, frame, hold_detections, fps_text):
        """video_detect.py:161-186 on the host: overlay (on a copy: callers may hand in their own frame arrays), RGB -> BGR, FPS text."""
        image = frame
        if hold_detections is not None and len(hold_detections):
            image = frame.copy()
            if self.tracker is not None:
                self.label_drawer.draw_labels_by_trackers(image, hold_detections, only_rect=False)
            else:
                self.label_drawer.draw_labels(image, hold_detections, only_rect=False)
        result = np.ascontiguousarray(image[:, :, ::-1])           # RGB -> BGR
        if fps_text:
            put_text(result, fps_text, (3, 15), 2, (255, 0, 0))
        return result

    def _render(self, frame, hold_detections, show_fps):
        now = time.time()
        text = self._fps_tick(now - self._fps_prev)
        self._fps_prev = now
        return self._render_host(frame, hold_detections, text if show_fps else None)

    def _fps_tick(self, dt):
        """The FPS counter of video_detect.py:176-182 advanced by one yielded frame that took `dt` seconds; returns the text to show."""
        self._fps_acc += dt
        self._fps_cnt += 1
        if self._fps_acc > 1:
            self._fps_acc -= 1
            self._fps_text = "FPS: " + str(self._fps_cnt)
            self._fps_cnt = 0
        return self._fps_text

    def _stage_batches(self, video_path, skip_secs, bgr, out_q, free_q, stop):
        """Reader side of the batched path (its own thread): groups of frames -> a pinned staging block -> HBM (synchronous copy
        on this thread, the GIL released), channel-swapped there when the source delivers BGR.  The consumer hands the
        blocks back through free_q once the pipeline and the output stage are done with them."""
        import queue

        def give(item):                                        # never blocks past a stop request
            while not stop.is_set():
                try:
                    out_q.put(item, timeout=0.1)
                    return
                except queue.Full:
                    pass
        try:
            for group in self._processed_batches(video_path, skip_secs, transform=not bgr):
                if stop.is_set():
                    return
                # processed frames first (the pipeline wants them contiguous), the others behind them
                order = [i for i, (_, proc) in enumerate(group) if proc] + [i for i, (_, proc) in enumerate(group) if not proc]
                n_proc = sum(1 for _, proc in group if proc)
                h, w = group[0][0].shape[:2]
                if any(f.shape != (h, w, 3) or f.dtype != np.uint8 for f, _ in group):
                    raise ValueError("VideoDetector: frames of one video must share one uint8 [h, w, 3] shape")
                blk = free_q.get()
                if blk is None or stop.is_set():
                    return
                slot_of = [0] * len(group)
                for slot, i in enumerate(order):
                    slot_of[i] = slot
                self._upload_group(blk, [group[i][0] for i in order], h, w, bgr)
                give(dict(blk=blk, flags=[proc for _, proc in group], slot_of=slot_of, n_proc=n_proc, h=h, w=w))
            give(None)
        except BaseException as e:                             # noqa: BLE001 - re-raised on the consumer's thread
            give(e)

    @staticmethod
    def _upload_group(blk, frames, h, w, bgr):
        """The device side of the reader thread: `frames` (slot order) -> the block's pinned staging array -> its HBM buffer, in the
        source's own channel order (round 6: a BGR source is no longer swapped on the device - Pipeline.set_frame_order makes the
        front end read it as it is).  blk["dev"] / blk["pin"] are (re)allocated when the group outgrows them."""
        from . import _lib
        lib = _lib.load()
        nbytes = len(frames) * h * w * 3
        if blk["dev"] is None or blk["dev"].nbytes < nbytes:
            blk["dev"] = _lib.DeviceBuffer(nbytes)
            blk["pin"] = _lib.PinnedArray((nbytes,), np.uint8)
        stage = blk["pin"].array[:nbytes].reshape(len(frames), h, w, 3)
        for slot, f in enumerate(frames):
            np.copyto(stage[slot], f)
        _lib.check(lib.yds_memcpy_h2d(blk["dev"].ptr, _lib.ptr(stage), nbytes))
        blk["n"] = len(frames)            # (a BGR source stays BGR in HBM: the pipeline reads it in that order, the output stage draws on it)

    def _render_batch(self, cur, holds, fps_texts, bgr):
        """Output stage of one staged batch: a list of BGR frames in group order.  Device form: csrc/overlay.hip over the frames in
        HBM; host form (device_overlay=False / YDS_HOST_OVERLAY: A/B runs and tests): label_draw.py on the staged copy."""
        h, w, n = cur["h"], cur["w"], len(holds)
        if self.device_overlay:
            from .label_draw import DeviceOverlay
            if getattr(self, "_overlay", None) is None:
                self._overlay = DeviceOverlay(self.label_drawer)
            return self._overlay.render(cur["blk"]["dev"].offset(0), cur["slot_of"], h, w, holds, fps_texts,
                                        bgr_frames=cur["blk"].get("n", n) if bgr else 0)
        stage = cur["blk"]["pin"].array[:n * h * w * 3].reshape(n, h, w, 3)
        if bgr:
            stage = stage[..., ::-1]
        return [self._render_host(np.ascontiguousarray(stage[cur["slot_of"][i]]), holds[i], fps_texts[i] if fps_texts else None) for i in range(n)]

    def _detect_batched(self, video_path, show_fps=True, skip_secs=0, batch_frames=None):
        """detect() through the batched pipeline: identical per-frame results, frames are read batch_frames ahead (the reference itself
        decodes up to 128 frames ahead, video_detect.py:86).  Round 5: frames go reader thread -> pinned staging -> HBM once; the
        pipeline and the OUTPUT STAGE (overlay, RGB -> BGR, FPS text: csrc/overlay.hip) both read them there, the rendered batch comes
        back in one D2H copy.  A capture / file source is uploaded as the BGR the decoder delivers and channel-swapped on the device
        (the reference converts on the host, video_detect.py:33-36)."""
        import queue
        import threading
        det = self.image_detector
        bf = batch_frames or self.batch_frames or 1
        if self._pipe is None:
            from . import pipeline as pl
            if det.model.batch_max < bf:
                det.model.set_batch_max(bf)
            self._pipe = pl.Pipeline(det.model, self.tracker, det.thres, det.nms_thres, class_mask=self.class_mask)
        self._batch_now = bf
        iterable = hasattr(video_path, "__iter__") and not isinstance(video_path, (str, bytes)) and not hasattr(video_path, "isOpened")
        if hasattr(self._pipe, "set_frame_order"):
            self._pipe.set_frame_order(not iterable)              # a capture / file source is staged as the decoder's BGR
        # Three threads, two hand-overs: reader (stage + upload) -> engine (pipeline steps, hold / action bookkeeping in frame order)
        # -> this generator's thread (output stage + yield).  The engine calls step(i + 1) while batch i is rendered and consumed:
        # with the output stage on the engine's thread the device idled for its 6 ms per batch (1193 against 1605 frames/s).
        out_q, done_q, free_q, stop = queue.Queue(maxsize=1), queue.Queue(maxsize=1), queue.Queue(), threading.Event()
        for _ in range(6):           # one being filled, one in out_q, two with the engine (this batch, the next), one in done_q, one being rendered
            free_q.put(dict(dev=None, pin=None))
        self.host_us = dict(wait_frames=0.0, step=0.0, wait_consumer=0.0, overlay=0.0, wait_engine=0.0, frames=0)
        reader = threading.Thread(target=self._stage_batches, args=(video_path, skip_secs, not iterable, out_q, free_q, stop), daemon=True)
        engine = threading.Thread(target=self._run_engine, args=(out_q, done_q, stop), daemon=True)
        reader.start()
        engine.start()
        t_prev = time.time()
        try:
            while True:
                t0 = time.perf_counter()
                item = done_q.get()
                if isinstance(item, BaseException):
                    raise item
                if item is None:
                    break
                cur, holds, acts = item
                t1 = time.perf_counter()
                now = time.time()
                dt = (now - t_prev) / len(holds)
                t_prev = now
                fps = [self._fps_tick(dt) for _ in holds]
                results = self._render_batch(cur, holds, fps if show_fps else None, not iterable)
                free_q.put(cur["blk"])
                u = self.host_us
                u["wait_engine"] += (t1 - t0) * 1e6; u["overlay"] += (time.perf_counter() - t1) * 1e6; u["frames"] += len(holds)
                for i in range(len(holds)):
                    yield results[i], holds[i], acts[i]
        finally:
            stop.set()
            free_q.put(None)
            for q in (out_q, done_q):                        # unblock a producer waiting on a full hand-over
                try:
                    while True:
                        q.get_nowait()
                except queue.Empty:
                    pass
            engine.join(timeout=5)
            reader.join(timeout=5)

    def _run_engine(self, out_q, done_q, stop):
        """Engine thread of the batched path: one pipeline step per staged batch (the next batch handed over for its early detector
        pass), then the per-frame bookkeeping of video_detect.py:134-159 in frame order - held rows, the action module."""
        import queue

        def take():
            while True:
                try:
                    item = out_q.get(timeout=0.1)
                except queue.Empty:
                    if stop.is_set():
                        return None
                    continue
                if isinstance(item, BaseException):
                    raise item
                return item

        def give(item):
            while not stop.is_set():
                try:
                    done_q.put(item, timeout=0.1)
                    return
                except queue.Full:
                    pass
        try:
            hold_detections, actions = None, []
            u = self.host_us
            t0 = time.perf_counter()
            cur = take()
            u["wait_frames"] += (time.perf_counter() - t0) * 1e6
            while cur is not None and not stop.is_set():
                t0 = time.perf_counter()
                nxt = take()
                t1 = time.perf_counter()
                outs = []
                h, w, n = cur["h"], cur["w"], cur["n_proc"]
                if n:
                    ahead = nxt["blk"]["dev"].offset(0) if nxt is not None and (nxt["h"], nxt["w"], nxt["n_proc"]) == (h, w, n) else None
                    outs = self._pipe.step(cur["blk"]["dev"].offset(0), h, w, n, ahead)
                u["wait_frames"] += (t1 - t0) * 1e6
                u["step"] += (time.perf_counter() - t1) * 1e6
                holds, acts, k = [], [], 0
                for proc in cur["flags"]:
                    if proc:
                        o = outs[k]
                        k += 1
                        hold_detections = None if o is None else (o if len(o) else [])
                        if hold_detections is not None:        # video_detect.py:137-154; a detector-None frame leaves `actions` as it was
                            actions = self.action_id.update(hold_detections) if self.action_id is not None else []
                    else:
                        actions = []                           # :158-159
                    holds.append(hold_detections)
                    acts.append(actions)
                t2 = time.perf_counter()
                give((cur, holds, acts))
                u["wait_consumer"] += (time.perf_counter() - t2) * 1e6
                cur = nxt
            give(None)
        except BaseException as e:                             # noqa: BLE001 - re-raised on the consumer's thread
            give(e)

    def detect(self, video_path, output_path=None, skip_secs=0, real_show=False, show_fps=True):
        """Generator of (bgr_image, hold_detections, actions) like video_detect.py:78-199.  output_path: every result is
        also written - through cv2.VideoWriter when cv2 is installed, as one uint8 [N,H,W,3] array for a path ending in
        ``.npy``; real_show needs cv2 (ignored without it); skip_secs seeks a capture to int(skip_secs) * int(fps) frames like
        video_detect.py:92-101 (an .npy source skips int(skip_secs) * 25 frames; an iterable of frames cannot seek and raises)."""
        writer, frames_out, cv2 = None, None, None
        try:
            import cv2 as _cv2
            cv2 = _cv2
        except ImportError:
            pass
        if output_path is not None:
            if str(output_path).endswith(".npy"):
                frames_out = []
            elif cv2 is None:
                raise IOError("writing %s needs cv2 (or use an .npy path)" % output_path)
        try:
            for result, det, actions in self._detect_impl(video_path, show_fps, skip_secs):
                if frames_out is not None:
                    frames_out.append(result.copy())
                elif output_path is not None:
                    if writer is None:
                        fourcc = cv2.VideoWriter_fourcc(*self.fourcc) if isinstance(self.fourcc, str) else self.fourcc
                        # video_detect.py:92,106: the writer takes int(SOURCE frame rate); 25 only where the source has none (.npy, iterables)
                        writer = cv2.VideoWriter(output_path, fourcc, self._source_fps or 25, (result.shape[1], result.shape[0]))
                    writer.write(result)
                if real_show and cv2 is not None:
                    cv2.imshow("result", result)
                yield result, det, actions
                if real_show and cv2 is not None and cv2.waitKey(1) & 0xFF == ord("q"):
                    break
        finally:
            if writer is not None:
                writer.release()
            if frames_out is not None:
                np.save(output_path, np.stack(frames_out, 0) if frames_out else np.zeros((0, 0, 0, 3), np.uint8))
            if real_show and cv2 is not None:
                cv2.destroyAllWindows()

    @staticmethod
    def _source_len(video_path):
        """Frames the source holds, when it can tell without being consumed (a sequence, an .npy file, an open capture); else None."""
        try:
            if hasattr(video_path, "isOpened"):
                n = int(video_path.get(7))                           # cv2.CAP_PROP_FRAME_COUNT
                return n if n > 0 else None
            if isinstance(video_path, str):
                return int(np.load(video_path, mmap_mode="r").shape[0]) if video_path.endswith(".npy") else None
            return len(video_path) if hasattr(video_path, "__len__") else None
        except Exception:                                             # noqa: BLE001 - a source that cannot tell simply gets the default
            return None

    def _detect_impl(self, video_path, show_fps=True, skip_secs=0):
        # (the tracker-side NMS option reorders detections on the host, so it keeps the frame-by-frame path)
        bf = self.batch_frames if self.batch_frames is not None else (1 if self._is_live(video_path) else self.AUTO_BATCH)
        if self.batch_frames is None and bf > 1:
            n = self._source_len(video_path)                          # a clip shorter than the default read-ahead: no buffers for frames
            if n is not None:                                         # that will never come (every capacity follows the batch size)
                bf = max(1, min(bf, n))
        if bf > 1 and self._batchable():
            yield from self._detect_batched(video_path, show_fps, skip_secs, bf)
            return
        hold_detections, actions, frames = None, [], 0
        for frame in self._frames(video_path, skip_secs):
            if frame is None:
                break
            if frames % self.skip_frames == 0:
                hold_detections = self.process(frame)
                if self._tracked:                          # video_detect.py:137-154; otherwise `actions` keeps its previous value
                    actions = self.action_id.update(hold_detections) if self.action_id is not None else []
                frames = 0
            else:
                actions = []
            result = self._render(frame, hold_detections, show_fps)
            frames += 1
            yield result, hold_detections, actions
