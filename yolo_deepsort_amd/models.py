"""Drop-in ``Darknet`` (reference yolo3/models/models.py:277-394) backed by libydsort.

Same constructor, attributes and call convention as the reference class; the graph
is planned and executed by the HIP engine (csrc/darknet.cpp).  No torch autograd:
this is an inference engine, ``parameters()`` exists only so that
``ImageDetector`` can read a device off it (img_detect.py:47).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .loaders import parse_model_config


def _to_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


class _Param:
    """Minimal stand-in for a torch parameter: carries the device only."""

    def __init__(self, device):
        self.device = device


class Darknet:
    def __init__(self, config_path, img_size=416, batch_max=1, cfg_text=None):
        _lib.init()            # the device this process is bound to (LOCAL_RANK under torch.distributed.run)
        if cfg_text is None:
            with open(config_path, "r") as f:
                cfg_text = f.read()
        self.cfg_text = cfg_text
        defs = parse_model_config(config_path, text=cfg_text)
        self.hyperparams = defs.pop(0)
        self.module_defs = defs
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.seen = 0
        self.header_info = np.array([0, 0, 0, self.seen, 0], dtype=np.int32)
        self.batch_max = int(batch_max)
        self.device = f"cuda:{_lib.current_device()}"
        self._half = False
        self._h = None
        self._create()

    def _create(self):
        lib = _lib.load()
        if self._h:
            lib.yds_darknet_destroy(self._h)
        self._h = _lib.check_ptr(lib.yds_darknet_create(self.cfg_text.encode(), self.img_size[0], self.img_size[1],
                                                        self.batch_max))
        self.num_boxes = lib.yds_darknet_num_boxes(self._h)
        self.num_attrs = lib.yds_darknet_num_attrs(self._h)
        blob = getattr(self, "_weights_blob", None)
        if blob is not None:
            _lib.check(lib.yds_darknet_load_weights(self._h, blob, len(blob), self._cutoff))

    def set_batch_max(self, batch_max):
        """Grow (or shrink) the activation buffers in place: the C handle, and every pipeline bound to it, stay valid."""
        if batch_max != self.batch_max:
            _lib.check(_lib.load().yds_darknet_set_batch_max(self._h, int(batch_max)))
            self.batch_max = int(batch_max)

    # -- reference API -------------------------------------------------
    def load_darknet_weights(self, weights_path, blob=None):
        """models.py:315-366; ``blob`` lets callers pass the file content directly."""
        if blob is None:
            with open(weights_path, "rb") as f:
                blob = f.read()
        self.header_info = np.frombuffer(blob[:20], dtype=np.int32).copy()
        self.seen = self.header_info[3]
        self._cutoff = 75 if (weights_path and "darknet53.conv.74" in str(weights_path)) else -1
        self._weights_blob = bytes(blob)
        _lib.check(_lib.load().yds_darknet_load_weights(self._h, self._weights_blob, len(blob), self._cutoff))

    def save_darknet_weights(self, path, cutoff=-1):
        """models.py:368-394: header (with ``seen``) then, per conv block up to ``cutoff``, [beta, gamma, mean, var]
        or [bias] and the weights.  The engine keeps the loaded file image, so saving re-emits exactly the floats
        that were loaded (inference never changes them)."""
        blob = getattr(self, "_weights_blob", None)
        if blob is None:
            raise RuntimeError("no weights loaded")
        from .synth import conv_shapes
        n_blocks = len(self.module_defs)
        stop = n_blocks if cutoff == -1 else (cutoff if cutoff >= 0 else n_blocks + cutoff)
        header = np.array(self.header_info, dtype=np.int32).copy()
        header[3] = self.seen
        floats = 0
        for idx, cin, cout, k, bn, _, _ in conv_shapes(self.cfg_text):
            if idx >= stop:
                break
            floats += (4 if bn else 1) * cout + cout * cin * k * k
        with open(path, "wb") as f:
            f.write(header.tobytes())
            f.write(blob[20:20 + floats * 4])

    def to(self, device):
        self.device = str(device)
        return self

    def cuda(self):
        return self.to(f"cuda:{_lib.current_device()}")

    def eval(self):
        return self

    def half(self):
        """img_detect.py:49-50 ``model.half()``: the convolutions switch to single-term fp16 operands (fp32 accumulation);
        fp16-class accuracy, see tests/test_gpu_detector.py::test_half_mode for the measured tolerance."""
        _lib.check(_lib.load().yds_darknet_set_half(self._h, 1))
        self._half = True
        return self

    def float(self):
        _lib.check(_lib.load().yds_darknet_set_half(self._h, 0))
        self._half = False
        return self

    def parameters(self):
        yield _Param(self.device)

    def forward(self, x, targets=None):
        if targets is not None:
            raise NotImplementedError("training loss is out of scope (SURVEY 8a)")
        xn = _to_numpy(x)
        if xn.ndim != 4 or xn.shape[2:] != self.img_size:
            raise ValueError(f"expected [B,C,{self.img_size[0]},{self.img_size[1]}], got {xn.shape}")
        b = xn.shape[0]
        if b > self.batch_max:
            self.set_batch_max(b)
        out = np.empty((b, self.num_boxes, self.num_attrs), np.float32)
        _lib.check(_lib.load().yds_darknet_forward_f32(self._h, _lib.ptr(xn), b, _lib.ptr(out)))
        return _wrap_like(x, out)

    __call__ = forward

    # -- engine extras ---------------------------------------------------
    def forward_u8(self, frames, want_output=True):
        """frames uint8 [B,H,W,3] (host) -> decoded predictions [B,N,5+C]."""
        self._last_frame_src = frames                      # identity of what was uploaded (see last_frame_dev)
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim == 3:
            frames = frames[None]
        b, h, w, _ = frames.shape
        if b > self.batch_max:
            self.set_batch_max(b)
        out = np.empty((b, self.num_boxes, self.num_attrs), np.float32) if want_output else None
        _lib.check(_lib.load().yds_darknet_forward_u8(self._h, _lib.ptr(frames), h, w, b, _lib.ptr(out)))
        return out

    def last_frame_dev(self, frame):
        """Device pointer of `frame` if it is the single frame the last forward_u8 uploaded, else None."""
        if getattr(self, "_last_frame_src", None) is not frame:
            return None
        h, w, b = C.c_int(), C.c_int(), C.c_int()
        p = _lib.load().yds_darknet_last_frames_dev(self._h, C.byref(h), C.byref(w), C.byref(b))
        if not p or b.value != 1 or (h.value, w.value) != tuple(frame.shape[:2]):
            return None
        return C.c_void_p(p)

    def nms(self, image, conf_thres, iou_thres, frame_hw=None, cap=300):
        """soft_non_max_suppression (+ resize_boxes when frame_hw is given) on the last forward."""
        out = np.empty((cap, 6), np.float32)
        n = C.c_int(0)
        fh, fw = frame_hw if frame_hw is not None else (0, 0)
        _lib.check(_lib.load().yds_nms(self._h, image, conf_thres, iou_thres, fh, fw, _lib.ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def detect_tiled(self, frame, tiles, conf_thres, iou_thres, cap=300):
        """Sliding-window detection of one host frame over `tiles` = [(x, y, tile_h, tile_w), ...]
        (ImageDetector.detect win_size branch, yolo3/detect/img_detect.py:97-151) -> [n,6] in frame pixels."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = frame.shape
        rects = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, 4)
        out = np.empty((cap, 6), np.float32)
        n = C.c_int(0)
        _lib.check(_lib.load().yds_detect_tiled(self._h, _lib.ptr(frame), h, w, _lib.ptr(rects), rects.shape[0], conf_thres, iou_thres,
                                                _lib.ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def layer_shape(self, i):
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.load().yds_darknet_layer_shape(self._h, i, C.byref(c), C.byref(h), C.byref(w)))
        return c.value, h.value, w.value

    def layer_format(self, i):
        """Storage format of layer i's output as the graph currently runs it: 0 fp32, 1 split-fp16 record, 2 fp16 (half mode)."""
        return int(_lib.load().yds_darknet_layer_format(self._h, i))

    def layer_output(self, i, batch=1):
        c, h, w = self.layer_shape(i)
        out = np.empty((batch, c, h, w), np.float32)
        _lib.check(_lib.load().yds_darknet_layer_output(self._h, i, batch, _lib.ptr(out)))
        return out

    def get_input(self, batch=1):
        out = np.empty((batch, int(self.hyperparams["channels"]), *self.img_size), np.float32)
        _lib.check(_lib.load().yds_darknet_get_input(self._h, batch, _lib.ptr(out)))
        return out

    def set_injection(self, image, rows, logit=6.0):
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, 9)
        _lib.check(_lib.load().yds_darknet_set_injection(self._h, image, _lib.ptr(rows), rows.shape[0], logit))

    def yolo_heads(self):
        """[(H, W, [(aw, ah)...])] per yolo layer, in network order (for synth.head_injection)."""
        heads = []
        for i, d in enumerate(self.module_defs):
            if d["type"] == "yolo":
                _, h, w = self.layer_shape(i)
                idx = [int(v) for v in d["mask"].split(",")]
                a = [int(v) for v in d["anchors"].split(",")]
                heads.append((h, w, [(a[2 * j], a[2 * j + 1]) for j in idx]))
        return heads

    def conv_flops(self):
        return int(_lib.load().yds_darknet_conv_flops(self._h))

    def __del__(self):
        try:
            if self._h:
                _lib.load().yds_darknet_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _wrap_like(x, arr):
    """Return a torch tensor when the caller passed one, else numpy."""
    if hasattr(x, "detach"):
        import torch
        return torch.from_numpy(arr)
    return arr
