"""Import the real reference (/root/reference) in THIS container (TEST ORACLE tooling).

The reference needs four shims to import on torch 2.10 / numpy 2.2 without
torchvision, cv2 and imutils.  They are installed here, before
the import.  Nothing from the reference is copied; this module only runs where
/root/reference exists (never on the GPU box) and is used by gen_golden.py and
by CPU tests that cross-check the oracle against the live reference.
"""

import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
# path -> dict(frames=uint8 [N,H,W,3] BGR, fps=float, on_read=callable(t) | None): clips the FileVideoStream shim serves
CLIPS = {}


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "yolo3"))


def _nms_stub(boxes, scores, iou_threshold):
    import torch
    from .nms import nms_greedy
    keep = nms_greedy(boxes.detach().cpu().numpy().astype(np.float32),
                      scores.detach().cpu().numpy().astype(np.float32), iou_threshold)
    return torch.from_numpy(keep)


def install_shims():
    import torch
    if not hasattr(np, "float"):
        # deep_sort/sort/preprocessing.py:41 uses the alias numpy removed in 1.24 (`boxes.astype(np.float)` = float64)
        np.float = float
    if not hasattr(torch, "solve") or getattr(torch.solve, "_yds_shim", False) is False:
        def solve(B, A):
            return torch.linalg.solve(A, B), None
        solve._yds_shim = True
        torch.solve = solve
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        boxes = types.ModuleType("torchvision.ops.boxes")
        boxes.nms = _nms_stub

        def batched_nms(b, s, idxs, thr):
            off = idxs.to(b) * (b.max() + 1)
            return _nms_stub(b + off[:, None], s, thr)
        boxes.batched_nms = batched_nms
        ops.boxes = boxes
        ops.nms = _nms_stub
        tv.ops = ops
        tr = types.ModuleType("torchvision.transforms")
        tv.transforms = tr
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops,
                            "torchvision.ops.boxes": boxes, "torchvision.transforms": tr})
    if "cv2" not in sys.modules:
        from .resize import resize_bilinear_u8
        cv2 = types.ModuleType("cv2")
        cv2.INTER_LINEAR = 1
        cv2.VideoWriter_fourcc = lambda *c: "".join(c)          # a default argument of VideoDetector.__init__ (video_detect.py:49)
        cv2.COLOR_RGB2BGR = 4
        cv2.COLOR_BGR2RGB = 4
        cv2.FONT_HERSHEY_SIMPLEX = 0
        cv2.FONT_HERSHEY_COMPLEX_SMALL = 5
        cv2.WINDOW_NORMAL = 0
        # capture properties VideoDetector.detect queries (video_detect.py:91-101); served by the FileVideoStream shim below
        cv2.CAP_PROP_POS_FRAMES, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT = 1, 3, 4
        cv2.CAP_PROP_FPS, cv2.CAP_PROP_FOURCC, cv2.CAP_PROP_FRAME_COUNT = 5, 6, 7

        def resize(img, size, interpolation=1):
            return resize_bilinear_u8(np.asarray(img), size)
        cv2.resize = resize
        cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
        for name in ("rectangle", "putText", "imshow", "waitKey", "destroyAllWindows", "line", "circle", "namedWindow", "resizeWindow"):
            setattr(cv2, name, lambda *a, **k: None)
        cv2.getTextSize = lambda *a, **k: ((0, 0), 0)

        class VideoWriter:                                       # video_detect.py:106,189,202: records what the generator wrote
            written = []

            def __init__(self, path, fourcc, fps, size):
                self.args = (path, fourcc, fps, size)
                VideoWriter.written.append(self)
                self.frames = 0

            def write(self, frame):
                self.frames += 1

            def release(self):
                pass
        cv2.VideoWriter = VideoWriter
        sys.modules["cv2"] = cv2
    if "imutils" not in sys.modules:
        im = types.ModuleType("imutils")
        vid = types.ModuleType("imutils.video")

        class _Capture:
            """cv2.VideoCapture stand-in over an in-memory clip (the queries of video_detect.py:89-101)."""

            def __init__(self, src):
                self.src, self.pos = src, 0

            def isOpened(self):
                return True

            def get(self, prop):
                n, h, w = self.src["frames"].shape[:3]
                return {5: self.src.get("fps", 25.0), 6: 0.0, 3: float(w), 4: float(h), 7: float(n), 1: float(self.pos)}[prop]

            def set(self, prop, value):
                assert prop == 1
                self.pos = int(value)

        class FileVideoStream:
            """imutils.video.FileVideoStream stand-in: serves the BGR frames registered under `path` in CLIPS one by one on
            the caller's thread (the real class decodes on a reader thread into a queue; the consumer-side protocol -
            .stream, start(), more(), read() with the transform applied - is the same).  `on_read(t)` lets a fixture
            script set per-frame state (the head-logit injection) right before the generator receives frame t."""

            def __init__(self, path, transform=None, queue_size=128):
                if path not in CLIPS:
                    raise IOError("no clip registered for %r" % (path,))
                self.src = CLIPS[path]
                self.stream = _Capture(self.src)
                self.transform = transform

            def start(self):
                return self

            def more(self):
                return self.stream.pos < len(self.src["frames"])

            def read(self):
                t = self.stream.pos
                if t >= len(self.src["frames"]):
                    return None
                self.stream.pos += 1
                if self.src.get("on_read") is not None:
                    self.src["on_read"](t)
                frame = np.array(self.src["frames"][t])
                return self.transform(frame) if self.transform is not None else frame

            def stop(self):
                pass
        vid.FileVideoStream = FileVideoStream
        im.video = vid
        sys.modules.update({"imutils": im, "imutils.video": vid})


def import_reference():
    """Returns a namespace with the reference modules used by the fixtures."""
    if not available():
        raise RuntimeError("reference tree not present")
    install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # the repo ships same-named drop-in packages (yolo3/, deep_sort/): make sure the
    # reference's own modules are the ones imported here
    for k in [k for k in sys.modules if k.split(".")[0] in ("yolo3", "deep_sort", "action")]:
        del sys.modules[k]
    sys.dont_write_bytecode = True
    ns = types.SimpleNamespace()
    import yolo3.models.models as models
    import yolo3.utils.model_build as model_build
    import yolo3.utils.parse_config as parse_config
    import yolo3.detect.img_detect as img_detect
    import yolo3.detect.video_detect as video_detect
    import deep_sort.deep_sort as ds
    import deep_sort.deep.model as reid_model
    import deep_sort.deep.feature_extractor as fe
    import deep_sort.sort.kalman_filter as kf
    import deep_sort.sort.linear_assignment as la
    import deep_sort.sort.iou_matching as iou
    import deep_sort.sort.nn_matching as nn
    import deep_sort.sort.tracker as tracker
    for m in (models, model_build, parse_config, img_detect, video_detect, ds, reid_model, fe, kf, la, iou, nn, tracker):
        assert m.__file__.startswith(REF_ROOT), m.__file__
    ns.models, ns.model_build, ns.parse_config, ns.img_detect = models, model_build, parse_config, img_detect
    ns.video_detect = video_detect
    ns.deep_sort, ns.reid_model, ns.feature_extractor = ds, reid_model, fe
    ns.kalman_filter, ns.linear_assignment, ns.iou_matching, ns.nn_matching, ns.tracker = kf, la, iou, nn, tracker
    return ns


def release_reference():
    """Drop the reference's modules so the repo's own drop-in packages can be imported again."""
    for k in [k for k in sys.modules if k.split(".")[0] in ("yolo3", "deep_sort", "action")]:
        del sys.modules[k]
    if REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)
