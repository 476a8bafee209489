"""Drop-in ``DeepSort`` / ``Extractor`` (reference deep_sort/deep_sort.py:15-146,
deep_sort/deep/feature_extractor.py:12-58) on libydsort.

``DeepSort.update(bbox_tlwh, confidences, ori_img, payload)`` keeps the reference's
signature and return convention (int32 ndarray [m,6] or ``[]``); crops, ReID CNN,
Kalman filter, cost matrices and the assignment run as HIP kernels, the embedding
never leaves HBM between the extractor and the tracker.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .loaders import load_reid_checkpoint

__all__ = ["DeepSort", "Extractor", "build_tracker"]


def _np(x, dtype=np.float32):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dtype)


class Extractor:
    """ReID feature extractor; ``model_path`` is a ckpt.t7 (``net_dict``) or a state dict."""

    def __init__(self, model_path, use_cuda=True, max_crops=256):
        _lib.init()
        lib = _lib.load()
        self.device = "cuda"
        self.size = (64, 128)
        self.max_crops = int(max_crops)
        sd = model_path if isinstance(model_path, dict) else load_reid_checkpoint(model_path)
        self._h = _lib.check_ptr(lib.yds_reid_create(self.max_crops))
        for name, arr in sd.items():
            a = _np(arr)
            shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            _lib.check(lib.yds_reid_load_tensor(self._h, name.encode(), _lib.ptr(a), shape, max(a.ndim, 1)))
        _lib.check(lib.yds_reid_finalize(self._h))
        if not isinstance(model_path, dict):
            print("Loading weights from {}... Done!".format(model_path))

    # frame + boxes entry (what DeepSort uses)
    def embed(self, frame, tlwh, to_host=True, frame_dev=None):
        """frame_dev: device copy of `frame` (the detector's upload) - skips a second host-to-device copy of the frame."""
        tlwh = _np(tlwh).reshape(-1, 4)
        d = tlwh.shape[0]
        out = np.empty((d, 512), np.float32) if to_host else None
        if d and frame_dev is not None:
            _lib.check(_lib.load().yds_reid_embed_dev(self._h, frame_dev, frame.shape[0], frame.shape[1], _lib.ptr(tlwh), d, _lib.ptr(out)))
        elif d:
            frame = np.ascontiguousarray(frame, dtype=np.uint8)
            _lib.check(_lib.load().yds_reid_embed(self._h, _lib.ptr(frame), frame.shape[0], frame.shape[1],
                                                  _lib.ptr(tlwh), d, _lib.ptr(out)))
        return out

    def preprocess(self, frame, tlwh):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        tlwh = _np(tlwh).reshape(-1, 4)
        out = np.empty((tlwh.shape[0], 3, 128, 64), np.float32)
        _lib.check(_lib.load().yds_reid_preprocess(self._h, _lib.ptr(frame), frame.shape[0], frame.shape[1],
                                                   _lib.ptr(tlwh), tlwh.shape[0], _lib.ptr(out)))
        return out

    def forward(self, batch):
        batch = _np(batch)
        out = np.empty((batch.shape[0], 512), np.float32)
        _lib.check(_lib.load().yds_reid_forward_f32(self._h, _lib.ptr(batch), batch.shape[0], _lib.ptr(out)))
        return out

    def features_dev(self):
        return _lib.load().yds_reid_features_dev(self._h)

    def __call__(self, im_crops):
        """Reference call convention: list of uint8 HxWx3 crops -> [n,512] features.  The crops are
        stacked on one canvas so that each is a box of a single frame."""
        if len(im_crops) == 0:
            return np.zeros((0, 512), np.float32)
        wmax = max(c.shape[1] for c in im_crops)
        htot = sum(c.shape[0] for c in im_crops)
        canvas = np.zeros((htot + 1, wmax + 1, 3), np.uint8)
        tlwh, y = [], 0
        for c in im_crops:
            canvas[y:y + c.shape[0], :c.shape[1]] = c
            tlwh.append((0, y, c.shape[1], c.shape[0]))
            y += c.shape[0]
        feats = self.embed(canvas, np.array(tlwh, np.float32))
        try:
            import torch
            return torch.from_numpy(feats)
        except ImportError:
            return feats

    def __del__(self):
        try:
            if self._h:
                _lib.load().yds_reid_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _TrackView:
    """Read-only snapshot of one track (attribute names of reference deep_sort/sort/track.py)."""

    def __init__(self, track_id, state, tsu, hits, mean, cov):
        self.track_id, self.state, self.time_since_update, self.hits = track_id, state, tsu, hits
        self.mean, self.covariance = mean, cov

    def is_confirmed(self):
        return self.state == 2

    def is_tentative(self):
        return self.state == 1


class _TrackerHandle:
    """Owns the C tracker; ``.tracks`` mirrors ``Tracker.tracks`` of the reference as snapshots."""

    def __init__(self, max_dist, max_iou_distance, max_age, n_init, nn_budget, metric="cosine"):
        if metric not in ("cosine", "euclidean"):
            raise ValueError("Invalid metric; must be either 'euclidean' or 'cosine'")          # nn_matching.py:132-134
        if nn_budget is not None and int(nn_budget) < 1:
            raise ValueError("nn_budget must be a positive integer or None")
        self._h = _lib.check_ptr(_lib.load().yds_tracker_create_ex(
            float(max_dist), float(max_iou_distance), int(max_age), int(n_init), 0 if nn_budget is None else int(nn_budget),
            1 if metric == "euclidean" else 0))
        self.device = "cuda"

    @property
    def tracks(self):
        st = self.state()
        return [_TrackView(int(st["ids"][i]), int(st["state"][i]), int(st["tsu"][i]), int(st["hits"][i]),
                           st["mean"][i:i + 1], st["cov"][i:i + 1]) for i in range(len(st["ids"]))]

    def state(self):
        lib = _lib.load()
        n = lib.yds_tracker_num_tracks(self._h)
        cap = max(n, 1)
        ids, state, tsu, hits = (np.zeros(cap, np.int32) for _ in range(4))
        mean, cov = np.zeros((cap, 8), np.float32), np.zeros((cap, 8, 8), np.float32)
        t = C.c_int(0)
        _lib.check(lib.yds_tracker_get_state(self._h, _lib.ptr(ids), _lib.ptr(state), _lib.ptr(tsu), _lib.ptr(hits),
                                             _lib.ptr(mean), _lib.ptr(cov), cap, C.byref(t)))
        n = t.value
        return dict(ids=ids[:n], state=state[:n], tsu=tsu[:n], hits=hits[:n], mean=mean[:n], cov=cov[:n])

    def step(self, tlwh, feats, payload, feats_dev=None, want_debug=False, feat_rows=None):
        """feat_rows: detection d uses row feat_rows[d] of the feature matrix (survivors of the tracker-side NMS)."""
        lib = _lib.load()
        tlwh = _np(tlwh).reshape(-1, 4)
        payload = _np(payload).reshape(-1)
        d = tlwh.shape[0]
        cap = lib.yds_tracker_num_tracks(self._h) + d + 1
        out = np.zeros((cap, 6), np.int32)
        m = C.c_int(0)
        rows_sel = None if feat_rows is None else np.ascontiguousarray(feat_rows, dtype=np.int32)
        dm = np.zeros((cap, 2), np.int32)
        nm = C.c_int(0)
        if feats_dev is not None:
            src, on_dev = feats_dev, 1
        else:
            n_rows = d if rows_sel is None else (int(rows_sel.max()) + 1 if d else 0)
            feats = _np(feats).reshape(-1, 512) if d else np.zeros((0, 512), np.float32)
            assert feats.shape[0] >= n_rows
            src, on_dev = _lib.ptr(feats), 0
        _lib.check(lib.yds_tracker_step_sel(self._h, _lib.ptr(tlwh), src, on_dev, _lib.ptr(rows_sel), _lib.ptr(payload), d,
                                            _lib.ptr(out), cap, C.byref(m), _lib.ptr(dm), cap, C.byref(nm)))
        dbg = dm[:nm.value].copy()
        rows = out[:m.value].copy()
        return (rows, dbg) if want_debug else rows

    def last_unmatched(self):
        lib = _lib.load()
        cap = 4096
        a, b = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        na, nb = C.c_int(0), C.c_int(0)
        _lib.check(lib.yds_tracker_last_unmatched(self._h, _lib.ptr(a), cap, C.byref(na), _lib.ptr(b), cap, C.byref(nb)))
        return a[:na.value].copy(), b[:nb.value].copy()

    def __del__(self):
        try:
            if self._h:
                _lib.load().yds_tracker_destroy(self._h)
                self._h = None
        except Exception:
            pass


class DeepSort(object):
    def __init__(self, model_path, max_dist=0.2, min_confidence=0.3, nms_max_overlap=1.0, max_iou_distance=0.7,
                 max_age=70, n_init=3, nn_budget=100, use_cuda=False, metric="cosine"):
        self.max_dist = max_dist
        self.min_confidence = min_confidence          # unused by the reference as well (deep_sort.py:51)
        self.nms_max_overlap = nms_max_overlap
        self.max_iou_distance = max_iou_distance
        self.max_age = max_age
        self.n_init = n_init
        self.nn_budget = nn_budget
        self.use_cuda = use_cuda
        _lib.init()
        if isinstance(model_path, (str, dict)):
            self.extractor = Extractor(model_path, use_cuda=use_cuda)
        else:
            self.extractor = model_path
        # the reference hard-wires "cosine" (deep_sort.py:34); `metric` is this build's way to reach the other
        # NearestNeighborDistanceMetric option (nn_matching.py:128-131)
        self.metric = metric
        self.tracker = _TrackerHandle(max_dist, max_iou_distance, max_age, n_init, nn_budget, metric)

    def clone(self):
        return DeepSort(self.extractor, self.max_dist, self.min_confidence, self.nms_max_overlap, self.max_iou_distance,
                        self.max_age, self.n_init, self.nn_budget, self.use_cuda, self.metric)

    def _nms_keep(self, tlwh):
        """deep_sort.py:52-57 + sort/preprocessing.py:6-73: every Detection carries confidence 1 (deep_sort.py:51), so
        ``np.argsort(scores)`` sorts a constant vector - numpy decides the order, as it does in the reference."""
        d = tlwh.shape[0]
        if d == 0:
            return np.zeros(0, np.int32)
        # (Detection.__init__ stores float(confidence), detection.py:29, so the reference sorts a float64 vector of ones: same
        # dtype here - argsort's order among ties is numpy's, per dtype)
        order = np.ascontiguousarray(np.argsort(np.ones(d, dtype=np.float64)), dtype=np.int32)
        pick = np.zeros(d, np.int32)
        n = C.c_int(0)
        _lib.check(_lib.load().yds_tracker_nms(_lib.ptr(tlwh), _lib.ptr(order), d, float(self.nms_max_overlap), _lib.ptr(pick), C.byref(n)))
        return pick[:n.value].copy()

    def update(self, bbox_xywh, confidences, ori_img, payload):
        self.height, self.width = ori_img.shape[:2]
        tlwh = _np(bbox_xywh).reshape(-1, 4)
        d = tlwh.shape[0]
        payload = _np(payload).reshape(-1)
        keep = None
        if isinstance(self.extractor, Extractor):
            hint, self.frame_source = getattr(self, "frame_source", None), None
            frame_dev = hint.last_frame_dev(ori_img) if hint is not None else None      # the detector already uploaded this frame
            self.extractor.embed(ori_img, tlwh, to_host=False, frame_dev=frame_dev)     # features of ALL boxes, like the reference
            if self.nms_max_overlap != 1:
                keep = self._nms_keep(tlwh)
                tlwh, payload = np.ascontiguousarray(tlwh[keep]), np.ascontiguousarray(payload[keep])
            rows = self.tracker.step(tlwh, None, payload, feats_dev=self.extractor.features_dev() if d else None, feat_rows=keep)
        else:                                          # user-supplied extractor callable (reference allows it)
            crops = []
            for x, y, w, h in tlwh:
                x1, y1 = max(int(x), 0), max(int(y), 0)
                x2 = min(int(np.float32(x + w)), self.width - 1)
                y2 = min(int(np.float32(y + h)), self.height - 1)
                crops.append(ori_img[y1:y2, x1:x2])
            feats = _np(self.extractor(crops)) if crops else np.zeros((0, 512), np.float32)
            if self.nms_max_overlap != 1:
                keep = self._nms_keep(tlwh)
                tlwh, payload = np.ascontiguousarray(tlwh[keep]), np.ascontiguousarray(payload[keep])
            rows = self.tracker.step(tlwh, feats, payload, feat_rows=keep)
        return rows if len(rows) else []


def build_tracker(cfg, use_cuda):
    """deep_sort/__init__.py:7-11"""
    return DeepSort(cfg.DEEPSORT.REID_CKPT, max_dist=cfg.DEEPSORT.MAX_DIST, min_confidence=cfg.DEEPSORT.MIN_CONFIDENCE,
                    nms_max_overlap=cfg.DEEPSORT.NMS_MAX_OVERLAP, max_iou_distance=cfg.DEEPSORT.MAX_IOU_DISTANCE,
                    max_age=cfg.DEEPSORT.MAX_AGE, n_init=cfg.DEEPSORT.N_INIT, nn_budget=cfg.DEEPSORT.NN_BUDGET,
                    use_cuda=use_cuda)
