"""The reference's ``deep_sort.sort`` building blocks at their own names (deep_sort/sort/*.py), each backed by the HIP
kernels behind the C ABI - for callers that assemble their own tracker loop instead of using ``DeepSort.update``.

Arrays are numpy (torch tensors are accepted and converted); tracks / detections are the small host objects the reference
uses.  Nothing here computes on the CPU except list bookkeeping: Kalman steps, gating, cosine / euclidean gallery
distances, IOU costs, the linear assignment and the tracker-side NMS all call libydsort.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

INFTY_COST = 1e+5                                   # linear_assignment.py:3
chi2inv95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070, 6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}   # kalman_filter.py:8-17


def _f32(x, shape=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def _lib_ready():
    _lib.init()
    return _lib.load()


# ------------------------------------------------------------------------------------------ detection.py
class Detection(object):
    """deep_sort/sort/detection.py:2-48"""

    def __init__(self, tlwh, confidence, feature, payload=None):
        self.tlwh = _f32(tlwh, (4,))
        self.confidence = float(confidence)
        self.feature = None if feature is None else _f32(feature).reshape(-1)
        self.payload = payload

    def to_tlbr(self):
        ret = self.tlwh.copy()
        ret[2:] += ret[:2]
        return ret

    def to_xyah(self):
        ret = self.tlwh.copy()
        ret[:2] += ret[2:] / 2
        ret[2] /= ret[3]
        return ret


# ------------------------------------------------------------------------------------------ track.py
class TrackState:
    """deep_sort/sort/track.py:1-13"""
    Tentative = 1
    Confirmed = 2
    Deleted = 3


class Track:
    """deep_sort/sort/track.py:16-166 as a snapshot of one row of the device-resident track table."""

    def __init__(self, mean, covariance, track_id, n_init, max_age, feature=None, payload=None, hits=1, age=1,
                 time_since_update=0, state=TrackState.Tentative):
        self.mean, self.covariance = _f32(mean, (1, 8)), _f32(covariance, (1, 8, 8))
        self.track_id, self.hits, self.age, self.time_since_update, self.state = int(track_id), int(hits), int(age), int(time_since_update), int(state)
        self.features = [] if feature is None else [feature]
        self.payload = payload
        self._n_init, self._max_age = n_init, max_age

    def to_tlwh(self):
        ret = self.mean[0, :4].copy()
        ret[2] *= ret[3]
        ret[:2] -= ret[2:] / 2
        return ret

    def to_tlbr(self):
        ret = self.to_tlwh()
        ret[2:] = ret[:2] + ret[2:]
        return ret

    def is_tentative(self):
        return self.state == TrackState.Tentative

    def is_confirmed(self):
        return self.state == TrackState.Confirmed

    def is_deleted(self):
        return self.state == TrackState.Deleted


# ------------------------------------------------------------------------------------------ kalman_filter.py
class KalmanFilter:
    """deep_sort/sort/kalman_filter.py:20-256 on batches: mean [n,8], covariance [n,8,8] (kf_*_kernel in csrc/tracker.hip)."""

    def __init__(self, device="cpu"):
        self.device = device

    def initiate(self, measurement):
        z = _f32(measurement).reshape(-1, 4)
        mean, cov = np.empty((z.shape[0], 8), np.float32), np.empty((z.shape[0], 8, 8), np.float32)
        _lib.check(_lib_ready().yds_kalman_initiate(_lib.ptr(z), z.shape[0], _lib.ptr(mean), _lib.ptr(cov)))
        return mean, cov

    def predict(self, mean, covariance):
        mean, cov = _f32(mean).reshape(-1, 8).copy(), _f32(covariance).reshape(-1, 8, 8).copy()
        _lib.check(_lib_ready().yds_kalman_predict(_lib.ptr(mean), _lib.ptr(cov), mean.shape[0]))
        return mean, cov

    def project(self, mean, covariance):
        mean, cov = _f32(mean).reshape(-1, 8), _f32(covariance).reshape(-1, 8, 8)
        m4, c4 = np.empty((mean.shape[0], 4), np.float32), np.empty((mean.shape[0], 4, 4), np.float32)
        _lib.check(_lib_ready().yds_kalman_project(_lib.ptr(mean), _lib.ptr(cov), mean.shape[0], _lib.ptr(m4), _lib.ptr(c4)))
        return m4, c4

    def update(self, mean, covariance, measurement):
        mean, cov = _f32(mean).reshape(-1, 8).copy(), _f32(covariance).reshape(-1, 8, 8).copy()
        z = _f32(measurement).reshape(-1, 4)
        _lib.check(_lib_ready().yds_kalman_update(_lib.ptr(mean), _lib.ptr(cov), _lib.ptr(z), mean.shape[0]))
        return mean, cov

    def gating_distance(self, mean, covariance, measurements, only_position=False):
        mean, cov = _f32(mean).reshape(-1, 8), _f32(covariance).reshape(-1, 8, 8)
        z = _f32(measurements).reshape(-1, 4)
        out = np.zeros((mean.shape[0], z.shape[0]), np.float32)
        _lib.check(_lib_ready().yds_kalman_gating_ex(_lib.ptr(mean), _lib.ptr(cov), mean.shape[0], _lib.ptr(z), z.shape[0],
                                                     1 if only_position else 0, _lib.ptr(out)))
        return out


# ------------------------------------------------------------------------------------------ nn_matching.py
class NearestNeighborDistanceMetric:
    """deep_sort/sort/nn_matching.py:103-187.  ``distance`` = per target the minimum over its samples of the cosine
    (or squared euclidean) distance to every feature - appearance_cost_kernel through yds_cosine_min_cost /
    yds_euclidean_min_cost.  (The reference's euclidean branch raises inside ``distance``; here it computes
    ``_nn_euclidean_distance`` per target)"""

    def __init__(self, metric, matching_threshold, budget=None):
        if metric not in ("euclidean", "cosine"):
            raise ValueError("Invalid metric; must be either 'euclidean' or 'cosine'")
        self.metric_name = metric
        self.matching_threshold = matching_threshold
        self.budget = budget
        self.samples = {}

    def partial_fit(self, features, targets, active_targets):
        for feature, target in zip(features, targets):
            self.samples.setdefault(target, []).append(_f32(feature).reshape(-1))
            if self.budget is not None:
                self.samples[target] = self.samples[target][-self.budget:]
        self.samples = {k: self.samples[k] for k in active_targets}

    def distance(self, features, targets):
        """nn_matching.py:158-187: cost matrix [len(targets), len(features)].  The embedding width is whatever the caller stores
        (the reference is dimension agnostic); every gallery row must have the width of `features` (ValueError otherwise), and
        the device kernel takes widths up to 512 (narrower rows are zero padded inside the library)."""
        feats = _f32(features)
        if feats.ndim != 2:
            raise ValueError("features must be a [N, dim] matrix, got shape %r" % (feats.shape,))
        dim = feats.shape[1]
        seg, rows = [0], []
        for t in targets:
            rows += self.samples[t]
            seg.append(len(rows))
        out = np.zeros((len(targets), feats.shape[0]), np.float32)
        if not targets or not feats.shape[0]:
            return out
        bad = [r.shape for r in rows if r.shape != (dim,)]
        if bad:
            raise ValueError("gallery rows of width %s do not match features of width %d" % (sorted(set(b[0] for b in bad)), dim))
        if not 1 <= dim <= 512:
            raise ValueError("embedding width %d is outside what the device kernel takes (1..512)" % dim)
        gal = np.ascontiguousarray(np.stack(rows, 0), dtype=np.float32) if rows else np.zeros((0, dim), np.float32)
        seg = np.asarray(seg, np.int32)
        lib = _lib_ready()
        fn = lib.yds_cosine_min_cost if self.metric_name == "cosine" else lib.yds_euclidean_min_cost
        _lib.check(fn(_lib.ptr(gal), _lib.ptr(seg), len(targets), _lib.ptr(feats), feats.shape[0], dim, _lib.ptr(out)))
        return out


# ------------------------------------------------------------------------------------------ preprocessing.py
def non_max_suppression(boxes, max_bbox_overlap, scores=None):
    """deep_sort/sort/preprocessing.py:6-73 (tlwh boxes, float64 arithmetic, +1 pixel convention) -> kept indices in pick order."""
    boxes = _f32(boxes).reshape(-1, 4)
    d = boxes.shape[0]
    if d == 0:
        return []
    order = np.argsort(np.asarray(scores)) if scores is not None else np.argsort(boxes[:, 1].astype(np.float64) + boxes[:, 3])   # :47-50
    order = np.ascontiguousarray(order, dtype=np.int32)
    pick, n = np.zeros(d, np.int32), C.c_int(0)
    _lib.check(_lib_ready().yds_tracker_nms(_lib.ptr(boxes), _lib.ptr(order), d, float(max_bbox_overlap), _lib.ptr(pick), C.byref(n)))
    return pick[:n.value].tolist()


# ------------------------------------------------------------------------------------------ iou_matching.py
def iou(bbox, candidates):
    """deep_sort/sort/iou_matching.py:5-41: bbox tlwh [4], candidates tlwh [n,4] -> [n] (asymmetric +1 in the intersection only)."""
    cand = _f32(candidates).reshape(-1, 4)
    out = np.zeros((1, cand.shape[0]), np.float32)
    if cand.shape[0]:
        _lib.check(_lib_ready().yds_iou_cost(_lib.ptr(_f32(bbox, (1, 4))), 1, _lib.ptr(cand), cand.shape[0], _lib.ptr(out)))
    return 1.0 - out[0]


def iou_cost(tracks, detections, track_indices=None, detection_indices=None):
    """deep_sort/sort/iou_matching.py:44-91: 1 - IoU, rows of tracks with time_since_update > 1 set to INFTY_COST."""
    if track_indices is None:
        track_indices = list(range(len(tracks)))
    if detection_indices is None:
        detection_indices = list(range(len(detections)))
    cost = np.zeros((len(track_indices), len(detection_indices)), np.float32)
    if not cost.size:
        return cost
    tb = np.ascontiguousarray(np.stack([tracks[i].to_tlwh() for i in track_indices], 0), dtype=np.float32)
    db = np.ascontiguousarray(np.stack([detections[i].tlwh for i in detection_indices], 0), dtype=np.float32)
    _lib.check(_lib_ready().yds_iou_cost(_lib.ptr(tb), tb.shape[0], _lib.ptr(db), db.shape[0], _lib.ptr(cost)))
    for row, ti in enumerate(track_indices):
        if tracks[ti].time_since_update > 1:
            cost[row, :] = INFTY_COST
    return cost


# ------------------------------------------------------------------------------------------ linear_assignment.py
def linear_assignment(cost_matrix):
    """scipy.optimize.linear_sum_assignment (call site linear_assignment.py:56) on the lsap kernels: (row_ind, col_ind)."""
    cost = _f32(cost_matrix)
    nr, nc = cost.shape
    n = min(nr, nc)
    rows, cols, k = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), C.c_int(0)
    if n:
        _lib.check(_lib_ready().yds_lsap(_lib.ptr(cost), nr, nc, _lib.ptr(rows), _lib.ptr(cols), C.byref(k)))
    return rows[:k.value].copy(), cols[:k.value].copy()


def min_cost_matching(distance_metric, max_distance, tracks, detections, track_indices=None, detection_indices=None):
    """deep_sort/sort/linear_assignment.py:8-73"""
    if track_indices is None:
        track_indices = list(range(len(tracks)))
    if detection_indices is None:
        detection_indices = list(range(len(detections)))
    if len(detection_indices) == 0 or len(track_indices) == 0:
        return [], track_indices, detection_indices
    cost_matrix = _f32(distance_metric(tracks, detections, track_indices, detection_indices)).copy()
    cost_matrix[cost_matrix > max_distance] = max_distance + 1e-5
    row_indices, col_indices = linear_assignment(cost_matrix)
    matches, unmatched_tracks, unmatched_detections = [], [], []
    cols, rows = set(col_indices.tolist()), set(row_indices.tolist())
    for col, detection_idx in enumerate(detection_indices):
        if col not in cols:
            unmatched_detections.append(detection_idx)
    for row, track_idx in enumerate(track_indices):
        if row not in rows:
            unmatched_tracks.append(track_idx)
    for row, col in zip(row_indices, col_indices):
        track_idx, detection_idx = track_indices[row], detection_indices[col]
        if cost_matrix[row, col] > max_distance:
            unmatched_tracks.append(track_idx)
            unmatched_detections.append(detection_idx)
        else:
            matches.append((track_idx, detection_idx))
    return matches, unmatched_tracks, unmatched_detections


def matching_cascade(distance_metric, max_distance, cascade_depth, tracks, detections, track_indices=None, detection_indices=None):
    """deep_sort/sort/linear_assignment.py:76-142: the reference's cascade is flat (one min_cost_matching over all given tracks)."""
    if track_indices is None:
        track_indices = list(range(len(tracks)))
    if detection_indices is None:
        detection_indices = list(range(len(detections)))
    matches, _, unmatched_detections = min_cost_matching(distance_metric, max_distance, tracks, detections, track_indices, detection_indices)
    unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))
    return matches, unmatched_tracks, unmatched_detections


def gate_cost_matrix(kf, cost_matrix, tracks, detections, track_indices, detection_indices, gated_cost=INFTY_COST, only_position=False):
    """deep_sort/sort/linear_assignment.py:147-203"""
    gating_threshold = chi2inv95[2 if only_position else 4]
    meas = np.stack([detections[i].to_xyah() for i in detection_indices], 0)
    means = np.concatenate([tracks[i].mean for i in track_indices], 0)
    covs = np.concatenate([tracks[i].covariance for i in track_indices], 0)
    gd = kf.gating_distance(means, covs, meas, only_position)
    cost_matrix[gd > gating_threshold] = gated_cost
    return cost_matrix


# ------------------------------------------------------------------------------------------ tracker.py
class Tracker:
    """deep_sort/sort/tracker.py:8-176 on the device-resident tracker (yds_tracker_*): ``predict()`` then
    ``update(detections)`` per frame, ``tracks`` as Track snapshots.  On the device predict + update run as one fused launch
    sequence inside update(); ``tracks`` read BETWEEN predict() and update() show the predicted states (the Kalman
    prediction of the snapshot plus track.py:115-116's age / time_since_update increments), like the reference's list does.
    Calling update() without predict() is refused (the reference would associate un-predicted states).  The appearance
    galleries live in HBM (``metric.samples`` of the metric object handed in is not populated by this class)."""

    def __init__(self, metric, max_iou_distance=0.7, max_age=70, n_init=3, use_cuda=False):
        from .deep_sort import _TrackerHandle
        self.metric = metric
        self.max_iou_distance, self.max_age, self.n_init = max_iou_distance, max_age, n_init
        _lib.init()
        self._handle = _TrackerHandle(metric.matching_threshold, max_iou_distance, max_age, n_init, metric.budget, metric.metric_name)
        self._predicted = False
        self.kf = KalmanFilter()

    def predict(self):
        self._predicted = True

    def update(self, detections):
        if not self._predicted:
            raise RuntimeError("Tracker.update() without Tracker.predict(): call predict() once per frame first (tracker.py:95-113)")
        self._predicted = False
        d = len(detections)
        tlwh = np.stack([x.tlwh for x in detections], 0) if d else np.zeros((0, 4), np.float32)
        feats = np.stack([x.feature for x in detections], 0) if d else np.zeros((0, 512), np.float32)
        payload = np.array([0.0 if x.payload is None else float(x.payload) for x in detections], np.float32)
        if d and feats.shape[1] != 512:
            raise ValueError("the device tracker stores 512-wide embeddings (deep_sort/deep/model.py:93), got %d" % feats.shape[1])
        self.last_rows = self._handle.step(tlwh, feats, payload)

    @property
    def tracks(self):
        st = self._handle.state()
        n = len(st["ids"])
        pay, age = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32)
        if n:
            _lib.check(_lib.load().yds_tracker_get_payload(self._handle._h, _lib.ptr(pay), n))
            _lib.check(_lib.load().yds_tracker_get_age(self._handle._h, _lib.ptr(age), n))
        mean, cov, tsu = st["mean"], st["cov"], st["tsu"]
        if self._predicted and n:                     # between predict() and update(): track.py:105-116
            mean, cov = self.kf.predict(mean, cov)
            age, tsu = age + 1, tsu + 1
        return [Track(mean[i:i + 1], cov[i:i + 1], st["ids"][i], self.n_init, self.max_age, payload=float(pay[i]),
                      hits=st["hits"][i], age=age[i], time_since_update=tsu[i], state=st["state"][i]) for i in range(n)]
