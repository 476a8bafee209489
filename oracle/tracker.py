"""numpy restatement of the DeepSORT association stage (TEST ORACLE).

Kalman filter: reference deep_sort/sort/kalman_filter.py:22-256.
Cosine metric + gallery: deep_sort/sort/nn_matching.py:30-53,77-100,139-187.
Gating / min-cost matching: deep_sort/sort/linear_assignment.py:8-73,147-203.
IOU cost: deep_sort/sort/iou_matching.py:5-91.
Track lifecycle: deep_sort/sort/track.py:63-152, deep_sort/sort/tracker.py:49-176.
Output stage: deep_sort/deep_sort.py:63-88,108-114.
LSAP: scipy.optimize.linear_sum_assignment (third party) restated in
oracle/csrc/oracle.c and pinned against real scipy in tests.
"""

import numpy as np

from . import clib

F32 = np.float32
INFTY_COST = F32(1e5)
CHI2_2DOF = 5.9915
TENTATIVE, CONFIRMED, DELETED = 1, 2, 3

_MOTION_T = np.eye(8, dtype=F32)
for _i in range(4):
    _MOTION_T[_i, 4 + _i] = 1
_MOTION_T = np.ascontiguousarray(_MOTION_T.T)           # stored transposed, kalman_filter.py:27-31
_STD_POS = np.array([[1. / 20, 1. / 20, 0, 1. / 20]], dtype=F32)
_STD_VEL = np.array([[1. / 160, 1. / 160, 0, 1. / 160]], dtype=F32)


def kf_initiate(xyah):
    """kalman_filter.py:54-87  (python-double coefficient times fp32 h, rounded to fp32)."""
    xyah = np.asarray(xyah, dtype=F32)
    mean = np.concatenate([xyah, np.zeros(4, F32)]).reshape(1, 8)
    h = xyah[3]
    c_pos = 2 * (1. / 20)
    c_vel = 10 * (1. / 160)
    sp, sv = F32(c_pos) * h, F32(c_vel) * h               # scalar cast to fp32, fp32 multiply
    std = np.array([[sp, sp, 1e-2, sp, sv, sv, 1e-5, sv]], dtype=F32)
    cov = np.zeros((1, 8, 8), F32)
    cov[0][np.arange(8), np.arange(8)] = (std[0] * std[0]).astype(F32)
    return mean, cov


def kf_predict(mean, cov):
    """kalman_filter.py:89-123; mean [T,8], cov [T,8,8]."""
    std_pos = (mean[:, 3:4] * _STD_POS).astype(F32)
    std_vel = (mean[:, 3:4] * _STD_VEL).astype(F32)
    std_pos[:, 2] = 1e-2
    std_vel[:, 2] = 1e-5
    d = np.concatenate([std_pos, std_vel], -1)
    q = (d * d).astype(F32)
    new_mean = (mean @ _MOTION_T).astype(F32)
    new_cov = np.matmul(np.matmul(cov.transpose(0, 2, 1), _MOTION_T).transpose(0, 2, 1), _MOTION_T).astype(F32)
    idx = np.arange(8)
    new_cov[:, idx, idx] += q
    return new_mean, new_cov


def kf_project(mean, cov):
    """kalman_filter.py:125-159"""
    std = (mean[:, 3:4] * _STD_POS).astype(F32)
    std[:, 2] = 1e-1
    s = cov[:, :4, :4].copy()
    idx = np.arange(4)
    s[:, idx, idx] += (std * std).astype(F32)
    return mean[:, :4].copy(), s


def kf_update(mean, cov, z):
    """kalman_filter.py:161-204; z [M,4] xyah."""
    pm, pc = kf_project(mean, cov)
    rhs = cov[:, :, :4].transpose(0, 2, 1)                       # (P H)^T  [M,4,8]
    kt = np.linalg.solve(pc.astype(F32), rhs.astype(F32)).astype(F32)   # K^T [M,4,8]
    innov = (z.reshape(-1, 4) - pm).astype(F32)
    new_mean = (mean + np.matmul(innov[:, None, :], kt).reshape(-1, 8)).astype(F32)
    ks = np.matmul(pc.transpose(0, 2, 1), kt).transpose(0, 2, 1)        # K S
    new_cov = (cov - np.matmul(ks, kt)).astype(F32)
    return new_mean, new_cov


def kf_gating_distance(mean, cov, xyah, only_position=True):
    """kalman_filter.py:206-256 -> [T,D] squared Mahalanobis distance."""
    pm, pc = kf_project(mean, cov)
    n = 2 if only_position else 4
    pm, pc, z = pm[:, None, :n], pc[:, :n, :n], xyah[None, :, :n]
    d = (-pm + z).astype(F32)                                           # [T,D,n]
    inv = np.linalg.inv(pc.astype(F32)).astype(F32)
    m = np.matmul(np.matmul(d, inv), d.transpose(0, 2, 1))
    return np.diagonal(m, axis1=-2, axis2=-1).astype(F32)


def tlwh_to_xyah(tlwh):
    """detection.py:41-48 / linear_assignment.py:186-189 (in-place order kept)."""
    r = np.array(tlwh, dtype=F32, copy=True).reshape(-1, 4)
    r[:, :2] += r[:, 2:] / F32(2)
    r[:, 2] /= r[:, 3]
    return r


def cosine_distance(a, b):
    """nn_matching.py:30-53 (re-normalises both sides)."""
    a = a / np.sqrt((a * a).sum(-1, keepdims=True, dtype=F32))
    b = b / np.sqrt((b * b).sum(-1, keepdims=True, dtype=F32))
    return (F32(1.) - a.astype(F32) @ b.astype(F32).T).astype(F32)


def euclidean_min_distance(samples, feats):
    """nn_matching.py:4-27,56-74: min over `samples` rows of sum((a - b)^2), clamped at 0 -> [len(feats)]."""
    d = ((samples[:, None, :] - feats[None, :, :]).astype(F32) ** 2).sum(-1, dtype=F32)
    return np.maximum(d.min(axis=0), F32(0)).astype(F32)


def tracker_nms(tlwh, max_bbox_overlap, order):
    """deep_sort/sort/preprocessing.py:6-73 with scores given (order = np.argsort(scores), decided by numpy)."""
    if len(tlwh) == 0:
        return []
    boxes = np.asarray(tlwh).astype(np.float64)
    x1, y1 = boxes[:, 0], boxes[:, 1]
    x2, y2 = boxes[:, 2] + boxes[:, 0], boxes[:, 3] + boxes[:, 1]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    idxs = np.asarray(order).copy()
    pick = []
    while len(idxs) > 0:
        last = len(idxs) - 1
        i = idxs[last]
        pick.append(int(i))
        rest = idxs[:last]
        w = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        overlap = (w * h) / area[rest]
        idxs = np.delete(idxs, np.concatenate(([last], np.where(overlap > max_bbox_overlap)[0])))
    return pick


def iou_matrix(bbox, cand):
    """iou_matching.py:5-41 (asymmetric +1 in the intersection only)."""
    b = bbox[:, None, :]
    c = cand[None, :, :]
    bmax = b[..., :2] + b[..., 2:]
    cmax = c[..., 2:] + c[..., :2]
    imin = np.maximum(b[..., :2], c[..., :2])
    imax = np.minimum(bmax, cmax)
    wh = np.maximum((imax - imin + F32(1)).astype(F32), F32(0))
    inter = (wh[..., 0] * wh[..., 1]).astype(F32)
    ba = (b[..., 2] * b[..., 3]).astype(F32)
    ca = (c[..., 2] * c[..., 3]).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / ((ba + ca).astype(F32) - inter).astype(F32)).astype(F32)


class _Track:
    __slots__ = ("mean", "cov", "track_id", "hits", "age", "tsu", "state", "features", "payload")

    def __init__(self, mean, cov, tid, feature, payload):
        self.mean, self.cov, self.track_id = mean, cov, tid
        self.hits, self.age, self.tsu = 1, 1, 0
        self.state = TENTATIVE
        self.features = [feature]
        self.payload = payload

    def to_tlwh(self):
        r = self.mean.reshape(-1)[:4].copy()
        r[2] *= r[3]
        r[:2] -= r[2:] / F32(2)
        return r


def min_cost_matching(cost, max_distance, track_indices, detection_indices):
    """linear_assignment.py:52-73 given the already-built cost matrix [len(ti), len(di)]."""
    md = F32(max_distance)
    cost = cost.astype(F32, copy=True)
    cost[cost > md] = F32(max_distance + 1e-5)
    rows, cols = clib.lsap(cost)
    matches, um_t, um_d = [], [], []
    colset, rowset = set(cols.tolist()), set(rows.tolist())
    for col, d in enumerate(detection_indices):
        if col not in colset:
            um_d.append(d)
    for row, t in enumerate(track_indices):
        if row not in rowset:
            um_t.append(t)
    for r, c in zip(rows, cols):
        if cost[r, c] > md:
            um_t.append(track_indices[r])
            um_d.append(detection_indices[c])
        else:
            matches.append((track_indices[r], detection_indices[c]))
    return matches, um_t, um_d, cost, (rows, cols)


class TrackerOracle:
    """DeepSort.update minus the ReID extractor: takes features directly."""

    def __init__(self, max_dist=0.2, max_iou_distance=0.7, max_age=70, n_init=3, nn_budget=100, metric="cosine", nms_max_overlap=1.0):
        if metric not in ("cosine", "euclidean"):
            raise ValueError("Invalid metric; must be either 'euclidean' or 'cosine'")
        self.metric, self.nms_max_overlap = metric, nms_max_overlap
        self.nms_order = None                      # test hook: the np.argsort(scores) the reference run observed
        self.max_dist, self.max_iou_distance = max_dist, max_iou_distance
        self.max_age, self.n_init, self.budget = max_age, n_init, nn_budget
        self.tracks = []
        self.samples = {}
        self.next_id = 1
        self.debug = {}

    # -- tracker.py:95-113
    def _predict(self):
        if not self.tracks:
            return
        m = np.concatenate([t.mean for t in self.tracks], 0)
        c = np.concatenate([t.cov for t in self.tracks], 0)
        m, c = kf_predict(m, c)
        for i, t in enumerate(self.tracks):
            t.mean, t.cov = m[i:i + 1], c[i:i + 1]
            t.age += 1
            t.tsu += 1

    # -- tracker.py:56-93
    def _match(self, tlwh, feats):
        D = tlwh.shape[0]
        confirmed = [i for i, t in enumerate(self.tracks) if t.state == CONFIRMED]
        unconfirmed = [i for i, t in enumerate(self.tracks) if t.state != CONFIRMED]
        det_idx = list(range(D))
        dbg = self.debug
        if len(det_idx) == 0 or len(confirmed) == 0:
            matches_a, um_t_a, um_d = [], confirmed, det_idx
        else:
            bp, samples = [0], []
            for i in confirmed:
                s = self.samples[self.tracks[i].track_id]
                samples += s
                bp.append(bp[-1] + len(s))
            if self.metric == "euclidean":
                # the reference's distance() hands _nn_euclidean_distance a third argument and raises (nn_matching.py:56,187);
                # the evident intent - the helper applied per track segment - is what is restated
                g = np.stack(samples, 0)
                cost = np.stack([euclidean_min_distance(g[bp[k]:bp[k + 1]], feats[det_idx]) for k in range(len(bp) - 1)], 0)
            else:
                dist = cosine_distance(np.stack(samples, 0), feats[det_idx])
                cost = np.stack([dist[bp[k]:bp[k + 1]].min(axis=0) for k in range(len(bp) - 1)], 0)
            xyah = tlwh_to_xyah(tlwh[det_idx])
            means = np.concatenate([self.tracks[i].mean for i in confirmed], 0)
            covs = np.concatenate([self.tracks[i].cov for i in confirmed], 0)
            gate = kf_gating_distance(means, covs, xyah, True)
            cost[gate > F32(CHI2_2DOF)] = INFTY_COST
            matches_a, um_t_a, um_d, c1, _ = min_cost_matching(cost, self.max_dist, confirmed, det_idx)
            dbg["cost_a"] = c1
        iou_cand = unconfirmed + [k for k in um_t_a if self.tracks[k].tsu == 1]
        um_t_a = [k for k in um_t_a if self.tracks[k].tsu != 1]
        if len(um_d) == 0 or len(iou_cand) == 0:
            matches_b, um_t_b = [], iou_cand
        else:
            cand = np.stack([tlwh[i] for i in um_d], 0)
            bbs = np.stack([self.tracks[i].to_tlwh() for i in iou_cand], 0)
            cost = (F32(1.) - iou_matrix(bbs, cand)).astype(F32)
            for r, k in enumerate(iou_cand):
                if self.tracks[k].tsu > 1:
                    cost[r, :] = INFTY_COST
            matches_b, um_t_b, um_d, c2, _ = min_cost_matching(cost, self.max_iou_distance, iou_cand, um_d)
            dbg["cost_b"] = c2
        return matches_a + matches_b, list(set(um_t_a + um_t_b)), um_d

    def update(self, tlwh, feats, payload):
        tlwh = np.asarray(tlwh, dtype=F32).reshape(-1, 4)
        feats = np.asarray(feats, dtype=F32).reshape(tlwh.shape[0], -1) if tlwh.shape[0] else np.zeros((0, 0), F32)
        payload = np.asarray(payload, dtype=F32).reshape(-1)
        self.debug = {}
        if self.nms_max_overlap != 1:                            # deep_sort.py:52-57 (confidence is 1 for every detection)
            order = self.nms_order if self.nms_order is not None else np.argsort(np.ones(len(tlwh), dtype=np.float64))
            keep = tracker_nms(tlwh, self.nms_max_overlap, order)
            tlwh, feats, payload = tlwh[keep], feats[keep], payload[keep]
            self.debug["nms_keep"] = keep
        self._predict()
        matches, um_t, um_d = self._match(tlwh, feats)
        self.debug.update(matches=list(matches), unmatched_tracks=sorted(um_t), unmatched_detections=list(um_d))
        # -- tracker.py:129-156
        if matches:
            m = np.concatenate([self.tracks[t].mean for t, _ in matches], 0)
            c = np.concatenate([self.tracks[t].cov for t, _ in matches], 0)
            z = tlwh_to_xyah(np.stack([tlwh[d] for _, d in matches], 0))
            m, c = kf_update(m, c, z)
            for i, (t, d) in enumerate(matches):
                tr = self.tracks[t]
                tr.mean, tr.cov = m[i:i + 1], c[i:i + 1]
                tr.features.append(feats[d])
                tr.hits += 1
                tr.tsu = 0
                if tr.state == TENTATIVE and tr.hits >= self.n_init:
                    tr.state = CONFIRMED
                tr.payload = payload[d]
        # -- track.py:146-152
        for t in um_t:
            tr = self.tracks[t]
            if tr.state == TENTATIVE:
                tr.state = DELETED
            elif tr.tsu > self.max_age:
                tr.state = DELETED
        # -- tracker.py:49-54
        for d in um_d:
            mean, cov = kf_initiate(tlwh_to_xyah(tlwh[d])[0])
            self.tracks.append(_Track(mean, cov, self.next_id, feats[d], payload[d]))
            self.next_id += 1
        self.tracks = [t for t in self.tracks if t.state != DELETED]
        # -- tracker.py:164-176 + nn_matching.py:139-156
        active = [t.track_id for t in self.tracks if t.state == CONFIRMED]
        for t in self.tracks:
            if t.state != CONFIRMED:
                continue
            for f in t.features:
                self.samples.setdefault(t.track_id, []).append(f)
                if self.budget is not None:
                    self.samples[t.track_id] = self.samples[t.track_id][-self.budget:]
            t.features = []
        self.samples = {k: self.samples[k] for k in active}
        # -- deep_sort.py:63-88
        rows = []
        for t in self.tracks:
            if t.state != CONFIRMED or t.tsu > 1:
                continue
            b = t.mean[0, :4].copy()
            b[2] *= b[3]
            b[:2] -= b[2:] / F32(2)
            b[2:] += b[:2]
            b[:2] = np.maximum(b[:2], F32(0))
            rows.append([b[0], b[1], b[2], b[3], t.track_id, t.payload])
        if rows:
            return np.array(rows, dtype=np.float64).astype(np.int32)
        return []

    def state(self):
        return dict(ids=[t.track_id for t in self.tracks], state=[t.state for t in self.tracks],
                    tsu=[t.tsu for t in self.tracks], hits=[t.hits for t in self.tracks],
                    mean=np.concatenate([t.mean for t in self.tracks], 0) if self.tracks else np.zeros((0, 8), F32),
                    cov=np.concatenate([t.cov for t in self.tracks], 0) if self.tracks else np.zeros((0, 8, 8), F32))
