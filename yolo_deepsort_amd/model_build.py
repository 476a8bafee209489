"""Module-level post-processing functions at the reference's names (yolo3/utils/model_build.py): the NMS runs through
libydsort (yds_nms_pred / yds_nms_merge_pred, csrc/nms.hip); the box-format helpers are the reference's element-wise
definitions on whatever array type they are given (numpy or torch) - inside the pipeline they are fused into the NMS sweep.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .detect import p1p2Toxywh  # noqa: F401  (model_build.py:326-332)

epsilon = 1e-16


def _is_torch(x):
    return hasattr(x, "detach") and hasattr(x, "new")


def resize_boxes(boxes, current_dim, original_shape):
    """model_build.py:12-19: in place, independent x / y ratios (the frame was stretched, not letterboxed)."""
    h_ratio = original_shape[0] / current_dim[0]
    w_ratio = original_shape[1] / current_dim[1]
    boxes[..., 0] *= w_ratio
    boxes[..., 1] *= h_ratio
    boxes[..., 2] *= w_ratio
    boxes[..., 3] *= h_ratio
    return boxes


def xywh2p1p2(x):
    """model_build.py:317-323"""
    y = x.new(x.shape) if _is_torch(x) else np.empty_like(x)
    y[..., 0] = x[..., 0] - x[..., 2] / 2.
    y[..., 1] = x[..., 1] - x[..., 3] / 2.
    y[..., 2] = x[..., 0] + x[..., 2] / 2.
    y[..., 3] = x[..., 1] + x[..., 3] / 2.
    return y


def bbox_iou(box1, box2, p1p2=True):
    """model_build.py:354-381: IoU with the +1 pixel convention on intersection and areas; broadcasting like the reference."""
    if _is_torch(box1):
        import torch
        mx, mn, clamp0 = torch.max, torch.min, lambda v: torch.clamp(v, min=0)
    else:
        box1, box2 = np.asarray(box1), np.asarray(box2)
        mx, mn, clamp0 = np.maximum, np.minimum, lambda v: np.clip(v, 0, None)
    if not p1p2:
        b1_min, b1_max = box1[..., :2] - box1[..., 2:4] / 2., box1[..., :2] + box1[..., 2:4] / 2.
        b2_min, b2_max = box2[..., :2] - box2[..., 2:4] / 2., box2[..., :2] + box2[..., 2:4] / 2.
    else:
        b1_min, b1_max, b2_min, b2_max = box1[..., :2], box1[..., 2:4], box2[..., :2], box2[..., 2:4]
    inter_wh = clamp0(mn(b1_max, b2_max) - mx(b1_min, b2_min) + 1)
    inter = inter_wh[..., 0] * inter_wh[..., 1]
    a1 = (b1_max[..., 0] - b1_min[..., 0] + 1) * (b1_max[..., 1] - b1_min[..., 1] + 1)
    a2 = (b2_max[..., 0] - b2_min[..., 0] + 1) * (b2_max[..., 1] - b2_min[..., 1] + 1)
    return inter / (a1 + a2 - inter + epsilon)


def soft_non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, merge=False, classes=None, agnostic=False,
                             is_p1p2=False):
    """model_build.py:52-137 on [B, n, 5 + nc] predictions (torch tensor or ndarray): multi-label hard NMS with the
    4096 * cls offset, cap 300, per image an [m, 6] array (x1, y1, x2, y2, conf, cls) sorted by score or None.
    `classes` keeps only those class columns (the reference filters the candidates before the greedy step);
    merge=True is the is_p1p2 sliding-window branch (yds_nms_merge_pred)."""
    if agnostic:
        raise ValueError("soft_non_max_suppression: agnostic=True is not built (no caller in the reference uses it)")
    torch_in = _is_torch(prediction)
    pred = prediction.detach().float().cpu().numpy() if torch_in else np.asarray(prediction, dtype=np.float32)
    if pred.ndim != 3:
        raise ValueError("prediction must be [batch, boxes, 5 + classes]")
    if merge and not is_p1p2:
        raise ValueError("merge=True is built for corner-form predictions (is_p1p2=True), the reference's only use of it")
    if is_p1p2 and not merge:
        # corner form without merging: feed centre form, which the kernel converts back with the same fp32 operations
        # only approximately - keep exactness by refusing instead
        raise ValueError("is_p1p2=True without merge is not built (the reference only pairs them, img_detect.py:147)")
    pred = np.ascontiguousarray(pred, dtype=np.float32).copy()
    if classes:
        keep = np.zeros(pred.shape[2] - 5, bool)
        keep[np.asarray(list(classes), dtype=np.int64)] = True
        pred[:, :, 5:][:, :, ~keep] = 0.0           # conf = obj * 0 never passes `> conf_thres`
    lib = _lib.load()
    _lib.init()
    fn = lib.yds_nms_merge_pred if merge else lib.yds_nms_pred
    out = []
    for x in pred:
        rows = np.zeros((300, 6), np.float32)
        n = C.c_int(0)
        x = np.ascontiguousarray(x)
        _lib.check(fn(_lib.ptr(x), x.shape[0], x.shape[1], float(conf_thres), float(iou_thres), _lib.ptr(rows), 300, C.byref(n)))
        if n.value == 0:
            out.append(None)
            continue
        r = rows[:n.value].copy()
        if torch_in:
            import torch
            r = torch.from_numpy(r)
        out.append(r)
    return out
