"""numpy restatement of the detector post-processing (TEST ORACLE).

soft_non_max_suppression: reference yolo3/utils/model_build.py:52-137 (despite
its name it is hard, multi-label NMS).  The greedy kernel itself is
torchvision.ops.boxes.nms (third party, call site model_build.py:119), restated
from its CPU kernel: stable sort by score descending, area=(x2-x1)*(y2-y1),
suppress j when inter/(a_i+a_j-inter) > thr (fp32 IoU, threshold held as double).
resize_boxes: model_build.py:12-19.  p1p2Toxywh: model_build.py:326-332.
"""

import numpy as np

F32 = np.float32


def nms_greedy(boxes, scores, iou_thres):
    n = boxes.shape[0]
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(F32)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = float(iou_thres)
    for a in range(n):
        i = order[a]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(F32(0), (xx2 - xx1).astype(F32)); h = np.maximum(F32(0), (yy2 - yy1).astype(F32))
        inter = (w * h).astype(F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = (inter / ((areas[i] + areas[rest]).astype(F32) - inter).astype(F32)).astype(F32)
        suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return np.array(keep, dtype=np.int64)


def xywh2p1p2(x):
    """model_build.py:317-323"""
    y = np.empty_like(x)
    y[..., 0] = x[..., 0] - x[..., 2] / F32(2.)
    y[..., 1] = x[..., 1] - x[..., 3] / F32(2.)
    y[..., 2] = x[..., 0] + x[..., 2] / F32(2.)
    y[..., 3] = x[..., 1] + x[..., 3] / F32(2.)
    return y


def soft_non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6):
    """prediction [B,N,5+C] fp32 -> list of [n,6] (x1,y1,x2,y2,score,cls) or None."""
    prediction = np.asarray(prediction, dtype=F32)
    ct = F32(conf_thres)                       # torch compares in the tensor dtype
    xc = prediction[..., 4] > ct
    max_wh, max_det = 4096, 300
    output = [None] * prediction.shape[0]
    for xi in range(prediction.shape[0]):
        x = prediction[xi][xc[xi]].copy()
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2p1p2(x[:, :4])
        i, j = np.nonzero(x[:, 5:] > ct)       # row-major, like torch.nonzero
        x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(F32)), 1).astype(F32)
        n = x.shape[0]
        if not n:
            continue
        c = (x[:, 5:6] * F32(max_wh)).astype(F32)
        boxes, scores = (x[:, :4] + c).astype(F32), x[:, 4]
        k = nms_greedy(boxes, scores, iou_thres)
        if k.shape[0] > max_det:
            k = k[:max_det]
        output[xi] = x[k]
    return output


def resize_boxes(boxes, current_dim, original_shape):
    """model_build.py:12-19 (python-double ratio, fp32 multiply, in place)."""
    h_ratio = original_shape[0] / current_dim[0]
    w_ratio = original_shape[1] / current_dim[1]
    boxes[..., 0] *= F32(w_ratio)
    boxes[..., 1] *= F32(h_ratio)
    boxes[..., 2] *= F32(w_ratio)
    boxes[..., 3] *= F32(h_ratio)
    return boxes


def p1p2_to_xywh(x):
    y = np.empty_like(x)
    y[..., 0] = x[..., 0]
    y[..., 1] = x[..., 1]
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def bbox_iou_plus1(b1, b2, eps=1e-16):
    """model_build.py:354-381 (p1p2 form): elementwise IoU with the +1 pixel convention, fp32."""
    imin = np.maximum(b1[..., :2], b2[..., :2])
    imax = np.minimum(b1[..., 2:4], b2[..., 2:4])
    wh = np.maximum((imax - imin + F32(1)).astype(F32), F32(0))
    inter = (wh[..., 0] * wh[..., 1]).astype(F32)
    a1 = ((b1[..., 2] - b1[..., 0] + F32(1)) * (b1[..., 3] - b1[..., 1] + F32(1))).astype(F32)
    a2 = ((b2[..., 2] - b2[..., 0] + F32(1)) * (b2[..., 3] - b2[..., 1] + F32(1))).astype(F32)
    return (inter / ((a1 + a2).astype(F32) - inter + F32(eps)).astype(F32)).astype(F32)


def soft_non_max_suppression_merge(prediction, conf_thres, iou_thres, is_p1p2=True):
    """soft_non_max_suppression(..., merge=True) as the reference actually behaves (model_build.py:122-131).

    `bbox_iou(boxes[i], boxes)` is elementwise, not pairwise: it only broadcasts when the number of kept boxes k is 1
    or equals the number of candidates n; otherwise it raises inside the bare `except` and the plain NMS result
    stands.  When it does broadcast, x[i, :4] is overwritten with ONE weighted-mean box (weights = score where the
    elementwise IoU exceeds the threshold) before `iou.sum(1)` raises - so a lone survivor becomes the mean of its
    cluster and, when nothing was suppressed, every box collapses onto the same mean.  Reproduced, not fixed."""
    prediction = np.asarray(prediction, dtype=F32)
    ct = F32(conf_thres)
    out = [None] * prediction.shape[0]
    for xi in range(prediction.shape[0]):
        x = prediction[xi][prediction[xi][:, 4] > ct].copy()
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = x[:, :4] if is_p1p2 else xywh2p1p2(x[:, :4])
        i, j = np.nonzero(x[:, 5:] > ct)
        x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(F32)), 1).astype(F32)
        n = x.shape[0]
        if not n:
            continue
        c = (x[:, 5:6] * F32(4096)).astype(F32)
        boxes, scores = (x[:, :4] + c).astype(F32), x[:, 4]
        k = nms_greedy(boxes, scores, iou_thres)[:300]
        if 1 < n < 3000 and (len(k) == n or len(k) == 1):
            iou = bbox_iou_plus1(boxes[k], boxes) > F32(iou_thres)            # [n] bools (broadcast of [k,4] with [n,4])
            w = (iou.astype(F32) * scores).astype(F32)[None, :]                  # [1,n]
            with np.errstate(divide="ignore", invalid="ignore"):
                merged = ((w @ x[:, :4]).astype(F32) / w.sum(1, keepdims=True, dtype=F32)).astype(F32)
            x[k, :4] = merged
        out[xi] = x[k]
    return out
