"""Whole hot path on the CPU oracle (TEST ORACLE): the loop body of VideoDetector.detect,
reference yolo3/detect/video_detect.py:134-157, over a list of frames."""

import numpy as np

from . import nms as onms, reid as oreid, tracker as otrk
from .resize import resize_bilinear_u8

F32 = np.float32


def make_injector(net, rows, logit=6.0):
    """Head-logit override equal to csrc/layers.hip inject_*_kernel (bench-only)."""
    yolo_idx = [i for i, d in enumerate(net.module_defs) if d["type"] == "yolo"]

    def inject(layer, head):
        hi = yolo_idx.index(layer)
        A = len(net.params[layer]["anchors"])
        attrs = net.params[layer]["classes"] + 5
        v = head.reshape(head.shape[0], A, attrs, head.shape[2], head.shape[3])
        v[:, :, 4] = -logit
        for r in rows[rows[:, 0] == hi]:
            a, gy, gx, cls = int(r[1]), int(r[2]), int(r[3]), int(r[8])
            cell = np.full(attrs, -logit, F32)
            cell[:4] = r[4:8]
            cell[4] = logit
            cell[5 + cls] = logit
            v[0, a, :, gy, gx] = cell
        return head
    return inject


def run_stream(net, reid_sd, ds_params, frames, inj_rows=None, conf=0.5, nms_thres=0.4, class_mask=(0, 2, 4), reid_fn=None):
    """Returns one entry per frame: int32 [m,6] rows, [] (tracker found nothing) or None (detector None).
    net: DarknetOracle or fast.DarknetFast; reid_fn (optional): callable crops -> features (fast.ReidFast)."""
    S = net.img_size
    trk = otrk.TrackerOracle(**ds_params)
    if reid_fn is None:
        def reid_fn(crops):
            return oreid.reid_forward(crops, reid_sd)
    outs = []
    for t, frame in enumerate(frames):
        x = resize_bilinear_u8(frame, (S[1], S[0])).astype(F32).transpose(2, 0, 1)[None] / F32(255.)
        inject = make_injector(net, inj_rows[t]) if inj_rows is not None else None
        pred = net.forward(x, inject=inject)
        det = onms.soft_non_max_suppression(pred, conf, nms_thres)[0]
        if det is None:
            outs.append(None)
            continue
        det = onms.resize_boxes(det, S, frame.shape[:2])
        keep = np.zeros(len(det), bool)
        for m in class_mask:
            keep |= det[:, 5] == m
        if class_mask is None or len(class_mask) == 0:
            keep[:] = True
        tlwh = onms.p1p2_to_xywh(det[keep, :4])
        feats = reid_fn(oreid.preprocess_crops(frame, tlwh)) if len(tlwh) else np.zeros((0, 512), F32)
        outs.append(trk.update(tlwh, feats, det[keep, 5]))
    return outs
