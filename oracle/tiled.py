"""numpy restatement of the tiled sliding-window branch of ImageDetector.detect (TEST ORACLE).

Reference yolo3/detect/img_detect.py:97-151: windows of win_size (+ overlap) are cut on a win_size grid, each is
stretched to the model size, the batch goes through the model, boxes become corner form, are scaled back to the
window's own size, shifted by the window origin, concatenated over windows and passed through
soft_non_max_suppression(merge=True, is_p1p2=True)."""

import numpy as np

from . import nms as onms
from .resize import resize_bilinear_u8

F32 = np.float32


def windows(h, w, win_size, overlap):
    win_w, win_h = win_size
    ox, oy = int(win_w * overlap), int(win_h * overlap)
    out = []
    for x in range(0, w, win_w):
        for y in range(0, h, win_h):
            out.append((x, y, min(y + win_h + oy, h) - y, min(x + win_w + ox, w) - x))     # x, y, tile_h, tile_w
    return out


def detect_tiled(net, img, win_size, overlap, thres, nms_thres):
    """net: DarknetOracle; img uint8 [H,W,3].  Returns [n,6] fp32 or None."""
    h, w, _ = img.shape
    S = net.img_size
    tiles = windows(h, w, win_size, overlap)
    batch = np.stack([resize_bilinear_u8(img[y:y + th, x:x + tw], (S[1], S[0])) for x, y, th, tw in tiles], 0)
    pred = net.forward(batch.astype(F32).transpose(0, 3, 1, 2) / F32(255.))
    pred[..., :4] = onms.xywh2p1p2(pred[..., :4])
    for b, (x, y, th, tw) in enumerate(tiles):
        onms.resize_boxes(pred[b], S, (th, tw))
        pred[b, :, :4] += np.array([x, y, x, y], dtype=F32)
    merged = pred.reshape(1, -1, pred.shape[-1])
    return onms.soft_non_max_suppression_merge(merged, thres, nms_thres, is_p1p2=True)[0]
