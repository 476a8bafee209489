"""Bilinear uint8 resize restatement (TEST ORACLE).

Stands in for cv2.resize(..., interpolation=cv2.INTER_LINEAR) at reference
yolo3/detect/img_detect.py:70 and deep_sort/deep/feature_extractor.py:45.
cv2 is third party and not installed in this image, so this step is
"parity unpinned": the definition below IS the spec both the oracle harness
(the cv2 shim used while importing the reference) and the HIP kernels follow:
half-pixel-centre sampling, source index clamped to the image, fp32 lerp with
every product and sum rounded separately (no FMA), round-half-even to uint8.
"""

import numpy as np

F32 = np.float32


def _axis_coords(dst, src):
    scale = F32(src) / F32(dst)
    f = ((np.arange(dst, dtype=F32) + F32(0.5)) * scale - F32(0.5)).astype(F32)
    i0 = np.floor(f).astype(np.int32)
    frac = (f - i0.astype(F32)).astype(F32)
    lo = i0 < 0
    i0[lo] = 0
    frac[lo] = 0
    hi = i0 >= src - 1
    i0[hi] = src - 1
    frac[hi] = 0
    i1 = np.minimum(i0 + 1, src - 1)
    return i0, i1, frac


def resize_bilinear_u8(img, size):
    """img uint8 [H,W,C]; size=(dst_w, dst_h) like cv2.  Returns uint8 [dst_h,dst_w,C]."""
    dst_w, dst_h = size
    H, W = img.shape[:2]
    x0, x1, fx = _axis_coords(dst_w, W)
    y0, y1, fy = _axis_coords(dst_h, H)
    im = img.astype(F32)
    fx = fx[None, :, None]
    fy = fy[:, None, None]
    one = F32(1)
    top = ((one - fx) * im[y0][:, x0]).astype(F32) + (fx * im[y0][:, x1]).astype(F32)
    bot = ((one - fx) * im[y1][:, x0]).astype(F32) + (fx * im[y1][:, x1]).astype(F32)
    v = ((one - fy) * top.astype(F32)).astype(F32) + (fy * bot.astype(F32)).astype(F32)
    return np.clip(np.rint(v.astype(F32)), 0, 255).astype(np.uint8)
