"""cv2.resize(..., interpolation=cv2.INTER_LINEAR) for 8-bit images, restated (TEST ORACLE).

Call sites in the reference: yolo3/detect/img_detect.py:70 (frame -> model input) and
deep_sort/deep/feature_extractor.py:45 (crop -> 64x128).  cv2 is a third-party dependency
(``opencv-python >= 4.1``, README.md:10, unpinned) that is not installed in this image and
whose source is not under /root/reference, so this file restates OpenCV's published
algorithm (modules/imgproc/src/resize.cpp, identical in 3.4.x and 4.x):

* ``scale = 1. / (double(dst) / src)``; per destination index
  ``f = float((d + 0.5) * scale - 0.5)`` (double arithmetic, one rounding to float),
  ``s = floor(f)``, ``f -= s`` (float); ``s < 0 -> s = 0, f = 0``;
  ``s >= src - 1 -> s = src - 1, f = 0`` (x axis only: the y axis clips the two ROW
  indices ``s, s + 1`` into ``[0, src - 1]`` and keeps its weights);
* 11-bit fixed-point weights ``saturate_cast<short>((1.f - f) * 2048), saturate_cast<short>(f * 2048)``
  (cvRound: round half to even);
* horizontal pass in int: ``h = S[s] * a0 + S[s + 1] * a1`` (HResizeLinear<uchar, int, short, 2048>);
* vertical pass (VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>, written so that the
  SIMD ``mulhi`` form is bit-identical):
  ``dst = uchar((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2)``;
* exact 2x down-scaling in both axes is routed to INTER_AREA's fast path (``resize()``:
  "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to INTER_LINEAR"):
  ``dst = (a + b + c + d + 2) >> 2`` over the 2x2 block;
* equal sizes: plain copy.

IPP is not used for 8-bit linear resizing unless ``cv::ipp::setUseIPP_NotExact(true)`` (ipp_resize:
"Resize which doesn't match OpenCV exactly"), so this is what ``cv2.resize`` computes by default.
Pinning: no cv2 here, hence no vectors from cv2 itself ("parity unpinned" for this third-party step);
``tests/test_oracle_resize.py`` holds this vectorised form to an independent scalar transcription of
the same published loops and to a hand-computed table.
"""

import numpy as np

F32 = np.float32
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis(dst, src, clamp_weights):
    """(s, a0, a1): source index and the two 11-bit weights per destination index (cv::resize's xofs/ialpha, yofs/ibeta)."""
    scale = 1.0 / (float(dst) / float(src))                       # double, like hal::resize
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(F32)                      # double arithmetic, then (float)
    s = np.floor(f).astype(np.int64)                               # cvFloor
    f = (f - s.astype(F32)).astype(F32)
    if clamp_weights:                                              # x axis
        lo = s < 0
        s[lo] = 0
        f[lo] = 0
        hi = s >= src - 1
        s[hi] = src - 1
        f[hi] = 0
    a0 = np.clip(np.rint(((F32(1.0) - f).astype(F32) * F32(COEF_SCALE)).astype(F32)), -32768, 32767).astype(np.int64)
    a1 = np.clip(np.rint((f * F32(COEF_SCALE)).astype(F32)), -32768, 32767).astype(np.int64)
    return s, a0, a1


def resize_bilinear_u8(img, size):
    """img uint8 [H,W,C]; size=(dst_w, dst_h) like cv2.  Returns uint8 [dst_h,dst_w,C]."""
    dst_w, dst_h = int(size[0]), int(size[1])
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W = img.shape[:2]
    if (dst_w, dst_h) == (W, H):
        return img.copy()
    if W == 2 * dst_w and H == 2 * dst_h:                          # INTER_AREA fast path
        v = img.astype(np.int64)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _axis(dst_w, W, True)
    sy, b0, b1 = _axis(dst_h, H, False)
    x1 = np.minimum(sx + 1, W - 1)                                 # weight is 0 wherever this clamps
    y0 = np.clip(sy, 0, H - 1)
    y1 = np.clip(sy + 1, 0, H - 1)
    v = img.astype(np.int64)
    hrow = v[:, sx] * a0[None, :, None] + v[:, x1] * a1[None, :, None]          # [H, dst_w, C] int
    h0, h1 = hrow[y0], hrow[y1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return (out & 0xFF).astype(np.uint8)                           # uchar(...) cast; values never leave [0, 255]
