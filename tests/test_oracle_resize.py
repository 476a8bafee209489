"""oracle/resize.py against (a) a hand-computed table from OpenCV's published fixed-point formulas and
(b) an independent scalar transcription of the same loops (xofs/ialpha tables, row buffers, HResize/VResize)."""
import math

import numpy as np

from oracle.resize import resize_bilinear_u8


def _cv_round(v):
    """cvRound on a float32 value: round half to even."""
    return int(np.rint(np.float32(v)))


def _scalar_cv_resize(src, dw, dh):
    """Straight-line transcription of cv::resize / resizeGeneric_ for CV_8UC3 INTER_LINEAR (no vectorisation)."""
    H, W, cn = src.shape
    if (dw, dh) == (W, H):
        return src.copy()
    inv_x, inv_y = float(dw) / W, float(dh) / H
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if abs(scale_x - 2) < 2.3e-16 and abs(scale_y - 2) < 2.3e-16:
        dst = np.zeros((dh, dw, cn), np.uint8)
        for y in range(dh):
            for x in range(dw):
                for c in range(cn):
                    dst[y, x, c] = (int(src[2 * y, 2 * x, c]) + int(src[2 * y, 2 * x + 1, c]) + int(src[2 * y + 1, 2 * x, c]) + int(src[2 * y + 1, 2 * x + 1, c]) + 2) >> 2
        return dst
    xofs, ialpha, xmax = [0] * dw, [(0, 0)] * dw, dw
    for dx in range(dw):
        fx = np.float32((dx + 0.5) * scale_x - 0.5)
        sx = math.floor(float(fx))
        fx = np.float32(fx - np.float32(sx))
        if sx < 0:
            fx, sx = np.float32(0), 0
        if sx + 1 >= W:
            xmax = min(xmax, dx)
            if sx >= W - 1:
                fx, sx = np.float32(0), W - 1
        xofs[dx] = sx
        ialpha[dx] = (_cv_round((np.float32(1) - fx) * np.float32(2048)), _cv_round(fx * np.float32(2048)))
    dst = np.zeros((dh, dw, cn), np.uint8)
    for dy in range(dh):
        fy = np.float32((dy + 0.5) * scale_y - 0.5)
        sy = math.floor(float(fy))
        fy = np.float32(fy - np.float32(sy))
        b0, b1 = _cv_round((np.float32(1) - fy) * np.float32(2048)), _cv_round(fy * np.float32(2048))
        rows = []
        for k in range(2):
            r = sy + k
            r = 0 if r < 0 else (r if r < H else H - 1)            # clip(sy - ksize2 + 1 + k, 0, ssize.height)
            buf = np.zeros((dw, cn), np.int64)
            for dx in range(dw):
                for c in range(cn):
                    if dx < xmax:
                        buf[dx, c] = int(src[r, xofs[dx], c]) * ialpha[dx][0] + int(src[r, xofs[dx] + 1, c]) * ialpha[dx][1]
                    else:
                        buf[dx, c] = int(src[r, xofs[dx], c]) * 2048
            rows.append(buf)
        for dx in range(dw):
            for c in range(cn):
                dst[dy, dx, c] = ((((b0 * (int(rows[0][dx, c]) >> 4)) >> 16) + ((b1 * (int(rows[1][dx, c]) >> 4)) >> 16) + 2) >> 2) & 0xFF
    return dst


def test_hand_computed_table():
    # one row [0, 255] stretched to 4 columns: weights (2048,0) (1536,512) (512,1536) (2048,0) ->
    # h = 0, 130560, 391680, 522240 -> ((2048 * (h >> 4)) >> 16 + 2) >> 2 = 0, 64, 191, 255
    src = np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8)
    assert resize_bilinear_u8(src, (4, 1))[0, :, 0].tolist() == [0, 64, 191, 255]
    # [0, 1]: 0.25 -> 512 -> ((2048 * 32) >> 16 + 2) >> 2 = 0 ; 0.75 -> 1536 -> (3 + 2) >> 2 = 1
    src = np.array([[[0, 0, 0], [1, 1, 1]]], np.uint8)
    assert resize_bilinear_u8(src, (4, 1))[0, :, 0].tolist() == [0, 0, 1, 1]
    # column [10, 20, 40] down to 2 rows: scale 1.5, fy = 0.25 / 0.75 with rows (0,1) / (1,2):
    # (1536*((10*2048)>>4)>>16) + (512*((20*2048)>>4)>>16) = 30 + 20 -> 52 >> 2 = 13 ; 40 + 120 -> 162 >> 2 = 40... (20*.25+40*.75 = 35)
    src = np.array([[[10] * 3], [[20] * 3], [[40] * 3]], np.uint8)
    got = resize_bilinear_u8(np.repeat(src, 2, axis=1), (2, 2))[:, 0, 0].tolist()
    assert got == [((1536 * 1280 >> 16) + (512 * 2560 >> 16) + 2) >> 2, ((512 * 2560 >> 16) + (1536 * 5120 >> 16) + 2) >> 2] == [13, 35]
    # exact 2x down-scale takes the INTER_AREA fast path: (a + b + c + d + 2) >> 2
    src = np.array([[[1] * 3, [2] * 3], [[2] * 3, [2] * 3]], np.uint8)
    assert resize_bilinear_u8(src, (1, 1))[0, 0, 0] == (1 + 2 + 2 + 2 + 2) >> 2 == 2
    # same size: copy
    assert np.array_equal(resize_bilinear_u8(src, (2, 2)), src)


def test_vectorised_equals_scalar_transcription():
    rng = np.random.RandomState(5)
    cases = [((7, 9), (5, 4)), ((5, 4), (13, 17)), ((1, 1), (6, 3)), ((3, 1), (2, 5)), ((1, 6), (4, 2)), ((12, 8), (6, 4)),
             ((37, 23), (64, 128)), ((20, 31), (9, 7)), ((2, 2), (3, 3)), ((48, 64), (24, 33))]
    for (h, w), (dw, dh) in cases:
        src = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        assert np.array_equal(resize_bilinear_u8(src, (dw, dh)), _scalar_cv_resize(src, dw, dh)), ((h, w), (dw, dh))
    # extreme values and .5 ties
    src = np.array([[[0] * 3, [255] * 3, [1] * 3, [254] * 3]] * 3, np.uint8)
    for dw, dh in ((8, 6), (3, 2), (5, 3), (16, 1)):
        assert np.array_equal(resize_bilinear_u8(src, (dw, dh)), _scalar_cv_resize(src, dw, dh))


def test_bench_geometries_stay_in_range():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (270, 480, 3)).astype(np.uint8)
    out = resize_bilinear_u8(img, (152, 152))
    ref = img.astype(np.float64)
    assert out.shape == (152, 152, 3) and abs(out.astype(np.float64).mean() - ref.mean()) < 1.0
    flat = np.full((11, 13, 3), 200, np.uint8)
    assert (resize_bilinear_u8(flat, (29, 31)) == 200).all()        # weights sum to 2048: constants are preserved


def test_geometry_against_an_independent_float_bilinear():
    """3P-2 stays unpinned against cv2 itself (not in the image), but its GEOMETRY - half-pixel sample positions, edge clamping, which
    two source pixels and what weights - can be held to an independent implementation that IS here: torch's
    F.interpolate(mode="bilinear", align_corners=False, antialias=False) computes the same interpolant in float.  The 8-bit
    fixed-point path (11-bit weights, two roundings) may differ from the rounded float result by one grey level, never by more; on the
    shapes of this path: frame -> detector input (down-scaling, non-integer ratios), crop -> 64 x 128 ReID input (up- and down-scaling),
    an exact 2x reduction (OpenCV's INTER_AREA shortcut = the mean of four, which is also what the float interpolant gives)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.RandomState(12)
    for (h, w), (dw, dh) in (((480, 640), (416, 416)), ((540, 960), (608, 608)), ((1080, 1920), (608, 416)), ((173, 61), (64, 128)),
                             ((40, 23), (64, 128)), ((64, 32), (16, 32)), ((97, 131), (200, 211))):
        # smooth + noisy content: pure noise would hide a half-pixel shift behind its own variance
        base = rng.randint(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
        img = np.kron(base, np.ones((8, 8, 1), np.float32))[:h, :w] * 0.8 + rng.randint(0, 52, (h, w, 3))
        img = np.clip(img, 0, 255).astype(np.uint8)
        got = resize_bilinear_u8(img, (dw, dh)).astype(np.int32)
        t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
        ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        diff = np.abs(got - ref)
        assert got.shape == (dh, dw, 3)
        assert diff.max() <= 1.0 + 1e-3, ((h, w), (dw, dh), float(diff.max()))
        assert diff.mean() < 0.3, ((h, w), (dw, dh), float(diff.mean()))
        # a half-pixel shift of the sampling grid would show as a mean error of several grey levels on this content
        shifted = F.interpolate(torch.roll(t, 1, dims=3), size=(dh, dw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - shifted).mean() > 3 * diff.mean() + 1.0
