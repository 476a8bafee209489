"""Every implicit-GEMM conv kernel variant (through the C ABI entry yds_conv_run) vs a float64 numpy convolution.

The detector tests only exercise the variant the autotuner picks per layer; this file pins each tile shape / staging
scheme on its own: 3x3 s1, 3x3 s2, 1x1, ragged M and Cout edges, every activation, both residual modes.
Tolerance: 1e-3 of the output scale (north_star) - the kernels sit at ~1e-6."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32
ACT = {"linear": 0, "leaky": 1, "mish": 2, "relu": 3}


def _act(v, act):
    if act == 1:
        return np.where(v > 0, v, v * 0.1)
    if act == 2:
        sp = np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20))))
        return v * np.tanh(sp)
    if act == 3:
        return np.maximum(v, 0)
    return v


def _conv_ref(x, w, bias, k, s, act, res, res_mode):
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (wd + 2 * pad - k) // s + 1
    xp = np.zeros((n, h + 2 * pad, wd + 2 * pad, cin), np.float64)
    xp[:, pad:pad + h, pad:pad + wd] = x
    y = np.zeros((n, ho, wo, cout), np.float64)
    w64 = w.astype(np.float64).reshape(cout, k, k, cin)
    for kh in range(k):
        for kw in range(k):
            patch = xp[:, kh:kh + s * ho:s, kw:kw + s * wo:s]
            y += patch @ w64[:, kh, kw].T
    y += bias.astype(np.float64)
    if res_mode == 2:
        y += res
    y = _act(y, act)
    if res_mode == 1:
        y += res
    return y.transpose(0, 3, 1, 2)


def _run(L, variant, x, w, bias, k, s, act, res, res_mode):
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (wd + 2 * pad - k) // s + 1
    y = np.empty((n, cout, ho, wo), F32)
    L.check(L.load().yds_conv_run(variant, n, h, wd, cin, cout, k, s, act, res_mode, L.ptr(x), L.ptr(w), L.ptr(bias),
                                  L.ptr(res) if res is not None else None, L.ptr(y)))
    return y


CASES = [  # n, h, w, cin, cout, k, s, act, res_mode
    (2, 19, 19, 64, 128, 3, 1, "leaky", 0),
    (1, 38, 38, 128, 256, 3, 1, "leaky", 1),          # fused shortcut
    (3, 20, 12, 96, 160, 3, 2, "mish", 0),            # stride 2, ragged M / Cout tiles
    (2, 26, 26, 256, 255, 1, 1, "linear", 0),         # head: fp32 output, Cout % 4 != 0
    (1, 16, 8, 64, 64, 3, 1, "relu", 2),              # ReID basic block (residual before the activation)
    (5, 13, 13, 32, 512, 1, 1, "leaky", 0),
    (1, 40, 40, 32, 64, 3, 1, "mish", 1),
    # 3x3 stride 1 at the detector's widths: tiles cross image boundaries, ragged tails, 1 / 2 / 4 channel groups
    (3, 19, 19, 128, 128, 3, 1, "leaky", 0),
    (2, 76, 76, 64, 128, 3, 1, "leaky", 1),
    (5, 13, 13, 32, 96, 3, 1, "mish", 0),
    (7, 8, 4, 256, 256, 3, 1, "relu", 2),
    (1, 12, 304, 32, 64, 3, 1, "leaky", 1),           # one channel group: single window buffer, wide image
    (2, 38, 38, 64, 256, 3, 1, "mish", 1),            # two N tiles, four half groups, 2 x 1444 pixels: tiles straddle the image boundary
    (1, 120, 127, 32, 128, 3, 1, "leaky", 0),         # widest image the two-workgroup window kernel takes (384 window rows)
    (3, 64, 32, 64, 64, 3, 1, "relu", 0),             # ReID layer1 shape: the 128x64 tile (two workgroups per CU), tiles cross crops
    (2, 30, 43, 96, 48, 3, 1, "leaky", 1),            # its widest image (216 window rows), three channel groups, ragged filters
]


@pytest.mark.parametrize("math", [1, 0])
def test_every_conv_variant_vs_float64(math):
    from yolo_deepsort_amd import _lib as L
    L.init(0)
    lib = L.load()
    lib.yds_conv_variant_name.restype = C.c_char_p
    prev = lib.yds_get_conv_math()
    L.check(lib.yds_set_conv_math(math))
    try:
        nvar = lib.yds_conv_num_variants()
        names = [lib.yds_conv_variant_name(v).decode() for v in range(nvar)]
        mine = [v for v, nme in enumerate(names) if ("f16x3" in nme) == bool(math) and "direct" not in nme]
        assert len(mine) >= 4
        rng = np.random.RandomState(17)
        worst = {}
        for n, h, wd, cin, cout, k, s, act, res_mode in CASES:
            x = rng.standard_normal((n, h, wd, cin)).astype(F32)
            w = (rng.standard_normal((cout, k * k * cin)) / np.sqrt(k * k * cin)).astype(F32)
            bias = rng.standard_normal(cout).astype(F32)
            pad = (k - 1) // 2
            ho, wo = (h + 2 * pad - k) // s + 1, (wd + 2 * pad - k) // s + 1
            res = rng.standard_normal((n, ho, wo, cout)).astype(F32) if res_mode else None
            want = _conv_ref(x, w, bias, k, s, ACT[act], res, res_mode)
            scale = float(np.abs(want).max())
            for v in mine:
                try:
                    got = _run(L, v, x, w, bias, k, s, ACT[act], res, res_mode)
                except L.YdsError as e:
                    # the window-resident kernels only take 3x3 stride-1 layers (the two-workgroup form: W <= 127 as well);
                    # they must say so, not compute garbage
                    if "splitK" in names[v]:              # split-K refuses layers that already have enough tiles / too few K steps
                        assert "split-K does not apply" in str(e), (names[v], str(e))
                        continue
                    assert "window-resident" in str(e), (names[v], str(e))
                    small = "win<128,64" in names[v] and (cout > 64 or wd > 43)      # the 128x64 tile: 64-filter layers, W <= 43
                    assert not (k == 3 and s == 1) or ("win2" in names[v] and wd > 127) or small, (names[v], str(e))
                    continue
                err = float(np.abs(got - want).max()) / scale
                worst[names[v]] = max(worst.get(names[v], 0.0), err)
                assert err < 1e-3, (names[v], (n, h, wd, cin, cout, k, s, act, res_mode), err)
        print({k: f"{v:.1e}" for k, v in worst.items()})
        if math:
            assert any("splitK" in k for k in worst) and any("win2" in k for k in worst) and any("win<128,64" in k for k in worst), worst.keys()   # these kernels really ran
    finally:
        lib.yds_set_conv_math(prev)


def test_direct_rgb_kernel_vs_float64():
    from yolo_deepsort_amd import _lib as L
    L.init(0)
    lib = L.load()
    lib.yds_conv_variant_name.restype = C.c_char_p
    direct = [v for v in range(lib.yds_conv_num_variants()) if b"direct" in lib.yds_conv_variant_name(v)]
    assert len(direct) == 1
    rng = np.random.RandomState(3)
    for cout, act in ((32, "leaky"), (64, "relu"), (32, "mish")):
        x = np.zeros((2, 33, 47, 4), F32)
        x[..., :3] = rng.uniform(0, 1, (2, 33, 47, 3))
        w = (rng.standard_normal((cout, 9 * 4)) / 5).astype(F32)
        bias = rng.standard_normal(cout).astype(F32)
        want = _conv_ref(x, w, bias, 3, 1, ACT[act], None, 0)
        got = _run(L, direct[0], x, w, bias, 3, 1, ACT[act], None, 0)
        assert float(np.abs(got - want).max()) / float(np.abs(want).max()) < 1e-5


def test_h16_range_saturates():
    """The H16 activation format holds |x| <= 65504 * 256 = 1.677e7 (csrc/h16.h).  Beyond it an encoded value SATURATES (every
    encoding kernel sets MODE.FP16_OVFL) instead of turning into inf / NaN: outputs of a layer whose pre-images exceed the range
    are finite, clamped to +-(65504 .. 65536) * 256 with the right sign, and every in-range output keeps its accuracy - for each
    f16x3 variant that takes the layer, and for an out-of-range INPUT tensor as well (the packing kernel clamps it)."""
    from yolo_deepsort_amd import _lib as L
    L.init(0)
    lib = L.load()
    lib.yds_conv_variant_name.restype = C.c_char_p
    prev = lib.yds_get_conv_math()
    L.check(lib.yds_set_conv_math(1))
    LIM, TOP = 65504.0 * 256, 65536.0 * 256
    try:
        names = [lib.yds_conv_variant_name(v).decode() for v in range(lib.yds_conv_num_variants())]
        mine = [v for v, nme in enumerate(names) if "f16x3" in nme and "direct" not in nme]
        rng = np.random.RandomState(23)
        ran = 0
        for n, h, wd, cin, cout, k in ((2, 19, 19, 64, 128, 3), (1, 26, 26, 128, 64, 1)):
            x = (rng.standard_normal((n, h, wd, cin)) * 1e4).astype(F32)
            w = (rng.standard_normal((cout, k * k * cin)) / np.sqrt(k * k * cin) * 1e3).astype(F32)
            bias = np.zeros(cout, F32)
            want = _conv_ref(x, w, bias, k, 1, 0, None, 0)
            over, inside = np.abs(want) > 1.01 * TOP, np.abs(want) < 0.99 * LIM
            assert over.mean() > 0.02 and inside.mean() > 0.5
            for v in mine:
                try:
                    got = _run(L, v, x, w, bias, k, 1, 0, None, 0)
                except L.YdsError:
                    continue
                ran += 1
                assert np.isfinite(got).all(), names[v]
                assert np.abs(got[inside] - want[inside]).max() < 1e-3 * LIM, names[v]
                assert (np.abs(got[over]) >= LIM).all() and (np.abs(got[over]) <= TOP).all(), names[v]
                assert np.array_equal(np.sign(got[over]), np.sign(want[over])), names[v]
        assert ran >= 6
        # an input beyond the range: clamped by the packing kernel, the layer's result stays finite
        x = np.full((1, 8, 8, 32), 3e7, F32)
        w = np.zeros((32, 32), F32)
        w[np.arange(32), np.arange(32)] = 1
        got = _run(L, mine[0], x, w, np.zeros(32, F32), 1, 1, 0, None, 0)
        assert np.isfinite(got).all() and (got >= LIM).all() and (got <= TOP).all()
    finally:
        lib.yds_set_conv_math(prev)
