#!/usr/bin/env python3
"""The window kernel's cross8 mode (fp16 hi x hi + fp8 cross terms) on single layers: error against a float64 convolution
with the mode off / on, and launch time off / on at the detector's shapes (tuning aid; run on the GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from yolo_deepsort_amd import _lib as L  # noqa: E402
from test_gpu_conv_variants import ACT, _conv_ref, _run  # noqa: E402

F32 = np.float32


def main():
    L.init(0)
    lib = L.load()
    lib.yds_conv_variant_name.restype = C.c_char_p
    names = [lib.yds_conv_variant_name(v).decode() for v in range(lib.yds_conv_num_variants())]
    win = names.index("conv3x3_f16x3_win<256,128,4x2>")
    rng = np.random.RandomState(5)
    cases = [(2, 19, 19, 64, 128, "leaky", 0), (1, 38, 38, 128, 256, "leaky", 1), (2, 76, 76, 64, 128, "leaky", 1), (5, 13, 13, 32, 96, "mish", 0),
             (7, 8, 4, 256, 256, "relu", 2), (1, 12, 304, 32, 64, "leaky", 1), (2, 38, 38, 256, 512, "leaky", 0), (1, 19, 19, 512, 1024, "leaky", 1)]
    for n, h, wd, cin, cout, act, res_mode in cases:
        x = (rng.standard_normal((n, h, wd, cin)) * rng.choice([0.05, 1.0, 30.0], (1, 1, 1, cin))).astype(F32)
        w = (rng.standard_normal((cout, 9 * cin)) / np.sqrt(9 * cin)).astype(F32)
        bias = rng.standard_normal(cout).astype(F32)
        res = rng.standard_normal((n, h, wd, cout)).astype(F32) if res_mode else None
        want = _conv_ref(x, w, bias, 3, 1, ACT[act], res, res_mode)
        scale = float(np.abs(want).max())
        errs = []
        for on in (0, 1):
            L.check(lib.yds_set_conv_cross8(on))
            got = _run(L, win, x, w, bias, 3, 1, ACT[act], res, res_mode)
            errs.append(float(np.abs(got - want).max()) / scale)
        print(f"n{n} {h}x{wd} {cin}->{cout} {act} res{res_mode}: max err / max|y|  f16x3 {errs[0]:.2e}   cross8 {errs[1]:.2e}", flush=True)
    if os.environ.get("CROSS8_SKIP_ERR"):
        pass
    # timing
    os.environ["YDS_CONV_FORCE"] = str(win)
    if os.environ.get("CROSS8_WIDE64"):                          # the 64-filter tile shapes (ReID 64->64 at 480 crops)
        os.environ["YDS_CONV_FORCE"] = str(names.index("conv3x3_f16x3_win<256,64,%s>" % os.environ["CROSS8_WIDE64"]))
        shapes64 = [(480, 64, 32, 64, 64, 3, 0), (16, 152, 76, 64, 64, 2, 1)]
    shapes = [(16, 76, 76, 128, 256, 1, 1), (16, 38, 38, 256, 512, 1, 1), (16, 19, 19, 512, 1024, 1, 1), (16, 152, 152, 64, 128, 1, 1),
              (16, 76, 76, 128, 128, 2, 0), (16, 38, 38, 256, 256, 2, 0), (16, 19, 19, 512, 512, 2, 0)]
    if os.environ.get("CROSS8_WIDE64"):
        shapes = shapes64
    for n, h, wd, cin, cout, act, res in shapes:
        row = []
        clk = []
        for on in (0, 1, 0, 1):
            L.check(lib.yds_set_conv_cross8(on))
            us, var, ghz, ms = C.c_double(), C.c_int(), C.c_double(), C.c_double()
            lib.yds_conv_clock(C.byref(ghz), C.byref(ms), 1)
            L.check(lib.yds_conv_bench(n, h, wd, cin, cout, 3, 1, act, res, 30, C.byref(us), C.byref(var)))
            lib.yds_conv_clock(C.byref(ghz), C.byref(ms), 1)
            row.append(us.value)
            clk.append(ghz.value)
        fl = 2.0 * n * h * wd * cin * cout * 9
        print(f"b{n} {h}x{wd} {cin}->{cout} act{act} res{res}: f16x3 {row[0]:.1f} / {row[2]:.1f} us ({fl / row[2] / 1e6:.0f} TF)   cross8 {row[1]:.1f} / {row[3]:.1f} us"
              f" ({fl / row[3] / 1e6:.0f} TF)   x{row[2] / row[3]:.2f}   clock {clk[2]:.2f} / {clk[3]:.2f} GHz", flush=True)


if __name__ == "__main__":
    main()
