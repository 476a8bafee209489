#!/usr/bin/env python3
"""Per-shape timing of the implicit-GEMM conv kernel (tuning aid; run on the GPU box).

    python tools/conv_bench.py [--batch 8] [--net yolov3|yolov4|reid]
Prints TFLOP/s per unique layer shape and the FLOP-weighted total."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import _lib, cfgs, synth  # noqa: E402


def detector_shapes(name, size):
    text = cfgs.cfg_text(name, size, size)
    blocks = synth._parse_cfg(text)[1:]
    shapes, hw, filt = {}, [], [3]
    h = w = size
    for i, b in enumerate(blocks):
        t = b["type"]
        c = filt[-1]
        if t == "convolutional":
            k, s = int(b["size"]), int(b["stride"])
            cin, c = filt[-1], int(b["filters"])
            act = {"leaky": 1, "mish": 2}.get(b["activation"], 0)
            res = 1 if i + 1 < len(blocks) and blocks[i + 1]["type"] == "shortcut" else 0
            key = (h, w, max(cin, 4), c, k, s, act, res)
            shapes[key] = shapes.get(key, 0) + 1
            h, w = (h + 2 * ((k - 1) // 2) - k) // s + 1, (w + 2 * ((k - 1) // 2) - k) // s + 1
        elif t == "maxpool" and not (int(b["size"]) == 2 and int(b["stride"]) == 1):
            k, s = int(b["size"]), int(b["stride"])
            h, w = (h + 2 * ((k - 1) // 2) - k) // s + 1, (w + 2 * ((k - 1) // 2) - k) // s + 1
        elif t == "upsample":
            h, w = h * int(b["stride"]), w * int(b["stride"])
        elif t == "route":
            ls = [int(v) for v in b["layers"].split(",")]
            c = sum(filt[1:][l] for l in ls)
            if "groups" in b:
                c //= int(b["groups"])
            l0 = ls[0] if ls[0] >= 0 else i + ls[0]
            h, w = hw[l0]
        elif t == "shortcut":
            c = filt[1:][int(b["from"])]
        hw.append((h, w))
        filt.append(c)
    return shapes


def reid_shapes():
    sh = {(128, 64, 4, 64, 3, 1, 3, 0): 1}
    h, w = 64, 32
    for cin, cout, down in ((64, 64, False), (64, 128, True), (128, 256, True), (256, 512, True)):
        if down:
            sh[(h, w, cin, cout, 3, 2, 3, 0)] = 1
            sh[(h, w, cin, cout, 1, 2, 0, 0)] = 1
            h, w = h // 2, w // 2
            sh[(h, w, cout, cout, 3, 1, 3, 0)] = 3
        else:
            sh[(h, w, cin, cout, 3, 1, 3, 0)] = 4
    return sh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--net", default="yolov3")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None, help="h,w,cin,cout,k,s,act,res : run just this shape")
    args = ap.parse_args()
    _lib.init(0)
    lib = _lib.load()
    shapes = reid_shapes() if args.net == "reid" else detector_shapes(args.net, 608)
    if args.only:
        shapes = {tuple(int(v) for v in args.only.split(",")): 1}
    tot_f = tot_t = 0.0
    print(f"{'h':>4} {'w':>4} {'cin':>5} {'cout':>5} k s act res  cnt   {'us':>9} {'TF/s':>7}  variant")
    for (h, w, cin, cout, k, s, act, res), cnt in shapes.items():
        us, var = C.c_double(), C.c_int()
        _lib.check(lib.yds_conv_bench(args.batch, h, w, cin, cout, k, s, act, res, args.iters, C.byref(us), C.byref(var)))
        pad = (k - 1) // 2
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        fl = 2.0 * args.batch * ho * wo * cout * k * k * cin
        tot_f += fl * cnt
        tot_t += us.value * cnt
        print(f"{h:4d} {w:4d} {cin:5d} {cout:5d} {k} {s} {act:3d} {res:3d} {cnt:4d} {us.value:10.1f} {fl / us.value / 1e6:7.1f}  "
              f"{lib.yds_conv_variant_name(var.value).decode()}")
    print(f"total {tot_t / 1e3:.2f} ms per batch of {args.batch}  ->  {tot_f / tot_t / 1e6:.1f} TFLOP/s  "
          f"({tot_t / args.batch:.0f} us/image)")


if __name__ == "__main__":
    main()
