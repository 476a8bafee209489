"""Numerics study (CPU, no GPU): what the detector's tensors lose if the two cross terms of the f16x3 product
(hi x lo, lo x hi) are computed in fp8 e4m3 on the 32x32x64 f8f6f4 MFMA instead of fp16 (tools/probes/fp8_cross_probe.hip
measures x1.67 on the matrix pipe under the power limit for that change).  Emulation in float64 of
    exact        : float64 product of the fp32 operands
    f16x3        : hi*hi + hi*lo + lo*hi with fp16 hi / lo                                   (what the kernels do today)
    f16 + fp8x   : hi*hi + q8(hi)*q8(lo) + q8(lo)*q8(hi), q8 = e4m3 with a power-of-two scale per 32 channels (MX block scale)
    f16 only     : hi*hi                                                                       (half mode)
through the whole yolov3 / yolov4 graph with the seeded synthetic weights at a reduced resolution; the error of every
scheme is measured on the raw head tensors against the exact pass, relative to the tensor's largest magnitude.

usage: python tools/fp8_cross_numerics.py [yolov3|yolov4] [size]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import darknet as od                      # noqa: E402   (a tool, not the product path)
from yolo_deepsort_amd import cfgs, synth             # noqa: E402


def split16(v):
    hi = v.astype(np.float16).astype(np.float64)
    lo = (v - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def q8(v, axis_blocks):
    """e4m3 (3 mantissa bits, normal exponents -6..8, subnormal step 2^-9, max 448) with one power-of-two scale per
    block of 32 along the last axis (v: [..., K], K a multiple of 32 or smaller than 32)."""
    shp = v.shape
    K = shp[-1]
    blk = 32 if K % 32 == 0 else K
    b = v.reshape(shp[:-1] + (K // blk, blk))
    amax = np.abs(b).max(-1, keepdims=True)
    with np.errstate(divide="ignore"):
        e = np.where(amax > 0, np.floor(np.log2(np.where(amax > 0, amax, 1.0))), 0.0)
    scale = 2.0 ** (e - 7)                                 # block maximum lands in [128, 256)
    t = b / scale
    a = np.abs(t)
    with np.errstate(divide="ignore"):
        ex = np.where(a > 0, np.floor(np.log2(np.where(a > 0, a, 1.0))), -6.0)
    ex = np.clip(ex, -6, 8)
    step = 2.0 ** (ex - 3)
    r = np.minimum(np.round(a / step) * step, 448.0)
    return (np.sign(t) * r * scale).reshape(shp)


def cols_of(x, k, stride, pad):
    B, C, H, W = x.shape
    if k == 1:
        return x.transpose(0, 2, 3, 1).reshape(B * H * W, 1, C), H, W
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), x.dtype)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    win = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(2, 3))[:, :, ::stride, ::stride]
    Ho, Wo = win.shape[2], win.shape[3]
    return np.ascontiguousarray(win.transpose(0, 2, 3, 4, 5, 1)).reshape(B * Ho * Wo, k * k, C), Ho, Wo      # [M, taps, C]


def conv_scheme(x, w, stride, pad, scheme):
    """x [B,C,H,W] float64 (fp32-representable), w [O,C,k,k] -> [B,O,Ho,Wo] float64."""
    O, C, k, _ = w.shape
    a, Ho, Wo = cols_of(x, k, stride, pad)                                 # [M, taps, C]
    wt = np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(O, k * k, C)   # [O, taps, C]
    M = a.shape[0]
    mm = lambda p, q: p.reshape(M, -1) @ q.reshape(O, -1).T
    if scheme == "exact":
        y = mm(a, wt)
    else:
        ah, al = split16(a)
        wh, wl = split16(wt)
        y = mm(ah, wh)
        if scheme == "f16x3":
            y = y + mm(ah, wl) + mm(al, wh)
        elif scheme == "fp8x":
            y = y + mm(q8(ah, 0), q8(wl, 0)) + mm(q8(al, 0), q8(wh, 0))
        elif scheme != "f16":
            raise ValueError(scheme)
    B = x.shape[0]
    return np.ascontiguousarray(y.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2))


def forward(net, x, scheme):
    """oracle.darknet.DarknetOracle.forward with the convolution swapped; returns the raw head tensors."""
    f32 = lambda v: v.astype(np.float32).astype(np.float64)
    outs, heads = [], []
    x = f32(x)
    for d, p in zip(net.module_defs, net.params):
        t = d["type"]
        if t == "convolutional":
            y = conv_scheme(x, p["w"].astype(np.float64), p["stride"], p["pad"], scheme)
            if p["bn"]:
                inv = 1.0 / np.sqrt(p["var"].astype(np.float64) + 1e-5)
                al = p["gamma"] * inv
                y = y * al[None, :, None, None] + (p["beta"] - p["mean"] * al)[None, :, None, None]
            else:
                y = y + p["bias"].astype(np.float64)[None, :, None, None]
            if p["act"] == "leaky":
                y = np.where(y > 0, y, 0.1 * y)
            elif p["act"] == "mish":
                y = y * np.tanh(np.where(y > 20, y, np.log1p(np.exp(np.minimum(y, 20)))))
            x = f32(y)
        elif t == "maxpool":
            k, s = int(d["size"]), int(d["stride"])
            x = od.maxpool_nchw(x.astype(np.float32), k, s, (k - 1) // 2).astype(np.float64)
        elif t == "upsample":
            s = int(d["stride"])
            x = x.repeat(s, axis=2).repeat(s, axis=3)
        elif t == "route":
            x = np.concatenate([outs[int(l)] for l in d["layers"].split(",")], 1)
        elif t == "shortcut":
            x = f32(outs[-1] + outs[int(d["from"])])
        elif t == "yolo":
            heads.append(x)
        outs.append(x)
    return heads, outs


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolov3"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    text = cfgs.cfg_text(name, size, size)
    net = od.DarknetOracle(text, img_size=size, is_text=True)
    blob = synth.darknet_weights_blob(text, seed=0)
    net.load_weights_array(np.frombuffer(blob, dtype=np.float32, offset=20))
    rng = np.random.default_rng(3)
    x = rng.random((1, 3, size, size), dtype=np.float32)
    ref_heads, ref_outs = forward(net, x, "exact")
    print(f"{name} {size}x{size}, {sum(1 for d in net.module_defs if d['type'] == 'convolutional')} convolutions; error of the raw head tensors against the float64-product pass")
    for scheme in ("f16x3", "fp8x", "f16"):
        heads, outs = forward(net, x, scheme)
        line = []
        for h, r in zip(heads, ref_heads):
            line.append(f"max|d| {np.abs(h - r).max():.3e} (rel to max {np.abs(h - r).max() / np.abs(r).max():.3e}, |head|max {np.abs(r).max():.2f})")
        worst = max(np.abs(o - r).max() / max(np.abs(r).max(), 1e-30) for o, r in zip(outs, ref_outs))
        print(f"  {scheme:6s} " + " | ".join(line) + f" | worst layer rel {worst:.3e}")


if __name__ == "__main__":
    main()
