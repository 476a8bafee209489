"""Multi-threaded form of the oracle's detector / ReID forward (TEST ORACLE, CPU baseline).

Same layer semantics as oracle/darknet.py and oracle/reid.py (which restate reference yolo3/models/models.py:25-102,
185-224,292-313 and deep_sort/deep/model.py:5-95), on NHWC tensors: patch gather, BatchNorm/activation/residual
epilogue, pooling and up-sampling in C + OpenMP (oracle/csrc/fastconv.c), the GEMM through numpy's BLAS.  It exists so
that bench.py's `cpu_baseline` leg uses every host core; tests/test_oracle_fast.py holds it to the plain numpy oracle.
"""

import ctypes
import os
import subprocess

import numpy as np

from .darknet import yolo_decode

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "fastconv.c")
_SO = os.path.join(_HERE, "_build", "libfastconv.so")
_lib = None
_ACT = {"linear": 0, "leaky": 1, "mish": 2, "relu": 3}


def build(force=False):
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-std=c11", _SRC, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def threads():
    """Host threads the baseline uses: OpenMP's team for the C parts (BLAS uses its own pool, capped by its build)."""
    try:
        omp = ctypes.CDLL("libgomp.so.1").omp_get_max_threads()
    except OSError:
        omp = os.cpu_count()
    return int(omp)


def conv_nhwc(x, wp, k, stride, pad, scale, shift, act, res=None, res_mode=0):
    """x [B,H,W,C] fp32, wp [k*k*C, O] -> act(conv * scale + shift (+res)) [B,Ho,Wo,O]"""
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if k == 1 and stride == 1:
        cols = x.reshape(B * H * W, C)
    else:
        cols = np.empty((B * Ho * Wo, k * k * C), F32)
        lib().im2col_nhwc(_p(x), B, H, W, C, k, stride, pad, _p(cols))
    y = cols @ wp
    lib().scale_shift_act(_p(y), ctypes.c_int64(y.shape[0]), y.shape[1], _p(scale), _p(shift), act,
                          _p(res) if res is not None else None, res_mode)
    return y.reshape(B, Ho, Wo, -1)


def _fold(p):
    """(wp, scale, shift) of one oracle conv block: BatchNorm2d(eps 1e-5) as scale/shift, or the plain bias."""
    w = p["w"]
    O, C, k, _ = w.shape
    wp = np.ascontiguousarray(w.transpose(2, 3, 1, 0).reshape(k * k * C, O))
    if p["bn"]:
        inv = (F32(1.0) / np.sqrt(p["var"] + F32(1e-5))).astype(F32)
        scale = (p["gamma"] * inv).astype(F32)
        shift = (p["beta"] - p["mean"] * scale).astype(F32)
    else:
        scale, shift = np.ones(O, F32), p["bias"].astype(F32)
    return wp, np.ascontiguousarray(scale), np.ascontiguousarray(shift)


class DarknetFast:
    """Forward of a loaded oracle.darknet.DarknetOracle on the multi-threaded kernels."""

    def __init__(self, oracle):
        self.o = oracle
        self.folded = [(_fold(p) if d["type"] == "convolutional" else None) for d, p in zip(oracle.module_defs, oracle.params)]

    def forward(self, x, inject=None):
        x = np.ascontiguousarray(x, dtype=F32)
        B, C, H, W = x.shape
        img_dim = (H, W)
        cur = np.empty((B, H, W, C), F32)
        lib().nchw_to_nhwc(_p(x), B, C, H, W, _p(cur))
        outs, yolo_out = [], []
        defs = self.o.module_defs
        for i, (d, p) in enumerate(zip(defs, self.o.params)):
            t = d["type"]
            if t == "convolutional":
                wp, scale, shift = self.folded[i]
                cur = conv_nhwc(cur, wp, p["k"], p["stride"], p["pad"], scale, shift, _ACT[p["act"]])
            elif t == "maxpool":
                k, s = int(d["size"]), int(d["stride"])
                zero_br = 1 if (k == 2 and s == 1) else 0
                pad = 0 if zero_br else (k - 1) // 2
                b, h, w, c = cur.shape
                ho = (h + zero_br + 2 * pad - k) // s + 1
                wo = (w + zero_br + 2 * pad - k) // s + 1
                y = np.empty((b, ho, wo, c), F32)
                lib().maxpool_nhwc(_p(cur), b, h, w, c, k, s, pad, zero_br, _p(y))
                cur = y
            elif t == "upsample":
                s = int(d["stride"])
                b, h, w, c = cur.shape
                y = np.empty((b, h * s, w * s, c), F32)
                lib().upsample_nhwc(_p(cur), b, h, w, c, s, _p(y))
                cur = y
            elif t == "route":
                srcs = [outs[int(v)] for v in d["layers"].split(",")]
                cur = srcs[0] if len(srcs) == 1 else np.concatenate(srcs, -1)
                if "groups" in d:
                    g, gi = int(d["groups"]), int(d["group_id"])
                    c = cur.shape[-1] // g
                    cur = np.ascontiguousarray(cur[..., gi * c:(gi + 1) * c])
            elif t == "shortcut":
                y = outs[-1].copy()
                a = outs[int(d["from"])]
                lib().add_inplace(_p(y), _p(np.ascontiguousarray(a)), ctypes.c_int64(y.size))
                cur = y
            elif t == "yolo":
                b, h, w, c = cur.shape
                head = np.empty((b, c, h, w), F32)
                lib().nhwc_to_nchw(_p(np.ascontiguousarray(cur)), b, h, w, c, _p(head))
                if inject is not None:
                    head = inject(i, head)
                cur = yolo_decode(head, p["anchors"], p["classes"], img_dim)
                yolo_out.append(cur)
            outs.append(cur)
        return np.concatenate(yolo_out, 1)

    __call__ = forward

    # what oracle.pipeline.run_stream reads off a net
    @property
    def img_size(self):
        return self.o.img_size

    @property
    def module_defs(self):
        return self.o.module_defs

    @property
    def params(self):
        return self.o.params


class ReidFast:
    """oracle.reid.reid_forward on the multi-threaded kernels (model.py:48-95)."""

    def __init__(self, sd):
        self.f = {}

        def fold(conv, bn, bias=None):
            p = dict(w=np.asarray(sd[conv + ".weight"], F32), bn=1, gamma=np.asarray(sd[bn + ".weight"], F32), beta=np.asarray(sd[bn + ".bias"], F32),
                     mean=np.asarray(sd[bn + ".running_mean"], F32), var=np.asarray(sd[bn + ".running_var"], F32))
            wp, scale, shift = _fold(p)
            if bias is not None:                                   # conv bias in front of the BN (the stem)
                shift = (shift + np.asarray(sd[bias], F32) * scale).astype(F32)
            return wp, scale, np.ascontiguousarray(shift)
        self.f["stem"] = fold("conv.0", "conv.1", "conv.0.bias")
        from .reid import STAGES
        self.stages = STAGES
        for name, cin, cout, down in STAGES:
            for b in range(2):
                pfx = f"{name}.{b}"
                self.f[pfx + ".1"] = fold(pfx + ".conv1", pfx + ".bn1")
                self.f[pfx + ".2"] = fold(pfx + ".conv2", pfx + ".bn2")
                if b == 0 and down:
                    self.f[pfx + ".d"] = fold(pfx + ".downsample.0", pfx + ".downsample.1")

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=F32)
        D, C, H, W = x.shape
        cur = np.empty((D, H, W, C), F32)
        lib().nchw_to_nhwc(_p(x), D, C, H, W, _p(cur))
        cur = conv_nhwc(cur, *self.f["stem"][:1], 3, 1, 1, *self.f["stem"][1:], 3)
        d, h, w, c = cur.shape
        y = np.empty((d, (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, c), F32)
        lib().maxpool_nhwc(_p(cur), d, h, w, c, 3, 2, 1, 0, _p(y))
        cur = y
        for name, cin, cout, down in self.stages:
            for b in range(2):
                pfx = f"{name}.{b}"
                ds = b == 0 and down
                wp, sc, sh = self.f[pfx + ".1"]
                y1 = conv_nhwc(cur, wp, 3, 2 if ds else 1, 1, sc, sh, 3)
                res = cur
                if ds:
                    wp, sc, sh = self.f[pfx + ".d"]
                    res = conv_nhwc(cur, wp, 1, 2, 0, sc, sh, 0)
                wp, sc, sh = self.f[pfx + ".2"]
                cur = conv_nhwc(y1, wp, 3, 1, 1, sc, sh, 3, res=np.ascontiguousarray(res).reshape(-1, res.shape[-1]), res_mode=2)
        f = cur.mean(axis=(1, 2), dtype=F32)
        nrm = np.sqrt((f * f).sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
        return (f / nrm).astype(F32)
