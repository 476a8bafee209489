"""numpy restatement of the Darknet detector forward pass (TEST ORACLE).

Follows reference yolo3/models/models.py: create_modules :25-102, Mish :16-22,
UpsampleExpand :118-133, YOLOLayer inference branch :185-224,
Darknet.forward :292-313, load_darknet_weights :315-366.  Tensors are NCHW
fp32 like the reference; convolution is im2col + sgemm.
"""

import numpy as np

from .cfg import parse_model_config, parse_model_config_text

F32 = np.float32


def conv2d_nchw(x, w, bias, stride, pad):
    """Plain cross-correlation (torch.nn.Conv2d semantics), fp32."""
    B, C, H, W = x.shape
    O, _, k, _ = w.shape
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    if k == 1 and stride == 1 and pad == 0:
        cols = x.transpose(0, 2, 3, 1).reshape(B * H * W, C)
    else:
        xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), F32)
        xp[:, :, pad:pad + H, pad:pad + W] = x
        win = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(2, 3))
        win = win[:, :, ::stride, ::stride]          # B,C,Ho,Wo,k,k
        cols = np.ascontiguousarray(win.transpose(0, 2, 3, 1, 4, 5)).reshape(B * Ho * Wo, C * k * k)
    out = cols @ np.ascontiguousarray(w.reshape(O, C * k * k).T)
    if bias is not None:
        out += bias[None, :]
    return np.ascontiguousarray(out.reshape(B, Ho, Wo, O).transpose(0, 3, 1, 2))


def batchnorm_eval(x, gamma, beta, mean, var, eps=1e-5):
    """torch eval-mode BatchNorm2d: (x-mean)/sqrt(var+eps)*gamma+beta, fp32."""
    inv = (F32(1.0) / np.sqrt(var + F32(eps))).astype(F32)
    alpha = (gamma * inv).astype(F32)
    shift = (beta - mean * alpha).astype(F32)
    return x * alpha[None, :, None, None] + shift[None, :, None, None]


def leaky(x, slope=0.1):
    return np.where(x > 0, x, x * F32(slope)).astype(F32)


def softplus(x):
    # torch softplus, beta=1, threshold=20
    with np.errstate(over="ignore"):
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(F32)


def mish(x):
    """models.py:20-22  x * tanh(softplus(x))"""
    return (x * np.tanh(softplus(x))).astype(F32)


def sigmoid(x):
    with np.errstate(over="ignore"):
        return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


def maxpool_nchw(x, k, stride, pad, fill=-np.inf):
    B, C, H, W = x.shape
    xp = np.full((B, C, H + 2 * pad, W + 2 * pad), fill, F32)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    win = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(2, 3))[:, :, ::stride, ::stride]
    return np.ascontiguousarray(win.max(axis=(4, 5)))


def yolo_decode(x, anchors, num_classes, img_dim):
    """YOLOLayer.forward inference branch, models.py:185-224 (+ compute_grid_offsets :167-183).

    x: [B, A*(5+C), H, W] -> [B, A*H*W, 5+C]; box index = a*H*W + y*W + x.
    Quirk kept: scale = (img_h/H, img_w/W) is applied as (x*s_h, y*s_w, w*s_h, h*s_w).
    """
    B, _, H, W = x.shape
    A = len(anchors)
    p = x.reshape(B, A, num_classes + 5, H, W).transpose(0, 1, 3, 4, 2)
    xy = sigmoid(p[..., 0:2])
    wh = p[..., 2:4]
    conf_cls = sigmoid(p[..., 4:])
    scale = np.array([[img_dim[0] / H, img_dim[1] / W]], dtype=F32)       # (1,2)
    gy, gx = np.meshgrid(np.arange(H, dtype=F32), np.arange(W, dtype=F32), indexing="ij")
    grid = np.stack((gx.reshape(-1), gy.reshape(-1)), 1).reshape(1, 1, H, W, 2)
    scaled_anchors = (np.array(anchors, dtype=F32) / scale).astype(F32)    # (A,2)
    anchor = scaled_anchors.reshape(1, A, 1, 1, 2)
    boxes = np.concatenate([xy + grid, np.exp(wh) * anchor], axis=-1).astype(F32)
    out = np.concatenate(
        (boxes.reshape(B, -1, 4) * np.tile(scale, (1, 2)),
         conf_cls[..., 0].reshape(B, -1, 1),
         conf_cls[..., 1:].reshape(B, -1, num_classes)), -1)
    return out.astype(F32)


class DarknetOracle:
    """Graph interpreter equivalent to reference Darknet (models.py:277-313)."""

    def __init__(self, cfg, img_size=416, is_text=False):
        defs = parse_model_config_text(cfg) if is_text else parse_model_config(cfg)
        self.hyperparams = defs.pop(0)
        self.module_defs = defs
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.params = [None] * len(defs)
        self.out_filters = []
        prev = int(self.hyperparams["channels"])
        filters_hist = [prev]
        for i, d in enumerate(defs):
            t = d["type"]
            filters = filters_hist[-1]
            if t == "convolutional":
                bn = int(d["batch_normalize"])
                filters = int(d["filters"])
                k = int(d["size"])
                cin = filters_hist[-1]
                self.params[i] = dict(bn=bn, k=k, stride=int(d["stride"]), pad=(k - 1) // 2,
                                      cin=cin, cout=filters, act=d["activation"])
            elif t == "route":
                layers = [int(v) for v in d["layers"].split(",")]
                filters = sum(filters_hist[1:][l] for l in layers)
                if "groups" in d:
                    filters //= int(d["groups"])
            elif t == "shortcut":
                filters = filters_hist[1:][int(d["from"])]
            elif t == "yolo":
                idx = [int(v) for v in d["mask"].split(",")]
                a = [int(v) for v in d["anchors"].split(",")]
                a = [(a[j], a[j + 1]) for j in range(0, len(a), 2)]
                self.params[i] = dict(anchors=[a[j] for j in idx], classes=int(d["classes"]))
            filters_hist.append(filters)
        self.out_filters = filters_hist[1:]

    # ---- weights -----------------------------------------------------
    def n_weight_floats(self):
        n = 0
        for p, d in zip(self.params, self.module_defs):
            if d["type"] == "convolutional":
                n += (4 if p["bn"] else 1) * p["cout"] + p["cout"] * p["cin"] * p["k"] * p["k"]
        return n

    def load_weights_array(self, weights, cutoff=None):
        """weights: fp32 stream after the 5xint32 header (models.py:323-366)."""
        ptr = 0
        for i, (p, d) in enumerate(zip(self.params, self.module_defs)):
            if i == cutoff:
                break
            if d["type"] != "convolutional":
                continue
            co = p["cout"]
            if d["batch_normalize"]:          # truthiness, like the reference (:336)
                p["beta"] = weights[ptr:ptr + co].copy(); ptr += co
                p["gamma"] = weights[ptr:ptr + co].copy(); ptr += co
                p["mean"] = weights[ptr:ptr + co].copy(); ptr += co
                p["var"] = weights[ptr:ptr + co].copy(); ptr += co
            else:
                p["bias"] = weights[ptr:ptr + co].copy(); ptr += co
            nw = co * p["cin"] * p["k"] * p["k"]
            p["w"] = weights[ptr:ptr + nw].reshape(co, p["cin"], p["k"], p["k"]).copy(); ptr += nw
        return ptr

    def load_darknet_weights(self, path):
        with open(path, "rb") as f:
            self.header_info = np.fromfile(f, dtype=np.int32, count=5)
            weights = np.fromfile(f, dtype=np.float32)
        cutoff = 75 if "darknet53.conv.74" in path else None
        return self.load_weights_array(weights, cutoff)

    # ---- forward -----------------------------------------------------
    def forward(self, x, keep_layers=False, inject=None):
        """x: [B,3,H,W] fp32 -> [B,N,5+C].  ``inject`` (optional) is a callable
        (layer_index, head_tensor NCHW) -> head_tensor applied to the raw head
        input of every yolo layer (bench logit injection)."""
        x = np.ascontiguousarray(x, dtype=F32)
        img_dim = (x.shape[2], x.shape[3])
        outs, yolo_out = [], []
        for i, (d, p) in enumerate(zip(self.module_defs, self.params)):
            t = d["type"]
            if t == "convolutional":
                x = conv2d_nchw(x, p["w"], None if p["bn"] else p["bias"], p["stride"], p["pad"])
                if p["bn"]:
                    x = batchnorm_eval(x, p["gamma"], p["beta"], p["mean"], p["var"])
                if p["act"] == "leaky":
                    x = leaky(x)
                elif p["act"] == "mish":
                    x = mish(x)
            elif t == "maxpool":
                k, s = int(d["size"]), int(d["stride"])
                if k == 2 and s == 1:
                    # models.py:61-63 ZeroPad2d((0,1,0,1)) then MaxPool2d(2,1,pad 0)
                    B, C, H, W = x.shape
                    xz = np.zeros((B, C, H + 1, W + 1), F32)
                    xz[:, :, :H, :W] = x
                    x = maxpool_nchw(xz, 2, 1, 0)
                else:
                    x = maxpool_nchw(x, k, s, (k - 1) // 2)
            elif t == "upsample":
                s = int(d["stride"])
                x = np.ascontiguousarray(x.repeat(s, axis=2).repeat(s, axis=3))
            elif t == "route":
                x = np.concatenate([outs[int(l)] for l in d["layers"].split(",")], 1)
                if "groups" in d:
                    g, gid = int(d["groups"]), int(d["group_id"])
                    c = x.shape[1] // g
                    x = np.ascontiguousarray(x[:, gid * c:(gid + 1) * c])
            elif t == "shortcut":
                x = outs[-1] + outs[int(d["from"])]
            elif t == "yolo":
                if inject is not None:
                    x = inject(i, x)
                x = yolo_decode(x, p["anchors"], p["classes"], img_dim)
                yolo_out.append(x)
            outs.append(x)
        self.layer_outputs = outs if keep_layers else None
        return np.concatenate(yolo_out, 1)

    __call__ = forward


def yolo_heads(oracle, H, W):
    """[(h, w, [(aw, ah) ...])] per yolo layer in network order (what the product's Darknet.yolo_heads() reports)."""
    sizes, heads = [], []
    h, w = H, W
    for d, p in zip(oracle.module_defs, oracle.params):
        t = d["type"]
        if t == "convolutional":
            h = (h + 2 * p["pad"] - p["k"]) // p["stride"] + 1
            w = (w + 2 * p["pad"] - p["k"]) // p["stride"] + 1
        elif t == "maxpool":
            k, s = int(d["size"]), int(d["stride"])
            if not (k == 2 and s == 1):
                pad = (k - 1) // 2
                h, w = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        elif t == "upsample":
            h, w = h * int(d["stride"]), w * int(d["stride"])
        elif t == "route":
            h, w = sizes[int(d["layers"].split(",")[0])]
        elif t == "yolo":
            heads.append((h, w, list(p["anchors"])))
        sizes.append((h, w))
    return heads


def conv_flops(oracle, H, W):
    """2*MAC over conv layers for an HxW input (SURVEY 8d algorithmic FLOPs)."""
    shapes = []
    h, w = H, W
    total = 0
    for d, p in zip(oracle.module_defs, oracle.params):
        t = d["type"]
        if t == "convolutional":
            ho = (h + 2 * p["pad"] - p["k"]) // p["stride"] + 1
            wo = (w + 2 * p["pad"] - p["k"]) // p["stride"] + 1
            total += 2 * ho * wo * p["cout"] * p["cin"] * p["k"] * p["k"]
            h, w = ho, wo
        elif t == "maxpool":
            k, s = int(d["size"]), int(d["stride"])
            if not (k == 2 and s == 1):
                pad = (k - 1) // 2
                h = (h + 2 * pad - k) // s + 1
                w = (w + 2 * pad - k) // s + 1
        elif t == "upsample":
            h, w = h * int(d["stride"]), w * int(d["stride"])
        elif t == "route":
            l = [int(v) for v in d["layers"].split(",")][0]
            h, w = shapes[l] if l >= 0 else shapes[len(shapes) + l]
        elif t == "shortcut":
            pass
        shapes.append((h, w))
    return total
