"""Seeded synthetic inputs for tests and bench.

No weights, videos or datasets ship with the reference and there is no network,
so everything the hot path consumes is synthesised here in the reference's own
file formats: Darknet ``.weights`` blobs (layout of reference
``yolo3/models/models.py:315-366``), a ReID ``net_dict`` state dict (keys of
``deep_sort/deep/model.py:48-95``), 1080p RGB frames with textured "persons",
and per-frame detection boxes / appearance features for tracker-only runs.
All generators use ``numpy.random.RandomState`` (frozen stream) so fixtures made
in one container reproduce bit-for-bit in another.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# cfg walking (only what weight synthesis needs)
# --------------------------------------------------------------------------
def _parse_cfg(text):
    blocks = []
    for line in text.split("\n"):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("["):
            blocks.append({"type": line[1:-1].strip()})
        else:
            k, v = line.split("=")
            blocks[-1][k.strip()] = v.strip()
    return blocks


def conv_shapes(cfg_text):
    """[(layer_index, cin, cout, k, bn, is_head, feeds_shortcut)] for every convolutional block."""
    blocks = _parse_cfg(cfg_text)
    net, blocks = blocks[0], blocks[1:]
    filt = [int(net.get("channels", 3))]
    out = []
    for i, b in enumerate(blocks):
        t = b["type"]
        f = filt[-1]
        if t == "convolutional":
            f = int(b["filters"])
            nxt = blocks[i + 1]["type"] if i + 1 < len(blocks) else ""
            out.append((i, filt[-1], f, int(b["size"]), int(b.get("batch_normalize", 0)),
                        nxt == "yolo", nxt == "shortcut"))
        elif t == "route":
            ls = [int(v) for v in b["layers"].split(",")]
            f = sum(filt[1:][l] for l in ls)
            if "groups" in b:
                f //= int(b["groups"])
        elif t == "shortcut":
            f = filt[1:][int(b["from"])]
        filt.append(f)
    return out


def _wide_bn(rng, c):
    """BatchNorm statistics over the range a trained net shows (VERDICT r5 'next' #2): running_var log-uniform over 1e-3 .. 1e2,
    gamma ~ U(0, 2) with one channel in 16 EXACTLY zero, |beta| <= 3, |running_mean| <= 3 standard deviations.  Returns
    (beta, gamma, mean, var, row): row[c] = sqrt(var[c]) * m[c] is the standard deviation the conv row of channel c gives its
    output on an input of unit second moment, so that var[c] IS (up to the mismatch m[c], log-uniform over two decades) the variance
    of what the layer normalises - as in a trained net, where running_var tracks the real pre-BN variance; unit-gain rows under
    var = 1e-3 would gain 30x per layer instead, which no trained net does and fp32 itself does not survive for a dozen layers.
    m is normalised per layer to mean(gamma^2 m^2) = 2: with the activation halving the power, what the input contributes passes at
    unit gain (He's criterion) while beta and mean add a constant ~3.5 per layer - the second moment grows linearly with depth
    (rms ~7 after 13 layers), channel magnitudes inside a layer span 1e-3 .. 1e2."""
    var = (10.0 ** rng.uniform(-3.0, 2.0, c)).astype(F32)
    gamma = rng.uniform(0.0, 2.0, c).astype(F32)
    gamma[rng.permutation(c)[:max(1, c // 16)]] = 0.0
    beta = rng.uniform(-3.0, 3.0, c).astype(F32)
    mean = (rng.uniform(-3.0, 3.0, c) * np.sqrt(var)).astype(F32)
    m = 10.0 ** rng.uniform(-1.5, 0.5, c)
    m *= np.sqrt(2.0 / np.mean(gamma.astype(np.float64) ** 2 * m ** 2))
    return beta, gamma, mean, var, np.sqrt(var.astype(np.float64)) * m


def darknet_weights_blob(cfg_text, seed=0, obj_bias=-4.0, num_classes=80, profile="unit"):
    """Synthetic ``.weights`` file bytes: 5xint32 header then, per conv block,
    [beta, gamma, mean, var] (BN) or [bias], then W[cout,cin,k,k] (fp32).

    profile "unit" (bench.py, most fixtures): conv W ~ N(0, 2/(k*k*cin)) (N(0, 1/cin) for the linear heads); gamma ~
    U(.8,1.2); beta, mean ~ N(0,.1); var ~ U(.8,1.25), so each conv block has
    unit gain on average; the conv feeding a shortcut gets gamma/4 so 23
    residual units do not blow activations up (a trained net's BN statistics do
    the same job).  Head (pre-yolo) convs get objectness bias ``obj_bias`` so
    candidates are sparse.
    profile "wide" (tests/golden/wide_range.npz): BatchNorm statistics over decades, see _wide_bn; the first layer's rows
    assume the image's second moment 1/3, the others 1 (see _wide_bn)."""
    rng = np.random.RandomState(seed)
    parts = [np.array([0, 2, 0, 0, 0], dtype=np.int32).tobytes()]
    if profile == "wide":
        for n, (_, cin, cout, k, bn, is_head, pre_short) in enumerate(conv_shapes(cfg_text)):
            s_in = 1.0 / 3.0 if n == 0 else (1.0 + 3.5 * n if is_head else 1.0)       # heads: about unit-rms logits
            if bn:
                beta, gamma, mean, var, row = _wide_bn(rng, cout)
                parts += [beta.tobytes(), gamma.tobytes(), mean.tobytes(), var.tobytes()]
            else:
                bias = (rng.randn(cout) * 0.1).astype(F32)
                if is_head:
                    bias[4::num_classes + 5] = obj_bias
                parts.append(bias.tobytes())
                row = np.ones(cout)
            w = rng.randn(cout, cin * k * k) * (row / np.sqrt(k * k * cin * s_in))[:, None]
            parts.append(w.astype(F32).tobytes())
        return b"".join(parts)
    if profile != "unit":
        raise ValueError(profile)
    for _, cin, cout, k, bn, is_head, pre_short in conv_shapes(cfg_text):
        if bn:
            beta = (rng.randn(cout) * 0.1).astype(F32)
            gamma = rng.uniform(0.8, 1.2, cout).astype(F32)
            if pre_short:
                gamma *= F32(0.25)
            mean = (rng.randn(cout) * 0.1).astype(F32)
            var = rng.uniform(0.8, 1.25, cout).astype(F32)
            parts += [beta.tobytes(), gamma.tobytes(), mean.tobytes(), var.tobytes()]
        else:
            bias = (rng.randn(cout) * 0.1).astype(F32)
            if is_head:
                bias[4::num_classes + 5] = obj_bias
            parts.append(bias.tobytes())
        gain = 1.0 if is_head else 2.0
        w = (rng.randn(cout * cin * k * k) * np.sqrt(gain / (k * k * cin))).astype(F32)
        parts.append(w.tobytes())
    return b"".join(parts)


REID_STAGES = (("layer1", 64, 64, False), ("layer2", 64, 128, True),
               ("layer3", 128, 256, True), ("layer4", 256, 512, True))


def reid_state_dict(seed=0, profile="unit"):
    """Seeded synthetic ReID weights with the ckpt.t7 ``net_dict`` key set (130
    tensors incl. the unused classifier, reference deep_sort/deep/model.py:48-80).
    profile "wide": every BatchNorm with _wide_bn's statistics (a conv's rows are drawn when its BatchNorm is)."""
    rng = np.random.RandomState(seed)
    sd = {}
    wide = profile == "wide"
    if profile not in ("unit", "wide"):
        raise ValueError(profile)
    pending = {}

    def conv(name, co, ci, k, bias=False):
        if wide:
            pending["conv"] = (name, co, ci, k)                      # drawn by the bn() that follows (every conv here has one)
        else:
            sd[name + ".weight"] = (rng.randn(co, ci, k, k) * np.sqrt(2.0 / (ci * k * k))).astype(F32)
        if bias:
            sd[name + ".bias"] = (rng.randn(co) * 0.1).astype(F32)

    def bn(name, c, damp=1.0):
        if wide:
            beta, gamma, mean, var, row = _wide_bn(rng, c)
            if "conv" in pending:
                cname, co, ci, k = pending.pop("conv")
                sd[cname + ".weight"] = (rng.randn(co, ci, k, k) * (row / np.sqrt(ci * k * k))[:, None, None, None]).astype(F32)
            sd[name + ".weight"], sd[name + ".bias"], sd[name + ".running_mean"], sd[name + ".running_var"] = gamma, beta, mean, var
            sd[name + ".num_batches_tracked"] = np.array(0, dtype=np.int64)
            return
        sd[name + ".weight"] = (rng.uniform(0.8, 1.2, c) * damp).astype(F32)
        sd[name + ".bias"] = (rng.randn(c) * 0.1).astype(F32)
        sd[name + ".running_mean"] = (rng.randn(c) * 0.1).astype(F32)
        sd[name + ".running_var"] = rng.uniform(0.8, 1.25, c).astype(F32)
        sd[name + ".num_batches_tracked"] = np.array(0, dtype=np.int64)

    conv("conv.0", 64, 3, 3, bias=True)
    bn("conv.1", 64)
    for name, cin, cout, down in REID_STAGES:
        for b in range(2):
            ci = cin if b == 0 else cout
            conv(f"{name}.{b}.conv1", cout, ci, 3)
            bn(f"{name}.{b}.bn1", cout)
            conv(f"{name}.{b}.conv2", cout, cout, 3)
            bn(f"{name}.{b}.bn2", cout, 0.5)
            if b == 0 and down:
                conv(f"{name}.{b}.downsample.0", cout, ci, 1)
                bn(f"{name}.{b}.downsample.1", cout)
    sd["classifier.0.weight"] = (rng.randn(256, 512) * 0.05).astype(F32)
    sd["classifier.0.bias"] = np.zeros(256, F32)
    bn("classifier.1", 256)
    sd["classifier.4.weight"] = (rng.randn(751, 256) * 0.05).astype(F32)
    sd["classifier.4.bias"] = np.zeros(751, F32)
    return sd


# --------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------
class PersonScene:
    """Scripted pedestrians on a 1080p canvas.

    ``n_persons`` walkers with box w~U(40,80), h~U(100,200), velocity U(-3,3)
    px/frame, bouncing at the frame border.  ``n_visible`` (optional) shows a
    seeded random subset each frame (crowd config: 150 of 200); otherwise
    ``occlude_frac`` of persons vanish for 1-5 frames now and then.
    ``long_occlude`` = H (frames; long-stream parity fixtures): on top of that, every fifth person (0, 5, 10 ...) vanishes
    once for 33-47 frames - longer than the demo's max_age = 30 (reference video_deepsort.py:18-25), so its track dies
    (deep_sort/sort/track.py:146-152) and the person returns under a new id - and persons 1, 6, 11 ... vanish once for
    12-28 frames (the confirmed track survives and is re-identified by appearance).  The windows start at frame 34 or
    later, so the first 32 frames equal the stream without the option, and end before frame H - 8."""

    def __init__(self, n_persons=30, frame_hw=(1080, 1920), seed=0, n_visible=None,
                 occlude_frac=0.05, feat_dim=512, feat_noise=0.03, long_occlude=None):
        self.rng = np.random.RandomState(seed)
        self.H, self.W = frame_hw
        self.n = n_persons
        r = self.rng
        self.wh = np.stack([r.uniform(40, 80, n_persons), r.uniform(100, 200, n_persons)], 1)
        self.pos0 = np.stack([r.uniform(0, self.W - 80, n_persons), r.uniform(0, self.H - 200, n_persons)], 1)
        self.vel = r.uniform(-3, 3, (n_persons, 2))
        self.n_visible = n_visible
        self.occlude_frac = occlude_frac
        self.feat_dim = feat_dim
        self.feat_noise = feat_noise
        base = r.randn(n_persons, feat_dim)
        self.base_feat = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(F32)
        self.patch = r.randint(0, 256, (n_persons, 32, 16, 3)).astype(np.uint8)
        self.background = r.randint(0, 256, (self.H // 8 + 1, self.W // 8 + 1, 3)).astype(np.uint8)
        self._occluded_until = np.zeros(n_persons, np.int64)
        self._seed = seed
        self.long_windows = {}                       # pid -> (first hidden frame, first visible frame again)
        if long_occlude:
            rw = np.random.RandomState((seed * 131 + 7) % (2 ** 31))
            for pid in range(n_persons):
                if pid % 5 == 0:
                    L = int(rw.randint(33, 48))
                elif pid % 5 == 1:
                    L = int(rw.randint(12, 29))
                else:
                    continue
                s0 = int(rw.randint(34, max(35, int(long_occlude) - L - 8)))
                self.long_windows[pid] = (s0, s0 + L)

    def _positions(self, t):
        span = np.array([self.W, self.H]) - self.wh
        p = self.pos0 + self.vel * t
        period = 2 * span
        p = np.mod(p, period)
        return np.where(p > span, period - p, p)

    def visible(self, t):
        r = np.random.RandomState((self._seed * 1000003 + t * 7919 + 17) % (2 ** 31))
        hidden = [pid for pid, (a, b) in self.long_windows.items() if a <= t < b]
        if self.n_visible is not None:
            if not hidden:
                return np.sort(r.choice(self.n, self.n_visible, replace=False))
            pool = np.setdiff1d(np.arange(self.n), hidden)
            return np.sort(r.choice(pool, min(self.n_visible, len(pool)), replace=False))
        vis = np.ones(self.n, bool)
        vis[hidden] = False
        # deterministic occlusions: each (person, start) pair hashed from the seed
        for pid in range(self.n):
            for back in range(5):
                s = t - back
                if s < 0:
                    continue
                rr = np.random.RandomState((self._seed * 92821 + pid * 613 + s * 31) % (2 ** 31))
                if rr.rand() < self.occlude_frac / 3.0 and back < rr.randint(1, 6):
                    vis[pid] = False
        return np.nonzero(vis)[0]

    def boxes(self, t):
        """(person ids [D], tlwh fp32 [D,4]) visible at frame t."""
        ids = self.visible(t)
        p = self._positions(t)[ids]
        return ids, np.concatenate([p, self.wh[ids]], 1).astype(F32)

    def features(self, t):
        ids, _ = self.boxes(t)
        r = np.random.RandomState((self._seed * 7 + t * 104729 + 3) % (2 ** 31))
        f = self.base_feat[ids] + self.feat_noise * r.randn(len(ids), self.feat_dim).astype(F32)
        return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(F32)

    def frame(self, t):
        """uint8 RGB [H,W,3]: blocky noise background + each visible person's texture patch."""
        img = np.repeat(np.repeat(self.background, 8, 0), 8, 1)[:self.H, :self.W].copy()
        ids, tlwh = self.boxes(t)
        for pid, (x, y, w, h) in zip(ids, tlwh):
            x0, y0 = int(x), int(y)
            x1, y1 = min(int(x + w), self.W), min(int(y + h), self.H)
            if x1 <= x0 or y1 <= y0:
                continue
            yy = ((np.arange(y0, y1) - y0) * 32 // max(y1 - y0, 1)).clip(0, 31)
            xx = ((np.arange(x0, x1) - x0) * 16 // max(x1 - x0, 1)).clip(0, 15)
            img[y0:y1, x0:x1] = self.patch[pid][yy][:, xx]
        return img


def head_injection(tlwh_frame, frame_hw, img_size, heads, cls=0, logit=6.0):
    """Rows for ``Darknet.inject`` so that decode yields the scripted persons.

    ``heads``: list of (H, W, [(aw, ah) * A]) per yolo layer, in network order.
    For each box the best-IoU anchor over all heads is chosen (Darknet's own
    assignment rule).  Returns float32 [n, 9]: head, anchor, gy, gx, tx, ty, tw, th, cls.
    The bench writes obj = cls_logit = +logit at those cells and obj = -logit
    everywhere else, leaving compute cost unchanged."""
    fh, fw = frame_hw
    mh, mw = img_size
    rows = []
    for x, y, w, h in np.asarray(tlwh_frame, dtype=np.float64):
        cx, cy = (x + w / 2) * mw / fw, (y + h / 2) * mh / fh
        bw, bh = w * mw / fw, h * mh / fh
        best = None
        for hi, (H, W, anchors) in enumerate(heads):
            for ai, (aw, ah) in enumerate(anchors):
                inter = min(bw, aw) * min(bh, ah)
                iou = inter / (bw * bh + aw * ah - inter)
                if best is None or iou > best[0]:
                    best = (iou, hi, ai, H, W, aw, ah)
        _, hi, ai, H, W, aw, ah = best
        gx, gy = cx * W / mw, cy * H / mh
        ix, iy = min(int(gx), W - 1), min(int(gy), H - 1)
        fx = min(max(gx - ix, 1e-4), 1 - 1e-4)
        fy = min(max(gy - iy, 1e-4), 1 - 1e-4)
        rows.append([hi, ai, iy, ix, np.log(fx / (1 - fx)), np.log(fy / (1 - fy)),
                     np.log(bw / aw), np.log(bh / ah), cls])
    return np.array(rows, dtype=F32).reshape(-1, 9)
