"""Generate tests/golden/*.npz by running the REAL reference (TEST ORACLE tooling).

Run in the build container only (needs /root/reference):
    python -m oracle.gen_golden [name ...]
Every fixture stores inputs (or the seed that regenerates them through
yolo_deepsort_amd.synth) and the reference's outputs.  Nothing from the
reference tree is copied; it is imported through oracle/ref_harness.py.
"""

import importlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness                                  # noqa: E402
from yolo_deepsort_amd import cfgs, synth                       # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
F32 = np.float32

MINI_CFG = """
[net]
channels=3
height=32
width=32

[convolutional]
batch_normalize=1
filters=8
size=3
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=mish

[shortcut]
from=-3
activation=linear

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=1

[route]
layers=-1
groups=2
group_id=1

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[maxpool]
size=5
stride=1

[route]
layers=-2

[maxpool]
size=9
stride=1

[route]
layers=-4

[maxpool]
size=13
stride=1

[route]
layers=-1,-3,-5,-6

[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=255
activation=linear

[yolo]
mask = 3,4,5
anchors = 10,14,  23,27,  37,58,  81,82,  135,169,  344,319
classes=80
num=6

[route]
layers = -4

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[upsample]
stride=2

[route]
layers = -1, 4

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=255
activation=linear

[yolo]
mask = 0,1,2
anchors = 10,14,  23,27,  37,58,  81,82,  135,169,  344,319
classes=80
num=6
"""

READ_KEYS = ("type", "batch_normalize", "filters", "size", "stride", "activation", "layers",
             "groups", "group_id", "from", "mask", "anchors", "classes", "channels", "height", "width")


def _save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def _tmp_write(data, suffix):
    f = tempfile.NamedTemporaryFile(suffix=suffix, delete=False)
    f.write(data if isinstance(data, bytes) else data.encode())
    f.close()
    return f.name


def _ref_darknet(ns, cfg_text, img_size, seed, obj_bias=-4.0):
    import torch
    cfg_path = _tmp_write(cfg_text, ".cfg")
    w_path = _tmp_write(synth.darknet_weights_blob(cfg_text, seed, obj_bias), ".weights")
    model = ns.models.Darknet(cfg_path, img_size=img_size)
    model.load_darknet_weights(w_path)
    model.eval()
    os.unlink(cfg_path)
    os.unlink(w_path)
    return model, torch


def gen_cfg_parse(ns):
    """a1: parse_model_config on the reference's own cfg files (keys the reference reads)."""
    out = {}
    for name in ("yolov3", "yolov3-tiny", "yolov4", "yolov4-tiny"):
        defs = ns.parse_config.parse_model_config(os.path.join(ref_harness.REF_ROOT, "config", name + ".cfg"))
        out[name] = [{k: v for k, v in d.items() if k in READ_KEYS} for d in defs]
    with open(os.path.join(GOLD, "cfg_parse.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("  wrote cfg_parse.json")


def gen_mini_darknet(ns):
    """a2/a3/a4/a5/a29: every layer type on a 32x32 input, all layer outputs checksummed."""
    model, torch = _ref_darknet(ns, MINI_CFG, (32, 32), seed=3, obj_bias=-1.0)
    rng = np.random.RandomState(11)
    x = rng.rand(1, 3, 32, 32).astype(F32)
    outs = {}
    with torch.no_grad():
        # re-run the interpreter by hand to capture every layer (forward keeps them local)
        xt = torch.from_numpy(x)
        layer_outputs, yolo = [], []
        img_dim = xt.shape[2], xt.shape[3]
        cur = xt
        for i, (d, m) in enumerate(zip(model.module_defs, model.module_list)):
            t = d["type"]
            if t in ("convolutional", "upsample", "maxpool"):
                cur = m(cur)
            elif t == "route":
                cur = torch.cat([layer_outputs[int(l)] for l in d["layers"].split(",")], 1)
                if "groups" in d:
                    cur = cur.chunk(m[0].groups, dim=1)[m[0].group_id]
            elif t == "shortcut":
                cur = layer_outputs[-1] + layer_outputs[int(d["from"])]
            elif t == "yolo":
                cur, _ = m[0](cur, None, img_dim)
                yolo.append(cur)
            layer_outputs.append(cur)
            if t != "yolo":
                outs[f"layer{i}"] = cur.numpy().copy()
        full = model(xt).numpy()
    assert np.array_equal(full, torch.cat(yolo, 1).numpy())
    _save("mini_darknet", x=x, out=full, seed=np.array(3), obj_bias=np.array(-1.0), **outs)


def gen_tiny416(ns):
    """a3 end-to-end: yolov3-tiny 416, seed-0 weights, [1,2535,85]."""
    cfg = cfgs.cfg_text("yolov3-tiny")
    model, torch = _ref_darknet(ns, cfg, (416, 416), seed=0)
    x = np.random.RandomState(0).rand(1, 3, 416, 416).astype(F32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    _save("darknet_tiny416_seed0", out=y)


QUIRK_CFG = """
[net]
channels=3
height=44
width=64

[convolutional]
batch_normalize=1
filters=8
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[maxpool]
size=2
stride=2

[convolutional]
filters=21
size=1
stride=1
pad=1
activation=linear

[yolo]
mask = 0,1,2
anchors = 4,6, 9,14, 20,12
classes=2
num=3
"""


def gen_rect(ns):
    """a5 on a NON-SQUARE model input (VERDICT r4 'missing' #3): Darknet(img_size=(h, w)) is public API (models.py:52-61)
    and YOLOLayer scales (x, y, w, h) by (s_h, s_w, s_h, s_w) with s = (img_h / grid_h, img_w / grid_w) and divides the
    anchors' (w, h) by the same pair (models.py:169-172,183,216) - x goes with the HEIGHT ratio.  Two cases:
    * yolov3-tiny at (416, 608): the realistic rectangular deployment, raw forward + ImageDetector.detect on a 640x480
      frame (cv2.resize to (608, 416), NMS, resize_boxes with the (h, w) pair).  Every net with an upsample + route needs
      both sides divisible by 32, and then s_h = s_w = the stride: the quirk is invisible there by construction.
    * QUIRK_CFG at (44, 64): three floor-mode 2x2 max-pools take 44 -> 5 rows and 64 -> 8 columns, s = (8.8, 8.0): the
      two ratios differ, the swapped pairing decides every box column."""
    arrays = {}
    cfg = cfgs.cfg_text("yolov3-tiny", 608, 416)
    model, torch = _ref_darknet(ns, cfg, (416, 608), seed=0, obj_bias=-1.0)
    x = np.random.RandomState(2).rand(1, 3, 416, 608).astype(F32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    idx, val = _sampled(y)
    names = _tmp_write(cfgs.coco_names_text(), ".names")
    det = ns.img_detect.ImageDetector(model, names, thres=0.5, nms_thres=0.4)
    os.unlink(names)
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    out = det.detect(frame)
    arrays.update(tiny_shape=np.array(y.shape), tiny_box=y[0, :, :5].copy(), tiny_idx=idx, tiny_val=val,
                  tiny_det=out.numpy() if out is not None else np.zeros((0, 6), F32))
    print(f"    rect 416x608: output {y.shape}, {arrays['tiny_det'].shape[0]} detections")
    model, torch = _ref_darknet(ns, QUIRK_CFG, (44, 64), seed=5, obj_bias=-1.0)
    x = np.random.RandomState(6).rand(2, 3, 44, 64).astype(F32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    yl = [m[0] for m in model.module_list if isinstance(m[0], ns.models.YOLOLayer)][0]
    arrays.update(quirk_out=y, quirk_scale=yl.scale.numpy().copy(), quirk_grid=np.array(yl.grid_size))
    print(f"    rect quirk net: output {y.shape}, grid {tuple(yl.grid_size)}, scale {yl.scale.numpy().ravel()}")
    _save("darknet_rect", **arrays)


def _sampled(y, n=4096, seed=5):
    idx = np.random.RandomState(seed).choice(y.size, n, replace=False)
    return idx.astype(np.int64), y.reshape(-1)[idx]


def gen_full608(ns):
    """a3: yolov3 / yolov4 @608 with seed-0 weights: per-head stats + 4096 sampled elements."""
    for name in ("yolov3", "yolov4"):
        cfg = cfgs.cfg_text(name, 608, 608)
        model, torch = _ref_darknet(ns, cfg, (608, 608), seed=0)
        x = np.random.RandomState(1).rand(1, 3, 608, 608).astype(F32)
        with torch.no_grad():
            y = model(torch.from_numpy(x)).numpy()
        idx, val = _sampled(y)
        stats = np.array([[y[0, :, c].mean(dtype=np.float64), np.abs(y[0, :, c]).max()] for c in range(85)])
        _save(f"darknet_{name}_608_seed0", idx=idx, val=val, col_stats=stats,
              shape=np.array(y.shape), obj=y[0, :, 4].copy())
        del model


def _make_pred(rng, n, n_pos, cls_choices, dup=False, tie=False):
    """Synthetic [1,n,85] decode output with n_pos clustered positive boxes."""
    p = np.zeros((1, n, 85), F32)
    p[0, :, :2] = rng.uniform(20, 580, (n, 2))
    p[0, :, 2:4] = rng.uniform(10, 200, (n, 2))
    p[0, :, 4] = rng.uniform(0, 0.3, n)
    p[0, :, 5:] = rng.uniform(0, 0.2, (n, 80))
    centers = rng.uniform(100, 500, (max(n_pos // 6, 1), 4))
    pos = rng.choice(n, n_pos, replace=False)
    for k, i in enumerate(pos):
        c = centers[k % len(centers)]
        p[0, i, :2] = c[:2] + rng.uniform(-6, 6, 2)
        p[0, i, 2:4] = np.abs(c[2:]) * 0.4 + 30 + rng.uniform(-4, 4, 2)
        p[0, i, 4] = rng.uniform(0.6, 1.0)
        cls = cls_choices[rng.randint(len(cls_choices))]
        p[0, i, 5 + cls] = rng.uniform(0.85, 1.0)
        if dup:
            p[0, i, 5 + cls_choices[(rng.randint(len(cls_choices)))]] = rng.uniform(0.85, 1.0)
    if tie:
        p[0, pos[: n_pos // 2], 4] = F32(0.75)
        p[0, pos[: n_pos // 2], 5:] = 0
        p[0, pos[: n_pos // 2], 5] = F32(1.0)
    return p


def gen_nms(ns):
    """a7 + 3P-1: soft_non_max_suppression incl. multi-label, cls=79 offset, ties, >300 kept, empty."""
    import torch
    rng = np.random.RandomState(21)
    cases = {
        "basic": (_make_pred(rng, 300, 60, [0, 2, 4]), 0.5, 0.4),
        "multilabel": (_make_pred(rng, 300, 60, [0, 1, 2, 3], dup=True), 0.5, 0.4),
        "cls79": (_make_pred(rng, 250, 80, [78, 79]), 0.5, 0.4),
        "ties": (_make_pred(rng, 250, 80, [0], tie=True), 0.5, 0.4),
        "empty": (_make_pred(rng, 100, 0, [0]), 0.5, 0.4),
        "thres03": (_make_pred(rng, 300, 90, [0, 5, 7]), 0.3, 0.6),
    }
    many = _make_pred(rng, 700, 0, [0])
    many[0, :, :2] = np.stack(np.meshgrid(np.arange(35) * 12.0, np.arange(20) * 20.0), -1).reshape(-1, 2)[:700]
    many[0, :, 2:4] = 8
    many[0, :, 4] = rng.uniform(0.6, 1, 700)
    many[0, :, 5] = 0.99
    cases["over300"] = (many, 0.5, 0.4)
    arrays = {}
    for name, (pred, ct, it) in cases.items():
        out = ns.model_build.soft_non_max_suppression(torch.from_numpy(pred.copy()), ct, it)[0]
        arrays[name + "_pred"] = pred
        arrays[name + "_thr"] = np.array([ct, it])
        arrays[name + "_out"] = out.numpy() if out is not None else np.zeros((0, 6), F32)
        print(f"    nms {name}: {0 if out is None else out.shape[0]} kept")
    _save("nms_cases", **arrays)


def gen_detect_plumbing(ns):
    """a6/a8, BASELINE cfg1: ImageDetector.detect, yolov3-tiny 416, tracker=None, 640x480 frame."""
    cfg = cfgs.cfg_text("yolov3-tiny")
    model, torch = _ref_darknet(ns, cfg, (416, 416), seed=0, obj_bias=-1.0)
    names = _tmp_write(cfgs.coco_names_text(), ".names")
    det = ns.img_detect.ImageDetector(model, names, thres=0.5, nms_thres=0.4)
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    out = det.detect(frame)
    os.unlink(names)
    n = 0 if out is None else out.shape[0]
    print(f"    plumbing: {n} detections")
    _save("detect_plumbing_640x480", out=(out.numpy() if out is not None else np.zeros((0, 6), F32)),
          obj_bias=np.array(-1.0))


def gen_reid(ns):
    """a11/a12/a13/a29: crops + Extractor on a synthetic ckpt (zip format written by torch.save)."""
    import torch
    sd = synth.reid_state_dict(0)
    path = _tmp_write(b"", ".t7")
    torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.0, "epoch": 0}, path)
    ex = ns.feature_extractor.Extractor(path, use_cuda=False)
    os.unlink(path)
    scene = synth.PersonScene(8, seed=4)
    frame = scene.frame(0)
    _, tlwh = scene.boxes(0)
    tlwh[0, :2] = (-5.5, -3.2)                       # clipped at the top-left corner
    tlwh[1, 0] = 1920 - 30.0                         # clipped at the right edge
    ds = ns.deep_sort.DeepSort(ex, use_cuda=False)
    ds.height, ds.width = frame.shape[:2]
    crops = [ds._s_tlwh_to_xyxy(b) for b in torch.from_numpy(tlwh)]
    feats = ds._get_features(torch.from_numpy(tlwh), frame).numpy()
    pre = ex._preprocess([frame[y1:y2, x1:x2] for x1, y1, x2, y2 in crops]).numpy()
    _save("reid_seed0", tlwh=tlwh, crops=np.array(crops, np.int32), feats=feats,
          pre_sample=pre[:, :, ::16, ::8].copy())


def gen_kalman(ns):
    """a15-a18, a24: the kalman_filter.py:259-273 scenario + 200 random tracks."""
    import torch
    kf = ns.kalman_filter.KalmanFilter()
    m0, c0 = kf.initiate(torch.tensor([10, 15, .5, 10], dtype=torch.float32))
    m1, c1 = kf.predict(m0, c0)
    m2, c2 = kf.update(m1, c1, torch.tensor([[12, 20, .6, 11]], dtype=torch.float32))
    rng = np.random.RandomState(8)
    T, D = 200, 150
    xyah = np.stack([rng.uniform(0, 1900, T), rng.uniform(0, 1000, T), rng.uniform(0.3, 0.6, T),
                     rng.uniform(80, 220, T)], 1).astype(F32)
    means, covs = [], []
    for i in range(T):
        m, c = kf.initiate(torch.from_numpy(xyah[i]))
        means.append(m)
        covs.append(c)
    mean = torch.cat(means, 0)
    cov = torch.cat(covs, 0)
    init_mean, init_cov = mean.numpy().copy(), cov.numpy().copy()
    steps = {}
    for s in range(3):
        mean, cov = kf.predict(mean, cov)
        steps[f"pred{s}_mean"], steps[f"pred{s}_cov"] = mean.numpy().copy(), cov.numpy().copy()
        z = (mean[:, :4] + torch.from_numpy((rng.randn(T, 4) * [3, 3, 0.01, 3]).astype(F32))).float()
        steps[f"z{s}"] = z.numpy().copy()
        mean, cov = kf.update(mean, cov, z)
        steps[f"upd{s}_mean"], steps[f"upd{s}_cov"] = mean.numpy().copy(), cov.numpy().copy()
    meas = np.stack([rng.uniform(0, 1900, D), rng.uniform(0, 1000, D), rng.uniform(0.3, 0.6, D),
                     rng.uniform(80, 220, D)], 1).astype(F32)
    meas[:T // 2:2] = (mean[:T // 2:2, :4].numpy() + rng.randn(len(meas[:T // 2:2]), 4) * [4, 4, 0.01, 2]).astype(F32)[:len(meas[:T // 2:2])]
    g2 = kf.gating_distance(mean, cov, torch.from_numpy(meas), True).numpy()
    g4 = kf.gating_distance(mean, cov, torch.from_numpy(meas), False).numpy()
    _save("kalman", ka_m0=m0.numpy(), ka_c0=c0.numpy(), ka_m1=m1.numpy(), ka_c1=c1.numpy(),
          ka_m2=m2.numpy(), ka_c2=c2.numpy(), xyah=xyah, init_mean=init_mean, init_cov=init_cov,
          meas=meas, gate2=g2, gate4=g4, **steps)


class _FakeExtractor:
    """Feeds scripted features to the real DeepSort (bypasses the ReID CNN)."""

    def __init__(self):
        self.next = None

    def __call__(self, crops):
        import torch
        assert len(crops) == len(self.next)
        return torch.from_numpy(self.next)


def run_reference_trace(ns, scene, n_frames, params, drop_frames=(), empty_frames=(), metric=None, boxes_of=None):
    """Drive the real DeepSort.update frame by frame; record everything observable.
    metric="euclidean": the reference's DeepSort hard-wires "cosine" (deep_sort.py:34) and its euclidean helper is called
    with one argument too many (nn_matching.py:56 vs :187 -> TypeError), so the trace is taken with the metric object
    swapped for NearestNeighborDistanceMetric("euclidean", ...) whose _metric applies the reference's OWN
    _nn_euclidean_distance per track segment - a harness-side adapter, flagged in the fixture name."""
    import torch
    ex = _FakeExtractor()
    ds = ns.deep_sort.DeepSort(ex, use_cuda=False, **params)
    if metric == "euclidean":
        nn = sys.modules[type(ds.tracker.metric).__module__]
        m = nn.NearestNeighborDistanceMetric("euclidean", params["max_dist"], params["nn_budget"])
        m._metric = lambda x, y, bp: torch.stack([nn._nn_euclidean_distance(x[bp[i]:bp[i + 1]], y) for i in range(len(bp) - 1)], 0)
        ds.tracker.metric = m
    pre = sys.modules.get(type(ds).__module__)
    orders = []
    if params.get("nms_max_overlap", 1.0) != 1:
        # record the np.argsort(scores) the reference observed (a constant vector: numpy's choice, platform dependent)
        orig_nms = pre.non_max_suppression

        def nms_spy(boxes, thr, scores=None):
            orders.append(np.argsort(scores).astype(np.int32))
            return orig_nms(boxes, thr, scores)
        pre.non_max_suppression = nms_spy
    rec = []
    orig_match = ds.tracker._match
    last = {}

    def spy(dets):
        r = orig_match(dets)
        last["m"] = r
        return r
    ds.tracker._match = spy
    frame = np.zeros((scene.H, scene.W, 3), np.uint8)
    for t in range(n_frames):
        if t in drop_frames:            # detector returned None: tracker not called (video_detect.py:137)
            rec.append(None)
            continue
        ids, tlwh = scene.boxes(t) if boxes_of is None else boxes_of(t)
        feats = scene.features(t) if boxes_of is None else boxes_of(t, feats=True)
        if t in empty_frames:           # class mask emptied the list: called with D = 0
            tlwh, feats, ids = tlwh[:0], feats[:0], ids[:0]
        ex.next = feats
        payload = torch.from_numpy((ids % 3 * 2).astype(F32))
        n_orders = len(orders)
        out = ds.update(torch.from_numpy(tlwh), torch.ones(len(tlwh)), frame, payload)
        m, ut, ud = last["m"]
        tr = ds.tracker.tracks
        extra = {"nms_order": orders[-1]} if len(orders) > n_orders else {}
        rec.append(dict(extra, 
            out=np.array(out, dtype=np.int32).reshape(-1, 6),
            matches=np.array(m, dtype=np.int32).reshape(-1, 2),
            um_t=np.array(sorted(ut), dtype=np.int32), um_d=np.array(ud, dtype=np.int32),
            ids=np.array([x.track_id for x in tr], np.int32), state=np.array([x.state for x in tr], np.int32),
            tsu=np.array([x.time_since_update for x in tr], np.int32),
            hits=np.array([x.hits for x in tr], np.int32),
            mean=(torch.cat([x.mean for x in tr], 0).numpy() if tr else np.zeros((0, 8), F32))))
    return rec


TRACE_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


def _pack_trace(rec):
    arrays = {}
    for t, r in enumerate(rec):
        if r is None:
            arrays[f"f{t}_skipped"] = np.array(1)
            continue
        for k, v in r.items():
            arrays[f"f{t}_{k}"] = v
    arrays["n_frames"] = np.array(len(rec))
    return arrays


def gen_track_traces(ns):
    """a9-a11, a14-a28 end to end with given features."""
    s30 = synth.PersonScene(30, seed=0, occlude_frac=0.15)
    rec = run_reference_trace(ns, s30, 60, TRACE_PARAMS, drop_frames=(20, 21), empty_frames=(35,))
    _save("track_trace_30", **_pack_trace(rec))
    short = dict(TRACE_PARAMS, max_age=4)
    rec = run_reference_trace(ns, synth.PersonScene(12, seed=2, occlude_frac=0.6), 60, short)
    _save("track_trace_12_maxage4", **_pack_trace(rec))
    crowd = synth.PersonScene(200, seed=0, n_visible=150)
    rec = run_reference_trace(ns, crowd, 40, TRACE_PARAMS)
    for r in rec:                         # keep the fixture small: drop the means
        if r is not None:
            r.pop("mean")
    _save("track_trace_200x150", **_pack_trace(rec))


def gen_tiled(ns):
    """8f row 1: tiled sliding-window detection + the merge=True NMS branch as it really behaves."""
    import contextlib
    import io
    import torch
    cfg = cfgs.cfg_text("yolov3-tiny")
    names = _tmp_write(cfgs.coco_names_text(), ".names")
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    arrays = {}
    # -1.3: some boxes suppressed (plain NMS stands); -1.4: nothing suppressed -> every box collapses (here onto NaN:
    # the elementwise IoU of score-ordered vs original-ordered boxes is 0 everywhere, so the weights sum to 0)
    for tag, ob in (("a", -1.3), ("b", -1.4)):
        model, torch = _ref_darknet(ns, cfg, (416, 416), seed=0, obj_bias=ob)
        det = ns.img_detect.ImageDetector(model, names, thres=0.5, nms_thres=0.4, win_size=(416, 416), overlap=0.15)
        with contextlib.redirect_stdout(io.StringIO()):        # the reference prints tensors from its bare except
            out = det.detect(frame)
        print(f"    tiled obj_bias {ob}: {0 if out is None else out.shape[0]} detections")
        arrays["tiled_out_" + tag] = out.numpy() if out is not None else np.zeros((0, 6), F32)
        arrays["obj_bias_" + tag] = np.array(ob)
    os.unlink(names)
    # merge branch unit cases on corner-form predictions
    rng = np.random.RandomState(33)

    def pred_of(boxes, n=60):
        p = np.zeros((1, n, 85), F32)
        p[0, :, 4] = 0.01
        for k, (x1, y1, x2, y2, c, obj, cls) in enumerate(boxes):
            p[0, k * 3, :4] = (x1, y1, x2, y2)
            p[0, k * 3, 4] = obj
            p[0, k * 3, 5 + c] = cls
        return p
    cases = {
        "all_kept": pred_of([(10, 10, 60, 80, 0, .9, .95), (200, 50, 260, 150, 2, .8, .95), (400, 300, 470, 420, 0, .7, .95)]),
        "one_kept": pred_of([(100, 100, 200, 220, 0, .9, .95), (104, 98, 205, 226, 0, .8, .9), (96, 103, 196, 215, 0, .7, .92),
                             (101, 101, 199, 219, 0, .85, .6)]),
        "plain": pred_of([(100, 100, 200, 220, 0, .9, .95), (104, 98, 205, 226, 0, .8, .9), (400, 300, 470, 420, 0, .7, .95),
                          (402, 301, 468, 424, 0, .75, .9), (20, 20, 50, 60, 4, .9, .9)]),
        "single": pred_of([(10, 10, 60, 80, 0, .9, .95)]),
    }
    for name, p in cases.items():
        with contextlib.redirect_stdout(io.StringIO()):
            o = ns.model_build.soft_non_max_suppression(torch.from_numpy(p.copy()), 0.5, 0.4, merge=True, is_p1p2=True)[0]
        arrays[name + "_pred"] = p
        arrays[name + "_out"] = o.numpy() if o is not None else np.zeros((0, 6), F32)
        print(f"    merge {name}: {arrays[name + '_out'].shape[0]} rows")
    _save("tiled_detect", **arrays)


def gen_track_options(ns):
    """NearestNeighborDistanceMetric / DeepSort options beside the demo's defaults: budget=None
    (nn_matching.py:152-154), tracker-side NMS (preprocessing.py:6-73 via deep_sort.py:52-57), the euclidean helpers
    (nn_matching.py:4-27,56-74)."""
    import torch
    nn = importlib.import_module("deep_sort.sort.nn_matching")
    pre = importlib.import_module("deep_sort.sort.preprocessing")
    # -- unbounded gallery
    unb = dict(TRACE_PARAMS, nn_budget=None)
    rec = run_reference_trace(ns, synth.PersonScene(10, seed=5, occlude_frac=0.3), 80, unb)
    _save("track_trace_budget_none", **_pack_trace(rec))
    # -- tracker-side NMS: duplicated (jittered) boxes make it bite; > 16 detections so that argsort is not trivially sorted
    scene = synth.PersonScene(14, seed=6, occlude_frac=0.1)
    rng = np.random.RandomState(3)
    jit = rng.uniform(-4, 4, (200, 14, 4)).astype(F32)

    def boxes_of(t, feats=False):
        ids, tlwh = scene.boxes(t)
        dup = (tlwh + jit[t, :len(ids)]).astype(F32)
        if feats:
            f = scene.features(t)
            return np.concatenate([f, f], 0)
        return np.concatenate([ids, ids]), np.concatenate([tlwh, dup], 0)
    rec = run_reference_trace(ns, scene, 40, dict(TRACE_PARAMS, nms_max_overlap=0.6), boxes_of=boxes_of)
    assert any("nms_order" in r for r in rec)
    arrays = _pack_trace(rec)
    arrays["jitter"] = jit[:40]
    _save("track_trace_nms06", **arrays)
    # -- euclidean metric through the adapter described in run_reference_trace (max_dist on squared distances of unit vectors)
    rec = run_reference_trace(ns, synth.PersonScene(20, seed=7, occlude_frac=0.2), 50, dict(TRACE_PARAMS, max_dist=0.6), metric="euclidean")
    _save("track_trace_euclidean_adapter", **_pack_trace(rec))
    # -- unit vectors for the helpers themselves
    arrays = {}
    g = rng.randn(37, 512).astype(F32)
    f = rng.randn(11, 512).astype(F32)
    f[3] = g[5]                                                     # an exact zero distance (clamp at 0)
    seg = np.array([0, 1, 6, 20, 37], np.int32)
    arrays.update(euc_gallery=g, euc_feats=f, euc_seg=seg,
                  euc_out=torch.stack([nn._nn_euclidean_distance(torch.from_numpy(g[seg[i]:seg[i + 1]]), torch.from_numpy(f)) for i in range(4)], 0).numpy(),
                  pdist=nn._pdist(torch.from_numpy(g[:6]), torch.from_numpy(f)).numpy())
    boxes = np.concatenate([rng.uniform(0, 300, (40, 2)), rng.uniform(20, 120, (40, 2))], 1).astype(F32)
    for k, (thr, scores) in enumerate([(0.5, np.ones(40)), (0.3, rng.uniform(0, 1, 40)), (0.9, np.ones(40)), (0.0, np.ones(40))]):
        arrays[f"nms{k}_order"] = np.argsort(scores).astype(np.int32)
        arrays[f"nms{k}_thr"] = np.array(thr)
        arrays[f"nms{k}_pick"] = np.array(pre.non_max_suppression(boxes, thr, scores), np.int32)
    arrays["nms_boxes"] = boxes
    _save("track_options_units", **arrays)


BENCH_DS_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)      # video_deepsort.py:18-25
BENCH_SHAPES = {                                  # bench.py CONFIGS (BASELINE.json configs[1], [2], [4])
    "cfg2": dict(net="yolov3", persons=30, visible=None),
    "cfg3": dict(net="yolov4", persons=30, visible=None),
    "cfg5": dict(net="yolov4", persons=200, visible=150),
}


class MarginSpy:
    """Records, inside the reference's own association (tracker.py:56-92), how close each frame's DECISIONS came to their
    thresholds - the robustness number of the long-stream fixtures (VERDICT r4 'next' #1).  Patches three module attributes
    the reference resolves at call time (KalmanFilter.gating_distance, linear_assignment.gate_cost_matrix,
    iou_matching.iou_cost) with pass-through wrappers; nothing the reference computes is altered.
      cos:  min |cosine cost - max_dist| over the entries whose gate passes (a gated-out entry is rejected either way);
      gate: min |d2 - chi2inv95[2]| over the entries whose cosine cost passes (linear_assignment.py:52 clamps every
            rejected entry to the same value, so the gate decides nothing where the appearance already rejects);
      iou:  min |iou cost - max_iou_distance| over the IOU stage's entries;
      lsap_eps: the largest eps of (1e-3, 1e-4, 1e-5) such that scipy's assignment of the clamped appearance matrix is
            unchanged under four uniform(-eps, eps) perturbations of the admissible entries (0 = none of them)."""

    def __init__(self, ns, max_dist, max_iou):
        self.ns, self.max_dist, self.max_iou = ns, max_dist, max_iou
        self.gate_thr = float(ns.kalman_filter.chi2inv95[2])
        self.frame = None
        self._orig = None

    def __enter__(self):
        ns, spy = self.ns, self
        KF, la, iou = ns.kalman_filter.KalmanFilter, ns.linear_assignment, ns.iou_matching
        self._orig = (KF.gating_distance, la.gate_cost_matrix, iou.iou_cost)
        o_gd, o_gate, o_iou = self._orig

        def gating_distance(kf, *a, **k):
            d = o_gd(kf, *a, **k)
            spy._gd = d.detach().cpu().numpy().astype(np.float64)
            return d

        def gate_cost_matrix(kf, cost_matrix, *a, **k):
            cos = cost_matrix.detach().cpu().numpy().astype(np.float64).copy()
            out = o_gate(kf, cost_matrix, *a, **k)
            spy._appearance(cos, spy._gd)
            return out

        def iou_cost(*a, **k):
            c = o_iou(*a, **k)
            m = c.detach().cpu().numpy().astype(np.float64)
            if m.size:
                spy.frame["iou"] = min(spy.frame["iou"], float(np.abs(m - spy.max_iou).min()))
            return c
        KF.gating_distance, la.gate_cost_matrix, iou.iou_cost = gating_distance, gate_cost_matrix, iou_cost
        return self

    def __exit__(self, *exc):
        ns = self.ns
        ns.kalman_filter.KalmanFilter.gating_distance, ns.linear_assignment.gate_cost_matrix, ns.iou_matching.iou_cost = self._orig

    def begin_frame(self):
        self.frame = dict(cos=np.inf, gate=np.inf, iou=np.inf, lsap_eps=1e-3)
        return self.frame

    def _appearance(self, cos, gd):
        from scipy.optimize import linear_sum_assignment
        f = self.frame
        gate_ok, cos_ok = gd <= self.gate_thr, cos <= self.max_dist
        if gate_ok.any():
            f["cos"] = min(f["cos"], float(np.abs(cos - self.max_dist)[gate_ok].min()))
        if cos_ok.any():
            f["gate"] = min(f["gate"], float(np.abs(gd - self.gate_thr)[cos_ok].min()))
        adm = gate_ok & cos_ok
        clamped = np.where(adm, cos, self.max_dist + 1e-5)
        base = linear_sum_assignment(clamped)
        base = {(int(r), int(c)) for r, c in zip(*base) if adm[r, c]}
        rng = np.random.RandomState(cos.size)
        stable = 0.0
        for eps in (1e-5, 1e-4, 1e-3):
            ok = True
            for _ in range(4):
                pert = np.where(adm, np.minimum(cos + rng.uniform(-eps, eps, cos.shape), self.max_dist), clamped)
                got = linear_sum_assignment(pert)
                if {(int(r), int(c)) for r, c in zip(*got) if adm[r, c]} != base:
                    ok = False
                    break
            if not ok:
                break
            stable = eps
        f["lsap_eps"] = min(f["lsap_eps"], stable)


def run_reference_stream(ns, cfg_name, n_frames, seed=0, sample=512, long_occlude=None, compact=False):
    """The reference's hot loop (yolo3/detect/video_detect.py:134-157) at the BENCHMARKED shape: 1920x1080 frames of
    bench.py's synthetic stream -> ImageDetector.detect (cv2.resize shim, Darknet 608x608, soft_non_max_suppression,
    resize_boxes) -> class mask [0, 2, 4] -> p1p2Toxywh -> DeepSort.update with the real Extractor on a synthetic
    ckpt.t7.  The only harness-side intervention is bench.py's logit injection (synthetic weights cannot see
    the scripted persons), applied to the raw head tensor in front of YOLOLayer.forward exactly as inject_*_kernel does."""
    import torch
    from functools import reduce
    from oracle.pipeline import make_injector                  # reused for its row semantics only (numpy on a torch buffer)
    cfg = BENCH_SHAPES[cfg_name]
    S = 608
    cfg_text = cfgs.cfg_text(cfg["net"], S, S)
    model, _ = _ref_darknet(ns, cfg_text, S, seed=0)
    names = _tmp_write(cfgs.coco_names_text(), ".names")
    det = ns.img_detect.ImageDetector(model, names, thres=0.5, nms_thres=0.4)
    os.unlink(names)
    sd = synth.reid_state_dict(0)
    ck = _tmp_write(b"", ".t7")
    torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.0, "epoch": 0}, ck)
    ds = ns.deep_sort.DeepSort(ck, use_cuda=False, **BENCH_DS_PARAMS)
    os.unlink(ck)
    scene = synth.PersonScene(cfg["persons"], seed=seed, n_visible=cfg["visible"], long_occlude=long_occlude)
    yolo = [m[0] for m in model.module_list if isinstance(m[0], ns.models.YOLOLayer)]
    heads = []
    hw = {32: S // 32, 16: S // 16, 8: S // 8}
    for d in model.module_defs:
        if d["type"] == "yolo":
            idx = [int(v) for v in d["mask"].split(",")]
            a = [int(v) for v in d["anchors"].split(",")]
            heads.append([None, None, [(a[2 * j], a[2 * j + 1]) for j in idx]])
    state = {"rows": None, "pred": None}
    YL = ns.models.YOLOLayer
    orig_fwd = YL.forward

    def fwd(self, x, targets=None, img_dim=None):
        hi = yolo.index(self)
        rows, logit = state["rows"], 6.0
        A, attrs = self.num_anchors, self.num_classes + 5
        v = x.view(x.shape[0], A, attrs, x.shape[2], x.shape[3])
        v[:, :, 4] = -logit
        for r in rows[rows[:, 0] == hi]:
            a, gy, gx, cls = int(r[1]), int(r[2]), int(r[3]), int(r[8])
            cell = torch.full((attrs,), -logit)
            cell[:4] = torch.from_numpy(r[4:8].copy())
            cell[4] = logit
            cell[5 + cls] = logit
            v[0, a, :, gy, gx] = cell
        return orig_fwd(self, x, targets, img_dim)
    orig_model_fwd = type(model).forward

    def model_fwd(self, x, targets=None):
        out = orig_model_fwd(self, x, targets)
        state["pred"] = out.detach().numpy().copy()
        return out
    # head geometry (H, W per yolo layer) from one dry pass
    with torch.no_grad():
        shapes = []
        hook = [m.register_forward_pre_hook(lambda mod, inp: shapes.append((yolo.index(mod), inp[0].shape[2], inp[0].shape[3]))) for m in yolo]
        model(torch.zeros(1, 3, S, S))
        for h in hook:
            h.remove()
    for hi, H, W in shapes:
        heads[hi][0], heads[hi][1] = H, W
    heads = [tuple(h) for h in heads]
    YL.forward, type(model).forward = fwd, model_fwd
    rng = np.random.RandomState(99)
    n_boxes = sum(3 * h * w for h, w, _ in heads)
    sample_idx = np.sort(rng.choice(n_boxes * 85, sample, replace=False))
    arrays = {"sample_idx": sample_idx, "n_frames": np.array(n_frames)}
    if compact:
        return _run_compact(ns, cfg_name, n_frames, scene, det, ds, heads, S, state, (YL, orig_fwd, model, orig_model_fwd))
    try:
        for t in range(n_frames):
            frame = scene.frame(t)
            state["rows"] = synth.head_injection(scene.boxes(t)[1], (scene.H, scene.W), (S, S), heads, cls=0)
            detections = det.detect(frame)                                         # video_detect.py:135
            arrays[f"f{t}_pred"] = state["pred"].reshape(-1)[sample_idx]
            # force the rows into the sample too: the sampled boxes that carry persons
            arrays[f"f{t}_det"] = detections.numpy().copy() if detections is not None else np.zeros((0, 6), F32)
            out = None
            if detections is not None:                                             # :137-149
                boxs = ns.model_build.p1p2Toxywh(detections[:, :4])
                class_ids = detections[:, -1]
                confidences = detections[:, 4]
                mask = reduce(lambda a, b: a | b, [class_ids == m for m in (0, 2, 4)])
                boxs, confidences, class_ids = boxs[mask], confidences[mask], class_ids[mask]
                out = ds.update(boxs.float(), confidences, frame, class_ids)
            arrays[f"f{t}_none"] = np.array(out is None)
            arrays[f"f{t}_out"] = np.array(out if out is not None else [], dtype=np.int32).reshape(-1, 6)
            arrays[f"f{t}_ids"] = np.array([x.track_id for x in ds.tracker.tracks], np.int32)
            arrays[f"f{t}_state"] = np.array([x.state for x in ds.tracker.tracks], np.int32)
            print(f"    {cfg_name} frame {t}: {arrays[f'f{t}_det'].shape[0]} detections, {arrays[f'f{t}_out'].shape[0]} rows", flush=True)
    finally:
        YL.forward, type(model).forward = orig_fwd, orig_model_fwd
    return arrays


def _run_compact(ns, cfg_name, n_frames, scene, det, ds, heads, S, state, restore):
    """The long-stream form of run_reference_stream's loop: the same statements per frame (video_detect.py:134-157), stored
    compactly - int32 rows, track ids / states / hits / time_since_update, gallery fill - with MarginSpy's decision margins."""
    from functools import reduce
    YL, orig_fwd, model, orig_model_fwd = restore
    out_rows, out_ptr, ids, ids_ptr, states, tsu = [], [0], [], [0], [], []
    none, n_det, at_budget = [], [], []
    margins = {k: [] for k in ("cos", "gate", "iou", "lsap_eps")}
    budget = BENCH_DS_PARAMS["nn_budget"]
    try:
        with MarginSpy(ns, BENCH_DS_PARAMS["max_dist"], BENCH_DS_PARAMS["max_iou_distance"]) as spy:
            for t in range(n_frames):
                frame = scene.frame(t)
                state["rows"] = synth.head_injection(scene.boxes(t)[1], (scene.H, scene.W), (S, S), heads, cls=0)
                m = spy.begin_frame()
                detections = det.detect(frame)                                         # video_detect.py:135
                out = None
                if detections is not None:                                             # :137-149
                    boxs = ns.model_build.p1p2Toxywh(detections[:, :4])
                    class_ids = detections[:, -1]
                    confidences = detections[:, 4]
                    mask = reduce(lambda a, b: a | b, [class_ids == c for c in (0, 2, 4)])
                    boxs, confidences, class_ids = boxs[mask], confidences[mask], class_ids[mask]
                    out = ds.update(boxs.float(), confidences, frame, class_ids)
                none.append(out is None)
                n_det.append(0 if detections is None else len(detections))
                rows = np.array(out if out is not None else [], dtype=np.int32).reshape(-1, 6)
                out_rows.append(rows)
                out_ptr.append(out_ptr[-1] + len(rows))
                tr = ds.tracker.tracks
                ids += [x.track_id for x in tr]
                states += [x.state for x in tr]
                tsu += [x.time_since_update for x in tr]
                ids_ptr.append(len(ids))
                at_budget.append(sum(1 for v in ds.tracker.metric.samples.values() if len(v) >= budget))
                for k in margins:
                    margins[k].append(m[k])
                print(f"    {cfg_name} frame {t}: {n_det[-1]} detections, {len(rows)} rows, {len(tr)} tracks, next id {ds.tracker._next_id}, "
                      f"margins cos {m['cos']:.2e} gate {m['gate']:.2e} iou {m['iou']:.2e} lsap {m['lsap_eps']:.0e}", flush=True)
    finally:
        YL.forward, type(model).forward = orig_fwd, orig_model_fwd
    arrays = dict(n_frames=np.array(n_frames), out_rows=np.concatenate(out_rows, 0), out_ptr=np.array(out_ptr, np.int32),
                  none=np.array(none), n_det=np.array(n_det, np.int32), ids=np.array(ids, np.int32), ids_ptr=np.array(ids_ptr, np.int32),
                  state=np.array(states, np.int8), time_since_update=np.array(tsu, np.int16), at_budget=np.array(at_budget, np.int32),
                  windows=np.array([[p, a, b] for p, (a, b) in sorted(scene.long_windows.items())], np.int32).reshape(-1, 3))
    for k, v in margins.items():
        arrays["margin_" + k] = np.array(v, np.float64)
    return arrays


LONG_STREAMS = {"cfg2": 256, "cfg3": 128, "cfg5": 96}      # frames (VERDICT r4 'next' #1)


def gen_long_stream(ns):
    """Long-stream parity at the benchmarked shape: the reference's hot loop over 256 / 128 / 96 frames of the cfg2 / cfg3 /
    cfg5 streams with synth.PersonScene's long occlusion windows (confirmed tracks die of max_age = 30, persons return under new
    ids, tracks are re-identified after 12-28 hidden frames, galleries run into nn_budget = 30)."""
    which = os.environ.get("YDS_LONG_STREAMS", "cfg2,cfg3,cfg5").split(",")
    for name in which:
        n = LONG_STREAMS[name]
        _save(f"long_stream_{name}", **run_reference_stream(ns, name, n, long_occlude=n, compact=True))


class CountingActions:
    """Stands in for ActionIdentify in the video_detect fixture (the generator only calls .update(rows), video_detect.py:151-152):
    the returned value depends on the call count and the row count only, so it pins WHEN the generator calls it and what it
    yields on frames where it does not (mirrored in tests/test_gpu_video_detect.py)."""

    def __init__(self):
        self.calls = 0

    def update(self, rows):
        self.calls += 1
        return [[self.calls, len(rows)]]


VIDEO_CASES = {
    # name: constructor arguments of VideoDetector (video_detect.py:42-76) + stream script
    "tracker_skip2_mask": dict(skip_frames=2, class_mask=[0, 2], tracker=True, action=True, empty=(8, 9, 10, 11), n=28, skip_secs=0),
    "tracker_every_frame": dict(skip_frames=-1, class_mask=[0, 2], tracker=True, action=True, empty=(5, 6, 13), n=20, skip_secs=0),
    "no_tracker_skip3": dict(skip_frames=3, class_mask=[0, 2], tracker=False, action=False, empty=(6, 7, 8), n=18, skip_secs=0),
    "tracker_skip_secs": dict(skip_frames=2, class_mask=None, tracker=True, action=False, empty=(), n=20, skip_secs=1.9),
}
VIDEO_SCENE = dict(persons=8, frame_hw=(360, 640), seed=5, fps=4.5, img=416, net="yolov3-tiny", thres=0.5, nms_thres=0.4)


def video_clip(case):
    """Frames (RGB), per-frame scripted boxes and classes of a video_detect case (shared with the GPU test)."""
    c, sc = VIDEO_CASES[case], VIDEO_SCENE
    scene = synth.PersonScene(sc["persons"], frame_hw=sc["frame_hw"], seed=sc["seed"], occlude_frac=0.0)
    frames, boxes, classes = [], [], []
    for t in range(c["n"]):
        frames.append(scene.frame(t))
        ids, tlwh = scene.boxes(t)
        if t in c["empty"]:
            ids, tlwh = ids[:0], tlwh[:0]
        boxes.append(np.asarray(tlwh, np.float64).reshape(-1, 4))
        classes.append((np.asarray(ids) % 3).astype(F32))          # persons carry classes 0, 1, 2: the mask [0, 2] drops a third
    return np.stack(frames, 0), boxes, classes


def video_injection(boxes, classes, heads):
    sc = VIDEO_SCENE
    rows = synth.head_injection(boxes, sc["frame_hw"], (sc["img"], sc["img"]), heads, cls=0)
    rows[:, 8] = classes
    return rows


def gen_video_detect(ns):
    """a9: the reference's OWN generator, VideoDetector.detect (video_detect.py:78-208), executed unmodified over
    scripted clips - skip gate (:134), detector-None frames (:137), class mask (:141-147), action call (:151-154), hold (:156,
    161-170), skip_secs seek (:98-101), output writer (:103-106,188-189).  Harness side: the FileVideoStream / VideoCapture shims
    serve the clip from memory, cv2 drawing calls are no-ops, and the bench's head-logit injection is applied in front of
    YOLOLayer.forward (synthetic weights cannot see the scripted persons)."""
    import torch
    sc = VIDEO_SCENE
    S = sc["img"]
    cfg_text = cfgs.cfg_text(sc["net"], S, S)
    model, _ = _ref_darknet(ns, cfg_text, (S, S), seed=0)
    names = _tmp_write(cfgs.coco_names_text(), ".names")
    sd = synth.reid_state_dict(0)
    ck = _tmp_write(b"", ".t7")
    torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.0, "epoch": 0}, ck)
    yolo = [m[0] for m in model.module_list if isinstance(m[0], ns.models.YOLOLayer)]
    heads = []
    for d in model.module_defs:
        if d["type"] == "yolo":
            idx = [int(v) for v in d["mask"].split(",")]
            a = [int(v) for v in d["anchors"].split(",")]
            heads.append([None, None, [(a[2 * j], a[2 * j + 1]) for j in idx]])
    with torch.no_grad():
        shapes = []
        hook = [m.register_forward_pre_hook(lambda mod, inp: shapes.append((yolo.index(mod), inp[0].shape[2], inp[0].shape[3]))) for m in yolo]
        model(torch.zeros(1, 3, S, S))
        for h in hook:
            h.remove()
    for hi, H, W in shapes:
        heads[hi][0], heads[hi][1] = H, W
    heads = [tuple(h) for h in heads]
    state = {"rows": None}
    YL = ns.models.YOLOLayer
    orig_fwd = YL.forward

    def fwd(self, x, targets=None, img_dim=None):
        hi = yolo.index(self)
        rows, logit = state["rows"], 6.0
        A, attrs = self.num_anchors, self.num_classes + 5
        v = x.view(x.shape[0], A, attrs, x.shape[2], x.shape[3])
        v[:, :, 4] = -logit
        for r in rows[rows[:, 0] == hi]:
            a, gy, gx, cls = int(r[1]), int(r[2]), int(r[3]), int(r[8])
            cell = torch.full((attrs,), -logit)
            cell[:4] = torch.from_numpy(r[4:8].copy())
            cell[4] = logit
            cell[5 + cls] = logit
            v[0, a, :, gy, gx] = cell
        return orig_fwd(self, x, targets, img_dim)
    YL.forward = fwd
    arrays = {"heads_hw": np.array([[h, w] for h, w, _ in heads], np.int32)}
    try:
        for case, c in VIDEO_CASES.items():
            frames, boxes, classes = video_clip(case)
            inj = [video_injection(b, k, heads) for b, k in zip(boxes, classes)]
            served = []

            def on_read(t):
                state["rows"] = inj[t]
                served.append(t)
            ref_harness.CLIPS["clip://" + case] = dict(frames=frames[..., ::-1], fps=sc["fps"], on_read=on_read)   # BGR, like a decoded video
            tracker = ns.deep_sort.DeepSort(ck, use_cuda=False, **BENCH_DS_PARAMS) if c["tracker"] else None
            act = CountingActions() if c["action"] else None
            vd = ns.video_detect.VideoDetector(model, names, thres=sc["thres"], nms_thres=sc["nms_thres"], skip_frames=c["skip_frames"],
                                               class_mask=c["class_mask"], tracker=tracker, action_id=act)
            sys.modules["cv2"].VideoWriter.written.clear()
            n = 0
            for result, hold, actions in vd.detect("clip://" + case, output_path="out-" + case, skip_secs=c["skip_secs"], show_fps=False):
                assert result.shape == frames.shape[1:] and result.dtype == np.uint8
                arrays[f"{case}_f{n}_none"] = np.array(hold is None)
                if hold is None:
                    h = np.zeros((0, 6), F32)
                elif c["tracker"]:
                    h = np.array(hold, dtype=np.int32).reshape(-1, 6)
                else:
                    h = hold.numpy().astype(F32).reshape(-1, 6)
                arrays[f"{case}_f{n}_hold"] = h
                arrays[f"{case}_f{n}_actions"] = np.array(json.dumps(actions))
                n += 1
            w = sys.modules["cv2"].VideoWriter.written
            assert len(w) == 1 and w[0].frames == n
            arrays[f"{case}_n"] = np.array(n)
            arrays[f"{case}_served"] = np.array(served, np.int32)                    # source frame index of every yield (skip_secs seek)
            arrays[f"{case}_writer_fps"] = np.array(w[0].args[2])
            arrays[f"{case}_writer_size"] = np.array(w[0].args[3], np.int32)
            nn = sum(1 for i in range(n) if arrays[f"{case}_f{i}_none"])
            print(f"    {case}: {n} yields (first source frame {served[0]}), {nn} with hold None, rows per yield "
                  f"{[arrays[f'{case}_f{i}_hold'].shape[0] for i in range(n)]}")
    finally:
        YL.forward = orig_fwd
        os.unlink(names)
        os.unlink(ck)
    _save("video_detect", **arrays)


def gen_bench_shape(ns):
    """parity AT the benchmarked shape (batch 16 x 1080p, 608x608, 2 steps) from the reference itself."""
    which = os.environ.get("YDS_BENCH_SHAPES", "cfg2,cfg3,cfg5,cfg4s1").split(",")
    for name in which:
        if name == "cfg4s1":         # BASELINE configs[3]: cfg3 on every rank, stream seed = rank; seed 0 is bench_shape_cfg3, this is rank 1's stream
            _save("bench_shape_cfg4_seed1", **run_reference_stream(ns, "cfg3", 16, seed=1))
            continue
        arrays = run_reference_stream(ns, name, 32)
        _save(f"bench_shape_{name}", **arrays)


def action_scene(n_frames=40):
    """The scripted int32 tracker rows the action fixture is recorded on (shared with tests/test_host_logic.py)."""
    rng = np.random.RandomState(0)
    pos = rng.randint(0, 500, (6, 2)).astype(float)
    vel = rng.randint(-6, 7, (6, 2)).astype(float)
    frames = []
    for t in range(n_frames):
        pos += vel
        if t % 7 == 0:
            vel = rng.randint(-6, 7, (6, 2)).astype(float)
        vis = [i for i in range(6) if rng.rand() > 0.25]
        frames.append(np.array([[pos[i, 0], pos[i, 1], pos[i, 0] + 20, pos[i, 1] + 40, i + 1, (i % 2) * 2] for i in vis], np.int32).reshape(-1, 6))
    return frames


def gen_action(ns):
    """action/ (SURVEY 8f row 2): ActionIdentify.update over 40 frames of scripted tracker rows with a stepped clock (the
    module reads time.time() in Orbit.update: 40 ms per call here and in the test), including the speed-based FastCrossing."""
    import json
    import time
    import action.action_Identify as rai
    import action.actions as ra
    assert ra.__file__.startswith(ref_harness.REF_ROOT)
    tick = [1000.0]

    def fake_time():
        tick[0] += 0.04
        return tick[0]
    real = time.time
    time.time = fake_time
    try:
        acts = [ra.TakeOff(0, (1, 2)), ra.Landing(0, (1, 2)), ra.Glide(0, (2, 4)), ra.FastCrossing(2, 0.05), ra.BreakInto(0, 2), ra.BreakInto(2, 1)]
        A = rai.ActionIdentify(acts, max_age=3, max_size=4)
        out = []
        for det in action_scene():
            out.append([[int(a), int(b), str(c)] for a, b, c in A.update(det)])
        none_case = A.update(None)
    finally:
        time.time = real
    with open(os.path.join(GOLD, "action_trace.json"), "w") as f:
        json.dump(dict(frames=out, none_returns_none=none_case is None), f)
    print("    action events:", sum(len(o) for o in out))


# ---- wide dynamic range (VERDICT r5 'next' #2): BatchNorm statistics over decades, black / saturated inputs --------------------
WIDE_RES_CFG = """
[net]
channels=3
height=64
width=64

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=64
size=3
stride=2
pad=1
activation=leaky
""" + """
[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=64
size=3
stride=1
pad=1
activation=leaky

[shortcut]
from=-3
activation=linear
""" * 4 + """
[convolutional]
batch_normalize=1
filters=128
size=3
stride=2
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=255
activation=linear

[yolo]
mask = 0,1,2
anchors = 10,14,  23,27,  37,58,  81,82,  135,169,  344,319
classes=80
num=6
"""
WIDE_SEED = 21
WIDE_TRACE = dict(persons=10, frame_hw=(360, 640), seed=7, occlude_frac=0.2, frames=64)
WIDE_DS_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


def wide_inputs(size, seed=2):
    """[random, all-black, saturated] images as the detector sees them (NCHW in [0, 1])."""
    x = np.random.RandomState(seed).rand(3, 3, size, size).astype(F32)
    x[1] = 0.0
    x[2] = 1.0
    return x


def wide_reid_frame():
    """A 360 x 640 frame of the trace's scene with a black and a white patch, and eight boxes (two inside the patches, two clipped)."""
    sc = WIDE_TRACE
    scene = synth.PersonScene(sc["persons"], frame_hw=sc["frame_hw"], seed=sc["seed"], occlude_frac=0.0)
    frame = scene.frame(0).copy()
    frame[40:200, 60:140] = 0
    frame[40:200, 300:380] = 255
    _, tlwh = scene.boxes(0)
    tlwh = tlwh[:8].astype(F32).copy()
    tlwh[0] = (62.0, 44.0, 70.0, 150.0)              # all black
    tlwh[1] = (303.0, 45.0, 72.0, 148.0)             # all white
    tlwh[2, :2] = (-6.5, -2.2)                       # clipped top-left
    tlwh[3, 0] = 640 - 25.0                          # clipped right
    return frame, tlwh


def _layer_samples(model, torch, x, per_layer, seed):
    """Output of every convolutional block (conv + BN + activation) of the reference model on x, `per_layer` sampled elements each."""
    taken, hooks = {}, []
    for i, (d, m) in enumerate(zip(model.module_defs, model.module_list)):
        if d["type"] == "convolutional":
            hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: taken.__setitem__(i, out.detach().numpy().copy())))
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    for h in hooks:
        h.remove()
    out = {}
    for i, t in taken.items():
        idx = np.sort(np.random.RandomState(seed + i).choice(t.size, min(per_layer, t.size), replace=False)).astype(np.int64)
        out[f"L{i}_idx"], out[f"L{i}_val"] = idx, t.reshape(-1)[idx]
        out[f"L{i}_absmax"], out[f"L{i}_absmin"] = np.abs(t).max(), np.abs(t[t != 0]).min() if (t != 0).any() else np.float32(0)
    return y, out


def gen_wide_range(ns):
    """f16x3 where it could break: yolov3-tiny 416, a 12-conv residual net and the ReID net with BatchNorm statistics over decades
    (synth profile "wide": running_var log-uniform 1e-3 .. 1e2, gamma U(0, 2) with exact zeros, |beta| <= 3, |mean| <= 3 sigma),
    inputs incl. an all-black and a saturated image; plus a 64-frame id trace of the reference's DeepSort with the REAL Extractor on
    those ReID weights."""
    import torch
    arrays = {}
    # (1) the residual net: every conv block sampled, three inputs
    model, _ = _ref_darknet_profile(ns, WIDE_RES_CFG, (64, 64), WIDE_SEED, -1.0)
    x = wide_inputs(64)
    y, lay = _layer_samples(model, torch, x, 1024, 100)
    arrays.update({"res_" + k: v for k, v in lay.items()})
    arrays["res_out"] = y
    print("    residual net: out", y.shape, "finite", bool(np.isfinite(y).all()),
          "layer |x| ranges", [(float(lay[k.replace('_idx', '_absmin')]), float(lay[k.replace('_idx', '_absmax')])) for k in lay if k.endswith("_idx")][:12])
    # (2) yolov3-tiny 416
    cfg = cfgs.cfg_text("yolov3-tiny")
    model, _ = _ref_darknet_profile(ns, cfg, (416, 416), WIDE_SEED + 1, -1.0)
    x = wide_inputs(416)
    y, lay = _layer_samples(model, torch, x, 512, 200)
    arrays.update({"tiny_" + k: v for k, v in lay.items()})
    idx = np.sort(np.random.RandomState(9).choice(y.size, 6144, replace=False)).astype(np.int64)
    arrays["tiny_idx"], arrays["tiny_val"], arrays["tiny_shape"] = idx, y.reshape(-1)[idx], np.array(y.shape)
    arrays["tiny_obj"] = y[:, :, 4].copy()
    print("    tiny416: out", y.shape, "finite", bool(np.isfinite(y).all()), "|out| max", float(np.abs(y[np.isfinite(y)]).max()))
    del model
    # (3) the ReID net: embeddings of eight crops (a black one, a white one, two clipped)
    sd = synth.reid_state_dict(WIDE_SEED, "wide")
    ck = _tmp_write(b"", ".t7")
    torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.0, "epoch": 0}, ck)
    ex = ns.feature_extractor.Extractor(ck, use_cuda=False)
    frame, tlwh = wide_reid_frame()
    ds = ns.deep_sort.DeepSort(ex, use_cuda=False)
    ds.height, ds.width = frame.shape[:2]
    feats = ds._get_features(torch.from_numpy(tlwh), frame).numpy()
    arrays["reid_tlwh"], arrays["reid_feats"] = tlwh, feats
    g = feats @ feats.T
    print("    reid: feats", feats.shape, "finite", bool(np.isfinite(feats).all()), "cosine between crops min/max off-diagonal",
          float((g - 2 * np.eye(len(g))).max()), float((g + 2 * np.eye(len(g))).min()))
    # (4) 64 frames of DeepSort.update with the real Extractor on the wide weights (boxes scripted, no detector)
    sc = WIDE_TRACE
    scene = synth.PersonScene(sc["persons"], frame_hw=sc["frame_hw"], seed=sc["seed"], occlude_frac=sc["occlude_frac"])
    ds = ns.deep_sort.DeepSort(ck, use_cuda=False, **WIDE_DS_PARAMS)
    os.unlink(ck)
    rows, ptr, ids, ids_ptr, states = [], [0], [], [0], []
    margins = {k: [] for k in ("cos", "gate", "iou", "lsap_eps")}
    with MarginSpy(ns, WIDE_DS_PARAMS["max_dist"], WIDE_DS_PARAMS["max_iou_distance"]) as spy:
        for t in range(sc["frames"]):
            f = scene.frame(t)
            pid, b = scene.boxes(t)
            m = spy.begin_frame()
            out = ds.update(torch.from_numpy(b.astype(F32)), torch.ones(len(b)), f, torch.from_numpy((pid % 3).astype(F32)))
            r = np.array(out, dtype=np.int32).reshape(-1, 6)
            rows.append(r)
            ptr.append(ptr[-1] + len(r))
            ids += [x.track_id for x in ds.tracker.tracks]
            states += [x.state for x in ds.tracker.tracks]
            ids_ptr.append(len(ids))
            for k in margins:
                margins[k].append(m[k])
    arrays.update(trace_rows=np.concatenate(rows, 0), trace_ptr=np.array(ptr, np.int32), trace_ids=np.array(ids, np.int32),
                  trace_ids_ptr=np.array(ids_ptr, np.int32), trace_state=np.array(states, np.int8),
                  **{"trace_margin_" + k: np.array(v, np.float64) for k, v in margins.items()})
    print("    trace: rows", ptr[-1], "next id", ds.tracker._next_id, "min margins",
          {k: float(np.min(v)) for k, v in margins.items()})
    _save("wide_range", **arrays)


WIDE_CSP_CFG = """
[net]
channels=3
height=64
width=64

[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=64
size=3
stride=2
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=64
size=1
stride=1
pad=1
activation=mish

[route]
layers = -2

[convolutional]
batch_normalize=1
filters=64
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=64
size=3
stride=1
pad=1
activation=mish

[shortcut]
from=-3
activation=linear

[convolutional]
batch_normalize=1
filters=64
size=1
stride=1
pad=1
activation=mish

[route]
layers = -1,-7

[convolutional]
batch_normalize=1
filters=64
size=1
stride=1
pad=1
activation=mish

[convolutional]
batch_normalize=1
filters=128
size=3
stride=2
pad=1
activation=mish

[route]
layers=-1
groups=2
group_id=1

[convolutional]
batch_normalize=1
filters=64
size=3
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=255
activation=linear

[yolo]
mask = 0,1,2
anchors = 10,14,  23,27,  37,58,  81,82,  135,169,  344,319
classes=80
num=6
"""


def gen_wide_mish(ns):
    """The yolov4 constructs under the wide BatchNorm statistics (round 6, after gen_wide_range): Mish everywhere, a CSP split (two 1x1
    convolutions reading one tensor - merged into one launch by the planner), a shortcut, a two-source route and a grouped route;
    random / all-black / saturated inputs; every conv block sampled + the decoded head."""
    import torch
    model, _ = _ref_darknet_profile(ns, WIDE_CSP_CFG, (64, 64), WIDE_SEED + 2, -1.0)
    x = wide_inputs(64, seed=3)
    y, lay = _layer_samples(model, torch, x, 1024, 300)
    print("    CSP / Mish net: out", y.shape, "finite", bool(np.isfinite(y).all()),
          "layer |x| max", [round(float(lay[k.replace('_idx', '_absmax')]), 2) for k in lay if k.endswith("_idx")])
    _save("wide_range_mish", out=y, **lay)


def _ref_darknet_profile(ns, cfg_text, img_size, seed, obj_bias):
    import torch
    cfg_path = _tmp_write(cfg_text, ".cfg")
    w_path = _tmp_write(synth.darknet_weights_blob(cfg_text, seed, obj_bias, profile="wide"), ".weights")
    model = ns.models.Darknet(cfg_path, img_size=img_size)
    model.load_darknet_weights(w_path)
    model.eval()
    os.unlink(cfg_path)
    os.unlink(w_path)
    return model, torch



ALL = dict(long_stream=gen_long_stream, wide=gen_wide_range, wide_mish=gen_wide_mish, rect=gen_rect, video_detect=gen_video_detect, action=gen_action, bench_shape=gen_bench_shape, options=gen_track_options, tiled=gen_tiled, cfg_parse=gen_cfg_parse, mini=gen_mini_darknet, tiny416=gen_tiny416, full608=gen_full608,
           nms=gen_nms, plumbing=gen_detect_plumbing, reid=gen_reid, kalman=gen_kalman,
           traces=gen_track_traces)


def main(argv):
    names = argv or list(ALL)
    ns = ref_harness.import_reference()
    os.makedirs(GOLD, exist_ok=True)
    for n in names:
        print(f"[{n}]")
        ALL[n](ns)


if __name__ == "__main__":
    main(sys.argv[1:])
