"""HIP NMS / ReID / Kalman / LSAP / tracker (through the C ABI) vs oracle + reference golden vectors."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from yolo_deepsort_amd import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
RTOL, ATOL = 1e-3, 1e-3


def _lib():
    from yolo_deepsort_amd import _lib
    _lib.init(0)
    return _lib


# ----------------------------------------------------------------------------------------- NMS
def _nms(pred, ct, it):
    L = _lib()
    pred = np.ascontiguousarray(pred, dtype=F32)
    out = np.zeros((300, 6), F32)
    n = C.c_int(0)
    L.check(L.load().yds_nms_pred(L.ptr(pred), pred.shape[0], pred.shape[1], ct, it, L.ptr(out), 300, C.byref(n)))
    return out[:n.value]


def test_nms_golden_cases_bit_exact():
    from oracle import nms as onms
    g = golden("nms_cases")
    names = sorted({k[:-5] for k in g.files if k.endswith("_pred")})
    for nme in names:
        ct, it = (float(v) for v in g[nme + "_thr"])
        got = _nms(g[nme + "_pred"][0], ct, it)
        ref = g[nme + "_out"]
        assert got.shape == ref.shape, nme
        assert np.array_equal(got, ref), nme
        want = onms.soft_non_max_suppression(g[nme + "_pred"], ct, it)[0]
        if want is None:
            assert got.shape[0] == 0
        else:
            assert np.array_equal(got, want), nme


def test_nms_random_vs_oracle():
    from oracle import nms as onms
    rng = np.random.RandomState(3)
    for trial in range(6):
        n = int(rng.choice([1, 37, 500, 3000]))
        p = np.zeros((1, n, 85), F32)
        p[0, :, :2] = rng.uniform(0, 608, (n, 2))
        p[0, :, 2:4] = rng.uniform(5, 250, (n, 2))
        p[0, :, 4] = rng.uniform(0, 1, n) ** 3
        p[0, :, 5:] = rng.uniform(0, 1, (n, 80)) ** 6
        want = onms.soft_non_max_suppression(p, 0.4, 0.45)[0]
        got = _nms(p[0], 0.4, 0.45)
        if want is None:
            assert got.shape[0] == 0
        else:
            assert np.array_equal(got, want), trial


def test_detect_plumbing_cfg1_golden():
    """BASELINE cfg1 through the drop-in ImageDetector: yolov3-tiny 416, tracker=None, 640x480 frame."""
    import os
    import tempfile
    from yolo_deepsort_amd import cfgs
    from yolo_deepsort_amd.detect import ImageDetector
    from yolo_deepsort_amd.models import Darknet
    g = golden("detect_plumbing_640x480")
    cfg = cfgs.cfg_text("yolov3-tiny")
    net = Darknet(None, img_size=(416, 416), cfg_text=cfg)
    net.load_darknet_weights(None, blob=synth.darknet_weights_blob(cfg, 0, float(g["obj_bias"])))
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
    det = ImageDetector(net, f.name, thres=0.5, nms_thres=0.4)
    os.unlink(f.name)
    assert det.num_classes == 80
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    out = det.detect(frame)
    out = out.numpy() if hasattr(out, "numpy") else out
    ref = g["out"]
    assert out.shape == ref.shape and ref.shape[0] > 0
    assert np.array_equal(out[:, 5], ref[:, 5])
    np.testing.assert_allclose(out, ref, rtol=RTOL, atol=ATOL)
    # a frame the detector finds nothing in -> None, exactly when the oracle's NMS returns None (img_detect.py:87-91)
    from oracle.darknet import DarknetOracle
    from oracle import nms as onms
    from oracle.resize import resize_bilinear_u8
    ora = DarknetOracle(cfg, 416, is_text=True)
    ora.load_weights_array(np.frombuffer(synth.darknet_weights_blob(cfg, 0, -30.0), dtype=F32, offset=20))
    net.load_darknet_weights(None, blob=synth.darknet_weights_blob(cfg, 0, -30.0))       # objectness bias -30: nothing passes
    x = resize_bilinear_u8(frame, (416, 416)).astype(F32).transpose(2, 0, 1)[None] / F32(255.)
    assert onms.soft_non_max_suppression(ora(x), 0.5, 0.4)[0] is None
    assert det.detect(frame) is None


def _nms_merge(pred, ct, it):
    L = _lib()
    pred = np.ascontiguousarray(pred, dtype=F32)
    out = np.zeros((300, 6), F32)
    n = C.c_int(0)
    L.check(L.load().yds_nms_merge_pred(L.ptr(pred), pred.shape[0], pred.shape[1], ct, it, L.ptr(out), 300, C.byref(n)))
    return out[:n.value]


def test_nms_merge_branch_golden_and_oracle():
    """soft_non_max_suppression(merge=True, is_p1p2=True): ids/scores bit-exact, merged boxes within 1e-4 px."""
    from oracle import nms as onms
    g = golden("tiled_detect")
    for nme in ("all_kept", "one_kept", "plain", "single"):
        got = _nms_merge(g[nme + "_pred"][0], 0.5, 0.4)
        ref = g[nme + "_out"]
        assert got.shape == ref.shape, nme
        assert np.array_equal(got[:, 4:], ref[:, 4:]), nme
        np.testing.assert_allclose(got[:, :4], ref[:, :4], rtol=1e-6, atol=1e-4, err_msg=nme)
    rng = np.random.RandomState(8)
    seen = set()
    for trial in range(40):
        n = int(rng.choice([2, 3, 5, 40]))
        p = np.zeros((1, n, 85), F32)
        if trial % 2:                                   # one tight cluster -> a single survivor
            p[0, :, :2] = 300 + rng.uniform(-3, 3, (n, 2))
            p[0, :, 2:4] = p[0, :, :2] + 120 + rng.uniform(-3, 3, (n, 2))
        else:
            p[0, :, :2] = rng.uniform(0, 1500, (n, 2))
            p[0, :, 2:4] = p[0, :, :2] + rng.uniform(5, 60, (n, 2))
        p[0, :, 4] = rng.uniform(0.3, 1, n)
        p[0, np.arange(n), 5 + (0 if trial % 2 else rng.randint(0, 3, n))] = rng.uniform(0.6, 1, n)
        want = onms.soft_non_max_suppression_merge(p, 0.5, 0.4)[0]
        got = _nms_merge(p[0], 0.5, 0.4)
        if want is None:
            assert got.shape[0] == 0
            continue
        assert got.shape == want.shape, trial
        assert np.array_equal(got[:, 4:], want[:, 4:]), trial
        assert np.array_equal(np.isnan(got), np.isnan(want)), trial
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-6, atol=1e-3, equal_nan=True, err_msg=str(trial))
        b = p[..., :4].astype(np.float64)
        centre = np.stack([(b[..., 0] + b[..., 2]) / 2, (b[..., 1] + b[..., 3]) / 2, b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]], -1)
        plain = onms.soft_non_max_suppression(np.concatenate([centre.astype(F32), p[..., 4:]], -1), 0.5, 0.4)[0]
        seen.add("merged" if not np.allclose(np.nan_to_num(want[:, :4]), plain[:, :4], atol=1e-3) else "plain")
    assert seen == {"merged", "plain"}


def _tiled_detector(obj_bias, batch_max):
    import os
    import tempfile
    from yolo_deepsort_amd import cfgs
    from yolo_deepsort_amd.detect import ImageDetector
    from yolo_deepsort_amd.models import Darknet
    cfg = cfgs.cfg_text("yolov3-tiny")
    net = Darknet(None, img_size=(416, 416), cfg_text=cfg, batch_max=batch_max)
    net.load_darknet_weights(None, blob=synth.darknet_weights_blob(cfg, 0, obj_bias))
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
    det = ImageDetector(net, f.name, thres=0.5, nms_thres=0.4, win_size=(416, 416), overlap=0.15)
    os.unlink(f.name)
    return det


def test_tiled_detection_golden():
    """ImageDetector(win_size=...) on a 640x480 frame = 4 windows, vs the reference's own output."""
    g = golden("tiled_detect")
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    for tag in ("a", "b"):
        ref = g["tiled_out_" + tag]
        outs = []
        for batch_max in (4, 3, 1):                       # windows in one batch, and in chunks
            out = _tiled_detector(float(g["obj_bias_" + tag]), batch_max).detect(frame)
            out = out.numpy() if hasattr(out, "numpy") else out
            assert out.shape == ref.shape, (tag, batch_max)
            assert np.array_equal(out[:, 5], ref[:, 5])
            np.testing.assert_allclose(out, ref, rtol=RTOL, atol=ATOL, equal_nan=True)
            outs.append(out)
        # (the windows of one frame in one batch or in chunks: the same boxes; low bits may differ - a batch of one window
        #  takes the split-K kernels on its deep layers, which add the K ranges in another order)
        for o in outs[1:]:
            np.testing.assert_allclose(o, outs[0], rtol=1e-5, atol=1e-4, equal_nan=True)
    # a frame smaller than the window takes the plain path (img_detect.py:67)
    small = np.random.RandomState(1).randint(0, 256, (300, 400, 3)).astype(np.uint8)
    det = _tiled_detector(-1.3, 1)
    a = det.detect(small)
    det.win_size = None
    b = det.detect(small)
    assert (a is None and b is None) or np.array_equal(np.asarray(a), np.asarray(b))


def test_tiled_detection_vs_oracle_odd_frame():
    """ragged windows (frame not a multiple of win_size, windows clipped at the border) vs the oracle."""
    from oracle.darknet import DarknetOracle
    from oracle.tiled import detect_tiled, windows
    from yolo_deepsort_amd import cfgs
    cfg = cfgs.cfg_text("yolov3-tiny")
    ob = -1.3
    frame = np.random.RandomState(4).randint(0, 256, (531, 977, 3)).astype(np.uint8)
    assert len(windows(531, 977, (416, 416), 0.15)) == 6
    ref = DarknetOracle(cfg, 416, is_text=True)
    ref.load_weights_array(np.frombuffer(synth.darknet_weights_blob(cfg, 0, ob), dtype=F32, offset=20))
    want = detect_tiled(ref, frame, (416, 416), 0.15, 0.5, 0.4)
    out = _tiled_detector(ob, 4).detect(frame)
    out = out.numpy() if hasattr(out, "numpy") else out
    assert want is not None and out.shape == want.shape
    assert np.array_equal(out[:, 5], want[:, 5])
    np.testing.assert_allclose(out, want, rtol=RTOL, atol=ATOL, equal_nan=True)


# ----------------------------------------------------------------------------------------- ReID
def test_reid_golden_and_oracle():
    from oracle import reid as oreid
    from yolo_deepsort_amd.deep_sort import Extractor
    g = golden("reid_seed0")
    sd = synth.reid_state_dict(0)
    ex = Extractor(sd)
    scene = synth.PersonScene(8, seed=4)
    frame = scene.frame(0)
    tlwh = g["tlwh"]
    pre = ex.preprocess(frame, tlwh)
    assert np.array_equal(pre, oreid.preprocess_crops(frame, tlwh))          # crop + resize + normalise: bit exact
    np.testing.assert_allclose(pre[:, :, ::16, ::8], g["pre_sample"], rtol=0, atol=1e-6)
    feats = ex.embed(frame, tlwh)
    np.testing.assert_allclose(feats, g["feats"], rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(feats, oreid.reid_forward(pre, sd), rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    # reference call convention: list of crops
    crops = [frame[y1:y2, x1:x2] for x1, y1, x2, y2 in g["crops"]]
    f2 = ex(crops)
    f2 = f2.numpy() if hasattr(f2, "numpy") else f2
    np.testing.assert_allclose(f2, feats, rtol=1e-5, atol=1e-6)


def test_reid_crop_resize_paths_bit_exact_and_growth():
    """Crop geometries that reach every branch of cv2's resize: generic up/down-scaling, the exact-2x INTER_AREA route
    (128x256 crop), the same-size copy (64x128), 1-pixel-wide crops, clipped boxes; more crops than max_crops (grows)."""
    from oracle import reid as oreid
    from yolo_deepsort_amd.deep_sort import Extractor
    sd = synth.reid_state_dict(0)
    ex = Extractor(sd, max_crops=4)
    rng = np.random.RandomState(12)
    frame = rng.randint(0, 256, (540, 960, 3)).astype(np.uint8)
    tlwh = np.array([[100, 50, 128, 256], [300, 200, 64, 128], [10, 10, 1.5, 300], [500, 100, 37, 91], [-20, -30, 90, 200],
                     [900, 400, 200, 300], [400, 300, 300, 2.2], [7.9, 8.9, 63.2, 127.2], [600, 20, 20, 40]], F32)
    pre = ex.preprocess(frame, tlwh)
    assert np.array_equal(pre, oreid.preprocess_crops(frame, tlwh))
    feats = ex.embed(frame, tlwh)
    np.testing.assert_allclose(feats, oreid.reid_forward(pre, sd), rtol=RTOL, atol=1e-5)


def test_reid_larger_batch_vs_oracle():
    from oracle import reid as oreid
    from yolo_deepsort_amd.deep_sort import Extractor
    sd = synth.reid_state_dict(1)
    ex = Extractor(sd, max_crops=64)
    x = np.random.RandomState(2).randn(37, 3, 128, 64).astype(F32)
    np.testing.assert_allclose(ex.forward(x), oreid.reid_forward(x, sd), rtol=RTOL, atol=1e-5)
    # one and three crops: the last stages have fewer rows (8x4 = 32 per crop) than one 256-row tile
    for d in (1, 3):
        np.testing.assert_allclose(ex.forward(x[:d]), oreid.reid_forward(x[:d], sd), rtol=RTOL, atol=1e-5)


def test_kalman_kernels_vs_golden():
    L = _lib()
    lib = L.load()
    g = golden("kalman")
    mean, cov = g["init_mean"].copy(), g["init_cov"].copy()
    T = mean.shape[0]
    m1, c1 = g["ka_m0"].copy(), g["ka_c0"].copy()
    L.check(lib.yds_kalman_predict(L.ptr(m1), L.ptr(c1), 1))
    assert np.array_equal(m1, g["ka_m1"]) and np.array_equal(c1, g["ka_c1"])     # predict is exact (2-term sums)
    z = np.array([[12, 20, .6, 11]], F32)
    L.check(lib.yds_kalman_update(L.ptr(m1), L.ptr(c1), L.ptr(z), 1))
    np.testing.assert_allclose(m1, g["ka_m2"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c1, g["ka_c2"], rtol=1e-4, atol=1e-6)
    for s in range(3):
        L.check(lib.yds_kalman_predict(L.ptr(mean), L.ptr(cov), T))
        np.testing.assert_allclose(mean, g[f"pred{s}_mean"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(cov, g[f"pred{s}_cov"], rtol=1e-4, atol=1e-5)
        zs = np.ascontiguousarray(g[f"z{s}"])
        L.check(lib.yds_kalman_update(L.ptr(mean), L.ptr(cov), L.ptr(zs), T))
        np.testing.assert_allclose(mean, g[f"upd{s}_mean"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(cov, g[f"upd{s}_cov"], rtol=1e-3, atol=1e-4)
    meas = np.ascontiguousarray(g["meas"])
    out = np.zeros((T, meas.shape[0]), F32)
    L.check(lib.yds_kalman_gating(L.ptr(mean), L.ptr(cov), T, L.ptr(meas), meas.shape[0], L.ptr(out)))
    np.testing.assert_allclose(out, g["gate2"], rtol=1e-3, atol=1e-3)


def test_cost_kernels_vs_oracle():
    from oracle import tracker as otrk
    L = _lib()
    lib = L.load()
    rng = np.random.RandomState(4)
    T, D = 23, 41
    tb = np.concatenate([rng.uniform(0, 1800, (T, 2)), rng.uniform(20, 200, (T, 2))], 1).astype(F32)
    db = np.concatenate([rng.uniform(0, 1800, (D, 2)), rng.uniform(20, 200, (D, 2))], 1).astype(F32)
    db[:10] = tb[:10] + rng.uniform(-5, 5, (10, 4)).astype(F32)
    db[10] = tb[10]                                       # identical boxes: IoU > 1 quirk (asymmetric +1)
    out = np.zeros((T, D), F32)
    L.check(lib.yds_iou_cost(L.ptr(tb), T, L.ptr(db), D, L.ptr(out)))
    # the kernel sees tracks as xyah means; rebuild tlwh the way Track.to_tlwh does
    m = otrk.tlwh_to_xyah(tb)
    tb2 = m.copy(); tb2[:, 2] *= tb2[:, 3]; tb2[:, :2] -= tb2[:, 2:] / F32(2)
    want = (F32(1) - otrk.iou_matrix(tb2, db)).astype(F32)
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-6)
    assert out[10, 10] < 0                               # IoU exceeds 1 for identical boxes
    # cosine nearest-neighbour cost over ragged galleries
    seg = np.concatenate([[0], np.cumsum(rng.randint(1, 31, T))]).astype(np.int32)
    gal = rng.randn(seg[-1], 512).astype(F32)
    feats = rng.randn(D, 512).astype(F32)
    feats[:5] = gal[seg[:5]] + 0.05 * rng.randn(5, 512).astype(F32)
    out = np.zeros((T, D), F32)
    L.check(lib.yds_cosine_min_cost(L.ptr(gal), L.ptr(seg), T, L.ptr(feats), D, 512, L.ptr(out)))
    dist = otrk.cosine_distance(gal, feats)
    want = np.stack([dist[seg[k]:seg[k + 1]].min(0) for k in range(T)], 0)
    np.testing.assert_allclose(out, want, rtol=1e-4, atol=2e-6)


# ----------------------------------------------------------------------------------------- LSAP
def _lsap(c):
    L = _lib()
    c = np.ascontiguousarray(c, dtype=F32)
    n = min(c.shape)
    r, k = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    m = C.c_int(0)
    L.check(L.load().yds_lsap(L.ptr(c), c.shape[0], c.shape[1], L.ptr(r), L.ptr(k), C.byref(m)))
    return r[:m.value], k[:m.value]


def test_tracker_grows_past_its_initial_capacity():
    """300 detections in the first frame start 300 tracks: more than the initial 256 slots (capacity doubles in place),
    and the next frames associate them (problem sizes beyond the single-wavefront LSAP)."""
    from yolo_deepsort_amd.deep_sort import _TrackerHandle
    from oracle import tracker as otrk
    _lib()
    rng = np.random.RandomState(3)
    n = 300
    trk = _TrackerHandle(0.3, 0.7, 30, 3, 30)
    ora = otrk.TrackerOracle(**TRACE_PARAMS)
    base = np.concatenate([rng.uniform(0, 1800, (n, 1)), rng.uniform(0, 900, (n, 1)), rng.uniform(40, 80, (n, 1)), rng.uniform(100, 200, (n, 1))], 1)
    feats = rng.randn(n, 512).astype(F32)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    for t in range(5):
        tlwh = (base + np.array([2.0 * t, 1.0 * t, 0, 0])).astype(F32)
        f = (feats + 0.02 * rng.randn(n, 512)).astype(F32)
        pay = (np.arange(n) % 3).astype(F32)
        out = trk.step(tlwh, f, pay)
        want = np.array(ora.update(tlwh, f, pay), np.int32).reshape(-1, 6)
        st = trk.state()
        assert np.array_equal(st["ids"], ora.state()["ids"]) and np.array_equal(st["state"], ora.state()["state"]), t
        assert out.shape == want.shape and np.array_equal(out[:, 4:], want[:, 4:]), t
        assert np.abs(out[:, :4] - want[:, :4]).max(initial=0) <= 1, t
    assert len(st["ids"]) >= 300 and out.shape[0] > 250


def test_lsap_bit_exact_vs_scipy_and_oracle():
    from scipy.optimize import linear_sum_assignment
    from oracle import clib
    rng = np.random.RandomState(0)
    for trial in range(300):
        nr, nc = rng.randint(1, 40, 2)
        kind = trial % 4
        if kind == 0:
            c = rng.rand(nr, nc).astype(F32)
        elif kind == 1:
            c = rng.randint(0, 3, (nr, nc)).astype(F32)
        elif kind == 2:
            c = rng.rand(nr, nc).astype(F32)
            c[c > 0.3] = F32(0.30001)
        else:
            c = np.full((nr, nc), 0.70001, F32)
            m = rng.rand(nr, nc) < 0.2
            c[m] = rng.rand(m.sum())
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = _lsap(c)
        r2, c2 = clib.lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (trial, nr, nc)
        assert np.array_equal(r2, r1) and np.array_equal(c2, c1)
    # kernel forms: one wavefront (<= 64 columns, above), workgroup with state + cost in LDS (200 x 150), state in LDS and
    # cost in global memory (500 x 400), state in the global scratch buffer (> ~3200 rows / columns: no size limit)
    for shape in ((200, 150), (150, 200), (1, 300), (300, 1), (257, 257), (64, 64), (65, 64), (64, 65), (500, 400), (3300, 6), (5, 3400)):
        c = np.full(shape, 0.30001, F32)
        m = rng.rand(*shape) < 0.03
        c[m] = rng.rand(m.sum()) * 0.3
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = _lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), shape
    assert _lsap(np.ones((4, 4)))[1].tolist() == [0, 1, 2, 3]
    # the register-resident workgroup form (64 < columns <= 256): every tie structure again at those sizes, tall and wide
    for trial in range(60):
        nr, nc = (int(v) for v in rng.randint(1, 257, 2))
        if max(nr, nc) <= 64:
            nc = 65 + trial
        kind = trial % 4
        if kind == 0:
            c = rng.rand(nr, nc).astype(F32)
        elif kind == 1:
            c = rng.randint(0, 3, (nr, nc)).astype(F32)
        elif kind == 2:
            c = rng.rand(nr, nc).astype(F32)
            c[c > 0.3] = F32(0.30001)
        else:
            c = np.full((nr, nc), 0.70001, F32)
            m = rng.rand(nr, nc) < 0.05
            c[m] = rng.rand(m.sum())
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = _lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (trial, nr, nc, kind)
    for shape in ((256, 256), (256, 1), (1, 256), (255, 256), (65, 65), (200, 150)):
        c = np.full(shape, F32(0.5))
        r0, c0 = linear_sum_assignment(c)                       # all ties: the order of scipy's scan decides everything
        r1, c1 = _lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), shape


# ----------------------------------------------------------------------------------------- traces
TRACE_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


from conftest import check_int_rows as _check_int_rows  # noqa: E402


def _run_trace(scene, g, params, drop=(), empty=(), frame_of=None):
    from yolo_deepsort_amd.deep_sort import _TrackerHandle
    from oracle import tracker as otrk
    _lib()
    trk = _TrackerHandle(params["max_dist"], params["max_iou_distance"], params["max_age"], params["n_init"], params["nn_budget"],
                         params.get("metric", "cosine"))
    ora = otrk.TrackerOracle(**params)
    n = int(g["n_frames"])
    stats = [0, 0]
    for t in range(n):
        if f"f{t}_skipped" in g.files:
            assert t in drop
            continue
        if frame_of is not None:
            ids, tlwh, feats = frame_of(t)
        else:
            ids, tlwh = scene.boxes(t)
            feats = scene.features(t)
        if t in empty:
            tlwh, feats, ids = tlwh[:0], feats[:0], ids[:0]
        payload = (ids % 3 * 2).astype(F32)
        out, matches = trk.step(tlwh, feats, payload, want_debug=True)
        want = np.array(ora.update(tlwh, feats, payload), dtype=np.int32).reshape(-1, 6)
        um_t, um_d = trk.last_unmatched()
        st = trk.state()
        assert np.array_equal(matches, g[f"f{t}_matches"]), t                    # assignment indices: bit exact
        assert np.array_equal(um_d, g[f"f{t}_um_d"]), t
        assert np.array_equal(um_t, g[f"f{t}_um_t"]), t
        assert np.array_equal(st["ids"], g[f"f{t}_ids"]), t                      # track ids: bit exact
        assert np.array_equal(st["state"], g[f"f{t}_state"]), t
        assert np.array_equal(st["tsu"], g[f"f{t}_tsu"]), t
        assert np.array_equal(st["hits"], g[f"f{t}_hits"]), t
        ref = g[f"f{t}_out"]
        assert out.shape == ref.shape == want.shape, t
        _check_int_rows(out, ref, st, stats)                                      # vs the reference trace
        _check_int_rows(out, want, st, [0, 0])                                    # vs the oracle
        if f"f{t}_mean" in g.files:
            np.testing.assert_allclose(st["mean"], g[f"f{t}_mean"], rtol=RTOL, atol=ATOL)
    assert stats[1] > 0 and stats[0] / stats[1] < 2e-3, stats                     # off-by-one box columns: < 0.2 % of all ints
    print("int32 box columns differing by one: %d of %d" % tuple(stats))
    return trk


def test_track_trace_30_golden():
    _run_trace(synth.PersonScene(30, seed=0, occlude_frac=0.15), golden("track_trace_30"), TRACE_PARAMS,
               drop=(20, 21), empty=(35,))


def test_track_trace_short_max_age_golden():
    _run_trace(synth.PersonScene(12, seed=2, occlude_frac=0.6), golden("track_trace_12_maxage4"),
               dict(TRACE_PARAMS, max_age=4))


def test_track_trace_crowd_200x150_golden():
    """BASELINE cfg5 association load: 200 live tracks, 150 detections per frame."""
    _run_trace(synth.PersonScene(200, seed=0, n_visible=150), golden("track_trace_200x150"), TRACE_PARAMS)


@pytest.mark.parametrize("name", ["track_trace_budget_none", "track_trace_euclidean_adapter"])
def test_tracker_metric_options_vs_reference_traces(name):
    """nn_budget=None (unbounded galleries: the per-track row capacity doubles past 32 and 64) and the euclidean metric."""
    from conftest import option_trace
    g, params, frame_of = option_trace(name)
    _run_trace(None, g, params, frame_of=frame_of)


def test_tracker_side_nms_trace_and_units():
    """DeepSort(nms_max_overlap=0.6).update: features for all boxes, preprocessing.non_max_suppression, survivors in pick
    order (deep_sort.py:46-57).  The constant-score argsort is numpy's choice in the reference; the trace pins the order
    the reference run observed."""
    from conftest import option_trace
    from yolo_deepsort_amd.deep_sort import DeepSort
    L = _lib()
    lib = L.load()
    g = golden("track_options_units")
    boxes = np.ascontiguousarray(g["nms_boxes"], dtype=F32)
    for k in range(4):
        order = np.ascontiguousarray(g[f"nms{k}_order"], dtype=np.int32)
        pick = np.zeros(len(boxes), np.int32)
        n = C.c_int(0)
        L.check(lib.yds_tracker_nms(L.ptr(boxes), L.ptr(order), len(boxes), float(g[f"nms{k}_thr"]), L.ptr(pick), C.byref(n)))
        assert pick[:n.value].tolist() == g[f"nms{k}_pick"].tolist(), k
    # euclidean helper per track segment
    seg = np.ascontiguousarray(g["euc_seg"], dtype=np.int32)
    gal, f = np.ascontiguousarray(g["euc_gallery"]), np.ascontiguousarray(g["euc_feats"])
    out = np.zeros((len(seg) - 1, len(f)), F32)
    L.check(lib.yds_euclidean_min_cost(L.ptr(gal), L.ptr(seg), len(seg) - 1, L.ptr(f), len(f), 512, L.ptr(out)))
    np.testing.assert_allclose(out, g["euc_out"], rtol=1e-5, atol=1e-4)
    assert out[1, 3] == 0.0
    # whole DeepSort.update with a callable extractor that returns the scripted features
    g, params, frame_of = option_trace("track_trace_nms06")
    state = {}

    def extractor(crops):
        assert len(crops) == len(state["feats"])
        return state["feats"]
    ds = DeepSort(extractor, use_cuda=True, **params)
    frame = np.zeros((1080, 1920, 3), np.uint8)
    for t in range(int(g["n_frames"])):
        ids, tlwh, feats = frame_of(t)
        state["feats"] = feats
        order = g[f"f{t}_nms_order"]
        ds._nms_keep = (lambda boxes, _o=order, _ds=ds: _keep_with_order(_ds, boxes, _o))
        out = ds.update(tlwh, np.ones(len(ids)), frame, (ids % 3 * 2).astype(F32))
        out = np.array(out, np.int32).reshape(-1, 6)
        ref = g[f"f{t}_out"]
        st = ds.tracker.state()
        assert np.array_equal(st["ids"], g[f"f{t}_ids"]) and np.array_equal(st["state"], g[f"f{t}_state"]), t
        assert out.shape == ref.shape and np.array_equal(out[:, 4:], ref[:, 4:]), t
        assert np.abs(out[:, :4] - ref[:, :4]).max(initial=0) <= 1, t
    with pytest.raises(ValueError):
        DeepSort(extractor, metric="manhattan")


def _keep_with_order(ds, tlwh, order):
    import ctypes as C
    from yolo_deepsort_amd import _lib as L
    d = tlwh.shape[0]
    order = np.ascontiguousarray(order, dtype=np.int32)
    pick = np.zeros(d, np.int32)
    n = C.c_int(0)
    L.check(L.load().yds_tracker_nms(L.ptr(tlwh), L.ptr(order), d, float(ds.nms_max_overlap), L.ptr(pick), C.byref(n)))
    return pick[:n.value].copy()


def test_unbounded_galleries_follow_live_tracks_not_frames_seen():
    """nn_budget=None: the per-track row capacity grows with the longest gallery a LIVE track holds - a stream whose
    tracks die young must not grow it with every frame (it used to double by frames seen: ~8.6 GB after 1k frames in pipeline
    mode), while a long-lived track still gets every row (reference nn_matching.py:152-156)."""
    from yolo_deepsort_amd.deep_sort import _TrackerHandle
    L = _lib()
    lib = L.load()
    trk = _TrackerHandle(0.3, 0.7, 3, 3, None)                    # max_age 3: a track that stops being detected dies within 4 frames
    rng = np.random.RandomState(5)
    feats = rng.randn(400, 512).astype(F32)
    cap0 = lib.yds_tracker_gallery_rows(trk._h)
    for t in range(400):
        gen = t // 20                                             # every 20 frames a new set of persons replaces the old one
        ids = np.arange(6) + 6 * gen
        tlwh = np.stack([100.0 + 150 * (ids % 6), 100.0 + 5 * (t % 20) + 0 * ids, 60.0 + 0 * ids, 120.0 + 0 * ids], 1).astype(F32)
        trk.step(tlwh, feats[ids % 400], np.zeros(6, F32))
    assert lib.yds_tracker_num_tracks(trk._h) <= 12
    assert cap0 <= lib.yds_tracker_gallery_rows(trk._h) <= 64      # galleries never exceeded ~20 rows; 400 frames seen
    keep = _TrackerHandle(0.3, 0.7, 30, 3, None)
    for t in range(150):                                          # one long-lived person: 150 rows must fit
        keep.step(np.array([[200.0 + t, 200.0, 60.0, 120.0]], F32), feats[:1], np.zeros(1, F32))
    assert lib.yds_tracker_gallery_rows(keep._h) >= 150 and keep.state()["hits"][0] == 150
