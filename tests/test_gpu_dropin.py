"""The drop-in boundary through the REFERENCE'S OWN import paths: `tests/dropin_caller.py` - a caller written
for this repo from the signature table, not from any reference script - builds Darknet / DeepSort / ActionIdentify /
VideoDetector through `yolo3.*`, `deep_sort`, `action.*` in a scratch directory holding config/yolov4.cfg,
weights/yolov4.weights, weights/ckpt.t7, config/coco.names (synthetic, real file formats) and an .npy video source."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden

pytestmark = pytest.mark.gpu

CALLER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin_caller.py")


def _scratch(tmp_path):
    import torch
    from yolo_deepsort_amd import cfgs, synth
    (tmp_path / "config").mkdir()
    (tmp_path / "weights").mkdir()
    cfg = cfgs.cfg_text("yolov4", 608, 608)
    (tmp_path / "config" / "yolov4.cfg").write_text(cfg)
    (tmp_path / "config" / "coco.names").write_text(cfgs.coco_names_text())
    (tmp_path / "weights" / "yolov4.weights").write_bytes(synth.darknet_weights_blob(cfg, seed=0, obj_bias=1.0))
    sd = synth.reid_state_dict(0)
    torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.9, "epoch": 40},
               str(tmp_path / "weights" / "ckpt.t7"))
    scene = synth.PersonScene(6, frame_hw=(270, 480), seed=3, occlude_frac=0.0)
    frames = np.stack([scene.frame(t)[:, :, ::-1] for t in range(6)], 0)         # BGR on disk, like a decoded video
    np.save(tmp_path / "frames.npy", frames)
    return cfg, sd, frames


def _run_demo(tmp_path, half):
    import json
    out = subprocess.run([sys.executable, CALLER, ROOT, half], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("ROWS ")][-1]
    return json.loads(line[5:])


def test_caller_through_reference_import_paths(tmp_path):
    """The shim-path caller with half=False, checked against the package's own frame-by-frame API on the same files."""
    from yolo_deepsort_amd import _lib, loaders
    from yolo_deepsort_amd.deep_sort import DeepSort
    from yolo_deepsort_amd.detect import ImageDetector, p1p2Toxywh
    from yolo_deepsort_amd.models import Darknet
    cfg, sd, frames = _scratch(tmp_path)
    rows = _run_demo(tmp_path, "0")
    assert len(rows) == 6
    assert os.path.exists(tmp_path / "out.npy") and np.load(tmp_path / "out.npy").shape == (6, 270, 480, 3)
    # the same path through the package's own names, frame by frame (skip_frames=2: frames 0, 2, 4 are processed, the rest hold)
    _lib.init()
    net = Darknet(str(tmp_path / "config" / "yolov4.cfg"), img_size=(608, 608))
    net.load_darknet_weights(str(tmp_path / "weights" / "yolov4.weights"))
    ds = DeepSort(str(tmp_path / "weights" / "ckpt.t7"), min_confidence=1, use_cuda=True, nn_budget=30, n_init=3, max_iou_distance=0.7,
                  max_dist=0.3, max_age=30)
    det = ImageDetector(net, str(tmp_path / "config" / "coco.names"), thres=0.5, nms_thres=0.4)
    want, hold, n_det = [], None, 0
    for t in range(6):
        if t % 2 == 0:
            rgb = np.ascontiguousarray(frames[t][:, :, ::-1])
            d = det.detect(rgb)
            if d is None:
                hold = None
            else:
                d = d.numpy() if hasattr(d, "numpy") else d
                m = np.isin(d[:, -1], [0, 2, 4])
                n_det += int(m.sum())
                hold = ds.update(p1p2Toxywh(d[:, :4])[m].astype(np.float32), d[m, 4], rgb, d[m, -1])
        want.append(None if hold is None else np.asarray(hold, np.int32).reshape(-1, 6).tolist())
    assert rows == want
    assert n_det > 0, "the synthetic head bias should let some candidates through (otherwise the loop only saw None)"


def test_caller_half_mode(tmp_path):
    """half=True (the mode the reference demo constructs its VideoDetector in): runs, yields one result per frame."""
    _scratch(tmp_path)
    rows = _run_demo(tmp_path, "1")
    assert len(rows) == 6


def test_model_build_names_vs_goldens():
    """yolo3.utils.model_build.{soft_non_max_suppression, xywh2p1p2, resize_boxes, bbox_iou, p1p2Toxywh} at the reference's
    import path, each against the reference-generated NMS goldens."""
    import torch
    from yolo3.utils.model_build import bbox_iou, p1p2Toxywh, resize_boxes, soft_non_max_suppression, xywh2p1p2
    g = golden("nms_cases")
    names = sorted({k[:-5] for k in g.files if k.endswith("_pred")})
    checked = 0
    for nme in names:
        pred, (conf, iou), want = g[nme + "_pred"], (float(v) for v in g[nme + "_thr"]), g[nme + "_out"]
        got = soft_non_max_suppression(torch.from_numpy(pred.copy()), conf, iou)[0]
        if want.shape[0] == 0:
            assert got is None
            continue
        assert isinstance(got, torch.Tensor) and tuple(got.shape) == want.shape, nme
        assert np.array_equal(got.numpy(), want), nme
        checked += 1
        # `classes` = keep those class columns only (the reference filters the candidates before the greedy step)
        cls = sorted(set(want[:, 5].astype(int)))[:1]
        sub = soft_non_max_suppression(pred.copy(), conf, iou, classes=cls)[0]
        assert sub is not None and set(sub[:, 5].astype(int)) == set(cls), nme
        if want.shape[0] < 300:                       # (below the 300 cap the other classes cannot displace any row)
            assert np.array_equal(sub, want[want[:, 5] == cls[0]]), nme
    assert checked >= 5
    # box helpers: element-wise definitions (model_build.py:12-19, 317-332, 354-381)
    b = np.array([[10., 20., 4., 6.], [0., 0., 2., 2.]], np.float32)
    assert np.array_equal(xywh2p1p2(b), np.array([[8., 17., 12., 23.], [-1., -1., 1., 1.]], np.float32))
    assert np.array_equal(xywh2p1p2(torch.from_numpy(b)).numpy(), xywh2p1p2(b))
    assert np.array_equal(p1p2Toxywh(np.array([[1., 2., 5., 9.]])), np.array([[1., 2., 4., 7.]]))
    r = resize_boxes(np.array([[304., 304., 608., 608.]], np.float32), (608, 608), (1080, 1920))
    np.testing.assert_allclose(r, [[960., 540., 1920., 1080.]], rtol=1e-6)
    i = bbox_iou(np.array([[0., 0., 9., 9.]]), np.array([[0., 0., 9., 9.], [5., 5., 14., 14.], [20., 20., 30., 30.]]))
    np.testing.assert_allclose(i, [1.0, 25. / 175., 0.0], rtol=1e-6)
    m = golden("tiled_detect")
    for nme in ("all_kept", "one_kept", "plain", "single"):
        got = soft_non_max_suppression(m[nme + "_pred"].copy(), 0.5, 0.4, merge=True, is_p1p2=True)[0]
        want = m[nme + "_out"]
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), nme
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-6, atol=1e-4, err_msg=nme)


def test_deep_sort_sort_modules_at_reference_paths():
    """deep_sort.sort.{detection, kalman_filter, nn_matching, preprocessing, iou_matching, linear_assignment, tracker}
    (reference deep_sort/deep_sort.py:5-9 imports them) against the reference-generated vectors and traces."""
    from deep_sort.deep.feature_extractor import Extractor                      # noqa: F401
    from deep_sort.sort import iou_matching, linear_assignment
    from deep_sort.sort.detection import Detection
    from deep_sort.sort.kalman_filter import KalmanFilter, chi2inv95
    from deep_sort.sort.nn_matching import NearestNeighborDistanceMetric
    from deep_sort.sort.preprocessing import non_max_suppression
    from deep_sort.sort.track import TrackState
    from deep_sort.sort.tracker import Tracker
    from yolo_deepsort_amd import synth
    from conftest import TRACE_PARAMS
    g = golden("kalman")
    kf = KalmanFilter()
    mean, cov = kf.initiate(g["xyah"])
    np.testing.assert_allclose(mean, g["init_mean"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cov, g["init_cov"], rtol=1e-5, atol=1e-12)
    for s in range(3):
        mean, cov = kf.predict(mean, cov)
        np.testing.assert_allclose(mean, g[f"pred{s}_mean"], rtol=1e-5, atol=1e-5)
        mean, cov = kf.update(mean, cov, g[f"z{s}"])
        np.testing.assert_allclose(mean, g[f"upd{s}_mean"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(cov, g[f"upd{s}_cov"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(kf.gating_distance(mean, cov, g["meas"], only_position=True), g["gate2"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(kf.gating_distance(mean, cov, g["meas"], only_position=False), g["gate4"], rtol=1e-3, atol=1e-3)
    m4, c4 = kf.project(mean, cov)
    assert np.array_equal(m4, mean[:, :4]) and np.allclose(c4[:, 2, 2], cov[:, 2, 2] + np.float32(1e-2)) and chi2inv95[2] == 5.9915
    # nn metric (euclidean unit vectors from the reference's own _nn_euclidean_distance) + partial_fit budget
    u = golden("track_options_units")
    met = NearestNeighborDistanceMetric("euclidean", 0.5, budget=None)
    seg = u["euc_seg"]
    feats, targets = [], []
    for t in range(len(seg) - 1):
        feats += list(u["euc_gallery"][seg[t]:seg[t + 1]])
        targets += [t] * int(seg[t + 1] - seg[t])
    met.partial_fit(feats, targets, list(range(len(seg) - 1)))
    np.testing.assert_allclose(met.distance(u["euc_feats"], list(range(len(seg) - 1))), u["euc_out"], rtol=1e-4, atol=1e-4)
    cm = NearestNeighborDistanceMetric("cosine", 0.3, budget=2)
    cm.partial_fit(u["euc_gallery"][:5], [7, 7, 7, 8, 8], [7, 8])
    assert len(cm.samples[7]) == 2 and len(cm.samples[8]) == 2
    gal = np.concatenate([u["euc_gallery"][1:3], u["euc_gallery"][3:5]], 0).astype(np.float64)
    gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    f = u["euc_feats"].astype(np.float64)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    d = 1.0 - gal @ f.T
    np.testing.assert_allclose(cm.distance(u["euc_feats"], [7, 8]), np.stack([d[:2].min(0), d[2:].min(0)], 0), rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        NearestNeighborDistanceMetric("manhattan", 0.3)
    # preprocessing.non_max_suppression with the recorded score orders
    for k in range(4):
        order = u[f"nms{k}_order"]
        scores = np.empty(len(order))
        scores[order] = np.arange(len(order))                       # argsort(scores) == the recorded order
        assert non_max_suppression(u["nms_boxes"], float(u[f"nms{k}_thr"]), scores) == u[f"nms{k}_pick"].tolist()
    # linear assignment on tie-heavy thresholded costs vs scipy, and the list bookkeeping of min_cost_matching
    from scipy.optimize import linear_sum_assignment
    rng = np.random.RandomState(0)
    for trial in range(30):
        nr, nc = int(rng.randint(1, 40)), int(rng.randint(1, 40))
        cost = rng.uniform(0, 0.6, (nr, nc)).astype(np.float32)
        cost[cost > 0.3] = np.float32(0.3 + 1e-5)
        r, c = linear_assignment.linear_assignment(cost)
        r2, c2 = linear_sum_assignment(cost)
        assert np.array_equal(r, r2) and np.array_equal(c, c2)
    cost = np.array([[0.1, 0.9, 0.9], [0.9, 0.9, 0.2]], np.float32)
    m, ut, ud = linear_assignment.min_cost_matching(lambda *a: cost.copy(), 0.5, [None, None], [None] * 3)
    assert m == [(0, 0), (1, 2)] and ut == [] and ud == [1]
    # iou: asymmetric +1 (identical boxes give IoU > 1)
    b = np.array([10., 10., 20., 40.], np.float32)
    i = iou_matching.iou(b, np.stack([b, b + [100, 0, 0, 0]], 0))
    assert i[0] > 1.0 and i[1] == 0.0
    # Tracker(metric).predict()/update(detections) on a reference trace: ids, states, output boxes
    gt = golden("track_trace_30")
    scene = synth.PersonScene(30, seed=0, occlude_frac=0.15)
    trk = Tracker(NearestNeighborDistanceMetric("cosine", TRACE_PARAMS["max_dist"], TRACE_PARAMS["nn_budget"]),
                  max_iou_distance=TRACE_PARAMS["max_iou_distance"], max_age=TRACE_PARAMS["max_age"], n_init=TRACE_PARAMS["n_init"])
    with pytest.raises(RuntimeError):
        trk.update([])
    for t in range(12):
        ids, tlwh = scene.boxes(t)
        feats = scene.features(t)
        dets = [Detection(tlwh[k], 1, feats[k], payload=float(ids[k] % 3 * 2)) for k in range(len(ids))]
        trk.predict()
        trk.update(dets)
        tracks = trk.tracks
        assert [x.track_id for x in tracks] == gt[f"f{t}_ids"].tolist()
        assert [x.state for x in tracks] == gt[f"f{t}_state"].tolist()
        assert [x.time_since_update for x in tracks] == gt[f"f{t}_tsu"].tolist()
        shown = [x for x in tracks if x.is_confirmed() and x.time_since_update <= 1]
        ref = gt[f"f{t}_out"]
        assert [x.track_id for x in shown] == ref[:, 4].tolist() and [int(x.payload) for x in shown] == ref[:, 5].tolist()
        if shown:
            tlbr = np.stack([x.to_tlbr() for x in shown], 0)
            assert np.abs(np.maximum(tlbr, [0, 0, -1e9, -1e9]).astype(np.int32) - ref[:, :4]).max() <= 1
    assert TrackState.Confirmed == 2 and any(x.is_confirmed() for x in tracks)
