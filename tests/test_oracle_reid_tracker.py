"""Oracle ReID / Kalman / association restatement vs golden vectors from the real reference."""
import numpy as np
import pytest

from conftest import golden
from oracle import reid as oreid
from oracle import tracker as otrk
from oracle import clib
from yolo_deepsort_amd import synth

F32 = np.float32
RTOL, ATOL = 1e-3, 1e-3


def test_reid_crops_and_features():
    g = golden("reid_seed0")
    scene = synth.PersonScene(8, seed=4)
    frame = scene.frame(0)
    tlwh = g["tlwh"]
    crops = oreid.crop_boxes(tlwh, *frame.shape[:2])
    assert np.array_equal(crops, g["crops"])
    pre = oreid.preprocess_crops(frame, tlwh)
    assert pre.shape == (8, 3, 128, 64)
    np.testing.assert_allclose(pre[:, :, ::16, ::8], g["pre_sample"], rtol=0, atol=1e-6)
    feats = oreid.reid_forward(pre, synth.reid_state_dict(0))
    np.testing.assert_allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(feats, g["feats"], rtol=RTOL, atol=1e-5)


def test_kalman_known_answer_and_batch():
    g = golden("kalman")
    m0, c0 = otrk.kf_initiate(np.array([10, 15, .5, 10], F32))
    assert np.array_equal(m0, g["ka_m0"]) and np.array_equal(c0, g["ka_c0"])
    m1, c1 = otrk.kf_predict(m0, c0)
    assert np.array_equal(m1, g["ka_m1"]) and np.array_equal(c1, g["ka_c1"])
    m2, c2 = otrk.kf_update(m1, c1, np.array([[12, 20, .6, 11]], F32))
    np.testing.assert_allclose(m2, g["ka_m2"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c2, g["ka_c2"], rtol=1e-4, atol=1e-6)
    # SURVEY appendix D
    np.testing.assert_allclose(m2[0], [11.73553753, 19.33884239, 0.50196081, 10.86776829,
                                       0.41322312, 1.03305781, 0, 0.20661156], rtol=1e-5, atol=1e-6)
    means, covs = zip(*[otrk.kf_initiate(x) for x in g["xyah"]])
    mean, cov = np.concatenate(means), np.concatenate(covs)
    assert np.array_equal(mean, g["init_mean"]) and np.array_equal(cov, g["init_cov"])
    for s in range(3):
        mean, cov = otrk.kf_predict(mean, cov)
        np.testing.assert_allclose(mean, g[f"pred{s}_mean"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(cov, g[f"pred{s}_cov"], rtol=1e-4, atol=1e-5)
        mean, cov = otrk.kf_update(mean, cov, g[f"z{s}"])
        np.testing.assert_allclose(mean, g[f"upd{s}_mean"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(cov, g[f"upd{s}_cov"], rtol=1e-3, atol=1e-4)
    g2 = otrk.kf_gating_distance(mean, cov, g["meas"], True)
    g4 = otrk.kf_gating_distance(mean, cov, g["meas"], False)
    np.testing.assert_allclose(g2, g["gate2"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(g4, g["gate4"], rtol=1e-3, atol=1e-2)


def test_lsap_matches_scipy_tie_heavy():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.RandomState(0)
    for trial in range(600):
        nr, nc = rng.randint(1, 16, 2)
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
        r1, c1 = clib.lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)
    # SURVEY appendix C observations
    assert clib.lsap(np.ones((4, 4)))[1].tolist() == [0, 1, 2, 3]
    assert clib.lsap(np.ones((5, 3)))[1].tolist() == [0, 1, 2]
    c = np.full((4, 4), 0.30001)
    c[0, 2], c[3, 2], c[1, 0] = 0.1, 0.05, 0.2
    assert clib.lsap(c)[1].tolist() == [3, 0, 1, 2]
    for shape in ((200, 150), (150, 200)):
        c = np.full(shape, 0.30001, F32)
        m = rng.rand(*shape) < 0.02
        c[m] = rng.rand(m.sum()) * 0.3
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = clib.lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)


def _run_trace(scene, g, params, drop=(), empty=(), frame_of=None):
    trk = otrk.TrackerOracle(**params)
    n = int(g["n_frames"])
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
        if f"f{t}_nms_order" in g.files:
            trk.nms_order = g[f"f{t}_nms_order"]          # the constant-vector argsort the reference run observed
        out = trk.update(tlwh, feats, (ids % 3 * 2).astype(F32))
        out = np.array(out, dtype=np.int32).reshape(-1, 6)
        st = trk.state()
        assert np.array_equal(np.array(trk.debug["matches"], np.int32).reshape(-1, 2), g[f"f{t}_matches"]), t
        assert np.array_equal(np.array(trk.debug["unmatched_detections"], np.int32), g[f"f{t}_um_d"]), t
        assert np.array_equal(np.array(trk.debug["unmatched_tracks"], np.int32), g[f"f{t}_um_t"]), t
        assert np.array_equal(st["ids"], g[f"f{t}_ids"]), t
        assert np.array_equal(st["state"], g[f"f{t}_state"]), t
        assert np.array_equal(st["tsu"], g[f"f{t}_tsu"]), t
        assert np.array_equal(st["hits"], g[f"f{t}_hits"]), t
        ref = g[f"f{t}_out"]
        assert out.shape == ref.shape, t
        assert np.array_equal(out[:, 4:], ref[:, 4:]), t                   # ids + classes bit-exact
        assert np.abs(out[:, :4] - ref[:, :4]).max(initial=0) <= 1, t      # int truncation of fp32 boxes
        if f"f{t}_mean" in g.files:
            np.testing.assert_allclose(st["mean"], g[f"f{t}_mean"], rtol=RTOL, atol=ATOL)
    return trk


TRACE_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


def test_track_trace_30():
    trk = _run_trace(synth.PersonScene(30, seed=0, occlude_frac=0.15), golden("track_trace_30"),
                     TRACE_PARAMS, drop=(20, 21), empty=(35,))
    assert max(t.track_id for t in trk.tracks) >= 30


def test_track_trace_short_max_age():
    _run_trace(synth.PersonScene(12, seed=2, occlude_frac=0.6), golden("track_trace_12_maxage4"),
               dict(TRACE_PARAMS, max_age=4))


def test_track_trace_crowd_200x150():
    _run_trace(synth.PersonScene(200, seed=0, n_visible=150), golden("track_trace_200x150"), TRACE_PARAMS)


@pytest.mark.parametrize("name", ["track_trace_budget_none", "track_trace_nms06", "track_trace_euclidean_adapter"])
def test_track_option_traces(name):
    """budget=None, tracker-side NMS and the euclidean metric against reference-generated traces."""
    from conftest import option_trace
    g, params, frame_of = option_trace(name)
    trk = _run_trace(None, g, params, frame_of=frame_of)
    if name == "track_trace_budget_none":
        assert max(len(v) for v in trk.samples.values()) > 30          # galleries really grew past the demo's budget
    if name == "track_trace_nms06":
        assert any(len(g[f"f{t}_matches"]) + len(g[f"f{t}_um_d"]) < 2 * len(frame_of(t)[0]) for t in range(int(g["n_frames"])))


def test_option_units_vs_reference_vectors():
    g = golden("track_options_units")
    seg = g["euc_seg"]
    got = np.stack([otrk.euclidean_min_distance(g["euc_gallery"][seg[i]:seg[i + 1]], g["euc_feats"]) for i in range(len(seg) - 1)], 0)
    np.testing.assert_allclose(got, g["euc_out"], rtol=1e-5, atol=1e-4)
    assert got[1, 3] == 0.0                                              # the planted exact match (clamp at 0)
    for k in range(4):
        assert otrk.tracker_nms(g["nms_boxes"], float(g[f"nms{k}_thr"]), g[f"nms{k}_order"]) == g[f"nms{k}_pick"].tolist(), k
    with pytest.raises(ValueError):
        otrk.TrackerOracle(metric="manhattan")


def test_reid_on_wide_range_weights():
    """The ReID restatement on the wide-dynamic-range weights (synth profile "wide") incl. an all-black and an all-white crop."""
    from oracle.gen_golden import WIDE_SEED, wide_reid_frame
    g = golden("wide_range")
    frame, tlwh = wide_reid_frame()
    assert np.array_equal(tlwh, g["reid_tlwh"])
    sd = synth.reid_state_dict(WIDE_SEED, "wide")
    var = np.concatenate([v.reshape(-1) for k, v in sd.items() if k.endswith("running_var")])
    assert var.min() < 2e-3 and var.max() > 50
    feats = oreid.reid_forward(oreid.preprocess_crops(frame, tlwh), sd)
    np.testing.assert_allclose(feats, g["reid_feats"], rtol=RTOL, atol=1e-5)
    d = 1.0 - g["reid_feats"] @ g["reid_feats"].T
    assert d[~np.eye(8, dtype=bool)].min() > 1e-3                      # the embeddings still tell the crops apart
