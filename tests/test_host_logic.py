"""CPU tests of host-side logic: loaders, cfg generators, action module, multi-process plumbing."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, has_reference
from yolo_deepsort_amd import cfgs, loaders, synth


def test_parse_model_config_matches_oracle_restatement():
    from oracle.cfg import parse_model_config_text
    for name in cfgs.CFG_BUILDERS:
        text = cfgs.cfg_text(name)
        assert loaders.parse_model_config(None, text=text) == parse_model_config_text(text)
    d = loaders.parse_model_config(None, text="[net]\nchannels=3\n# c\n\n[convolutional]\n filters = 8 \nsize=1\n")
    assert d[1] == {"type": "convolutional", "batch_normalize": 0, "filters": "8", "size": "1"}


def test_load_classes_drops_last_element():
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
    names = loaders.load_classes(f.name)
    os.unlink(f.name)
    assert len(names) == 80 and names[0] == "person" and names[2] == "car" and names[-1] == "toothbrush"


def test_reid_checkpoint_roundtrip_zip_and_legacy():
    import torch
    sd = synth.reid_state_dict(0)
    assert len(sd) == 130
    for legacy in (False, True):
        with tempfile.NamedTemporaryFile(suffix=".t7", delete=False) as f:
            pass
        torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.5, "epoch": 1},
                   f.name, _use_new_zipfile_serialization=not legacy)
        back = loaders.load_reid_checkpoint(f.name)
        full = loaders.read_torch_checkpoint(f.name)
        # the reader needs no torch: same result in an interpreter where importing torch fails
        code = ("import sys; sys.modules['torch'] = None; sys.path.insert(0, %r)\n"
                "import numpy as np\nfrom yolo_deepsort_amd import loaders\n"
                "sd = loaders.load_reid_checkpoint(%r)\nprint(len(sd), float(sum(np.abs(v).sum() for v in sd.values())))\n" % (ROOT, f.name))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        os.unlink(f.name)
        assert out.returncode == 0, out.stderr[-1500:]
        n, tot = out.stdout.split()
        assert int(n) == 130 and abs(float(tot) - float(sum(np.abs(v).sum() for v in sd.values()))) < 1e-2
        assert full["acc"] == 0.5 and full["epoch"] == 1
        assert set(back) == set(sd)
        for k in sd:
            assert np.array_equal(back[k], sd[k]) and back[k].dtype == np.asarray(sd[k]).dtype and back[k].shape == np.asarray(sd[k]).shape, k


def test_checkpoint_reader_refuses_code():
    """A pickle that names anything but tensors/containers is rejected instead of executed."""
    import pickle
    import zipfile

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))
    with tempfile.NamedTemporaryFile(suffix=".t7", delete=False) as f:
        pass
    with zipfile.ZipFile(f.name, "w") as z:
        z.writestr("archive/data.pkl", pickle.dumps({"net_dict": Evil()}, protocol=2))
        z.writestr("archive/version", "3")
    try:
        with pytest.raises(pickle.UnpicklingError):
            loaders.read_torch_checkpoint(f.name)
    finally:
        os.unlink(f.name)


def test_weight_blob_layout():
    from oracle.darknet import DarknetOracle
    text = cfgs.cfg_text("yolov3-tiny")
    blob = synth.darknet_weights_blob(text, 0)
    assert len(blob) == 35434956            # canonical yolov3-tiny.weights size
    net = DarknetOracle(text, 416, is_text=True)
    assert net.load_weights_array(np.frombuffer(blob, dtype=np.float32, offset=20)) * 4 + 20 == len(blob)
    # head objectness bias lands on channels 4, 89, 174
    head = net.params[15]
    assert np.allclose(head["bias"][4::85], -4.0)


def test_scene_is_deterministic_and_bounded():
    a, b = synth.PersonScene(30, seed=0), synth.PersonScene(30, seed=0)
    for t in (0, 7, 133):
        ia, ba = a.boxes(t)
        ib, bb = b.boxes(t)
        assert np.array_equal(ia, ib) and np.array_equal(ba, bb)
        assert (ba[:, 0] >= 0).all() and (ba[:, 0] + ba[:, 2] <= 1920 + 1e-3).all()
        assert (ba[:, 1] >= 0).all() and (ba[:, 1] + ba[:, 3] <= 1080 + 1e-3).all()
        assert np.array_equal(a.features(t), b.features(t))
    f = a.frame(3)
    assert f.shape == (1080, 1920, 3) and f.dtype == np.uint8
    crowd = synth.PersonScene(200, seed=0, n_visible=150)
    assert len(crowd.boxes(5)[0]) == 150


def test_head_injection_decodes_back_to_boxes():
    """Injected logits must decode (oracle YOLO decode) to the scripted boxes."""
    from oracle.darknet import yolo_decode
    V3 = [(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]
    heads = [(19, 19, V3[0]), (38, 38, V3[1]), (76, 76, V3[2])]
    scene = synth.PersonScene(30, seed=0)
    _, tlwh = scene.boxes(0)
    rows = synth.head_injection(tlwh, (1080, 1920), (608, 608), heads)
    assert rows.shape == (30, 9)
    for r, (x, y, w, h) in zip(rows, tlwh):
        hi, a, gy, gx = (int(v) for v in r[:4])
        H, W, anchors = heads[hi]
        t = np.full((1, 3 * 85, H, W), -6.0, np.float32)
        t[0, a * 85:a * 85 + 4, gy, gx] = r[4:8]
        box = yolo_decode(t, anchors, 80, (608, 608))[0, a * H * W + gy * W + gx, :4]
        want = np.array([(x + w / 2) * 608 / 1920, (y + h / 2) * 608 / 1080, w * 608 / 1920, h * 608 / 1080])
        np.testing.assert_allclose(box, want, rtol=2e-3, atol=0.05)


@pytest.mark.ref
@pytest.mark.skipif(not has_reference(), reason="reference tree not present")
def test_action_module_matches_reference():
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import ref_harness
from yolo_deepsort_amd import action as mine
ref_harness.install_shims()
sys.path.insert(0, ref_harness.REF_ROOT)
for k in [k for k in sys.modules if k.split(".")[0] == "action"]:
    del sys.modules[k]
import action.action_Identify as rai, action.actions as ra
assert ra.__file__.startswith(ref_harness.REF_ROOT)
def mk(mod):
    return [mod.TakeOff(0, (1, 2)), mod.Landing(0, (1, 2)), mod.Glide(0, (2, 4)), mod.BreakInto(0, 2), mod.BreakInto(2, 1)]
A = rai.ActionIdentify(mk(ra), max_age=3, max_size=4)
B = mine.ActionIdentify(mk(mine), max_age=3, max_size=4)
rng = np.random.RandomState(0)
pos = rng.randint(0, 500, (6, 2)).astype(float)
vel = rng.randint(-6, 7, (6, 2)).astype(float)
for t in range(40):
    pos += vel
    if t %% 7 == 0: vel = rng.randint(-6, 7, (6, 2)).astype(float)
    vis = [i for i in range(6) if rng.rand() > 0.25]
    det = np.array([[pos[i, 0], pos[i, 1], pos[i, 0] + 20, pos[i, 1] + 40, i + 1, (i %% 2) * 2] for i in vis], np.int32).reshape(-1, 6)
    ra_out, rb_out = A.update(det), B.update(det)
    assert [(int(a), int(b), c) for a, b, c in ra_out] == [(int(a), int(b), c) for a, b, c in rb_out], (t, ra_out, rb_out)
assert A.update(None) is None and B.update(None) is None
print("OK")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_two_rank_gloo_stream_sharding():
    """N>1 path on CPU: rendezvous, barriers, max-over-ranks timing and per-rank stream seeds."""
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
from yolo_deepsort_amd.dist import Ranks
from yolo_deepsort_amd import synth
r = Ranks("gloo")
assert r.world == 2 and r.comm is None
scene = synth.PersonScene(5, seed=r.stream_seed())
ids, boxes = scene.boxes(0)
r.barrier()
dt = r.max_over_ranks(1.0 + r.rank)            # slowest rank defines the job time
tot = r.sum_over_ranks(float(boxes[:, 0].sum()))
frames = r.total_frames(10, 8)
assert r.gather_objects(("r", r.rank)) == [("r", 0), ("r", 1)]
import numpy as np
mine = [np.full((r.rank + 1, 6), 10 * r.rank + 1, np.int32), None if r.rank else np.zeros((0, 6), np.int32)]
streams = r.connect().gather_rows(mine)          # the exchange step: every rank ends up with both streams' rows
assert len(streams) == 2 and streams[0][0].shape == (1, 6) and streams[1][0].shape == (2, 6)
assert int(streams[1][0][0, 0]) == 11 and streams[0][1].shape == (0, 6) and streams[1][1] is None
assert r.rows == 64 and r.transport == "gloo"
# a frame with more rows than the block holds (only on rank 1): every rank grows its block alike and the exchange is repeated
crowd = [np.arange(300 * 6, dtype=np.int32).reshape(300, 6) if r.rank else np.zeros((2, 6), np.int32)]
streams = r.gather_rows(crowd)
assert r.rows == 320 and np.array_equal(streams[1][0], np.arange(300 * 6, dtype=np.int32).reshape(300, 6)) and streams[0][0].shape == (2, 6)
streams = r.gather_rows(mine)                     # the block stays grown, small frames still round-trip
assert r.rows == 320 and streams[1][0].shape == (2, 6) and streams[1][1] is None
if r.rank == 0:
    other = synth.PersonScene(5, seed=1).boxes(0)[1]
    assert dt == 2.0 and frames == 160
    assert abs(tot - float(boxes[:, 0].sum()) - float(other[:, 0].sum())) < 1e-3
    print("OK", frames / dt)
r.shutdown()
''' % ROOT
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    try:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", "29541", f.name],
                             capture_output=True, text=True, timeout=600, env=env)
    finally:
        os.unlink(f.name)
    assert out.returncode == 0 and "OK 80.0" in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]


def test_save_darknet_weights_layout_roundtrip():
    """save_darknet_weights re-emits the loaded stream (reference models.py:368-394); checked on the host-side
    float accounting with the oracle's reader (no GPU needed: the engine is not touched)."""
    from oracle.darknet import DarknetOracle
    from yolo_deepsort_amd.models import Darknet
    text = cfgs.cfg_text("yolov3-tiny")
    blob = synth.darknet_weights_blob(text, 3)
    net = Darknet.__new__(Darknet)                      # host-only skeleton: no device handle
    net.cfg_text, net._weights_blob, net.seen = text, blob, 7
    net.header_info = np.frombuffer(blob[:20], dtype=np.int32).copy()
    net.module_defs = loaders.parse_model_config(None, text=text)[1:]
    net._h = None
    with tempfile.NamedTemporaryFile(suffix=".weights", delete=False) as f:
        pass
    try:
        net.save_darknet_weights(f.name)
        back = open(f.name, "rb").read()
        assert back[20:] == blob[20:] and np.frombuffer(back[:20], np.int32)[3] == 7
        net.save_darknet_weights(f.name, cutoff=13)      # backbone only, like darknet53.conv.74 style cut-offs
        part = open(f.name, "rb").read()
        ref = DarknetOracle(text, 416, is_text=True)
        used = ref.load_weights_array(np.frombuffer(blob, dtype=np.float32, offset=20), cutoff=13)
        assert len(part) == 20 + used * 4 and part[20:] == blob[20:20 + used * 4]
    finally:
        os.unlink(f.name)


def test_resize_restatement_geometry_vs_torch_interpolate():
    """The GEOMETRY of oracle/resize.py (half-pixel centres, edge clamping, weights) cross-checked against an independent
    bilinear implementation that IS in the image: torch's interpolate (align_corners=False, fp64) on the same uint8 data.
    OpenCV's 8-bit path is fixed point (11-bit weights, truncating shifts, +2 >> 2), so it sits within one grey level of
    the real-valued result, biased slightly downwards: 12 % of the pixels differ from round(real value)."""
    import torch
    from oracle.resize import resize_bilinear_u8
    rng = np.random.RandomState(9)
    for (h, w), (dh, dw) in (((480, 640), (416, 416)), ((1080, 1920), (608, 608)), ((37, 91), (128, 64)), ((300, 200), (128, 64))):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = resize_bilinear_u8(img, (dw, dh)).astype(np.int32)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        diff = got - ref
        assert -0.9 < diff.min() and diff.max() < 0.6, ((h, w), float(diff.min()), float(diff.max()))
        assert np.abs(diff).mean() < 0.3
        assert 0.05 < (got != np.rint(ref)).mean() < 0.2          # the fp32-lerp + rint definition of round 1 was NOT cv2's
        assert got.shape == (dh, dw, 3)


def test_label_drawer_and_file_video_stream():
    """SURVEY 8f row 4 host stages: LabelDrawer (label_draw.py:118-191) and the FileVideoStream reader role."""
    from yolo_deepsort_amd.detect import FileVideoStream, _transform
    from yolo_deepsort_amd.label_draw import LabelDrawer
    names = ["person", "bicycle", "car"]
    d = LabelDrawer(names, None, 10, 2, (608, 608))
    state = np.random.get_state()[1][:4].copy()
    np.random.seed(1)                                         # the reference's colour table (label_draw.py:140-146)
    ref_colors = [tuple(int(v) for v in c) for c in (np.random.rand(3, 3) * 255).astype(int)]
    np.random.seed(None)
    assert d.colors == ref_colors
    img = np.zeros((240, 320, 3), np.uint8)
    rows = np.array([[20, 60, 120, 200, 7, 0], [150, 30, 300, 110, 12, 2]], np.int32)
    out, _, _ = d.draw_labels_by_trackers(img, rows, only_rect=False)
    assert out is img
    assert tuple(img[60, 70]) == d.colors[0] and tuple(img[200, 70]) == d.colors[0]          # top / bottom edge of box 0
    assert tuple(img[70, 150]) == d.colors[2] and tuple(img[130, 70]) == (0, 0, 0)           # left edge of box 1 / interior untouched
    assert (img[40:58, 20:60] == 0).all(axis=-1).any() and (img[40:58, 20:60] == d.colors[0]).all(axis=-1).any()   # plate with black text
    img2 = np.zeros((240, 320, 3), np.uint8)
    d.draw_labels_by_trackers(img2, rows, only_rect=True)
    assert (img2[40:58, 20:60] == 0).all()                                                    # no plate
    det = np.array([[10, 10, 50, 80, 0.9, 0.8, 1]], np.float32)
    d.draw_labels(np.zeros((240, 320, 3), np.uint8), det, only_rect=False)
    assert d.draw_labels(img, None, False)[0] is img
    # frame reader: .npy source, BGR -> RGB on the reader thread, bounded queue, end of stream
    frames = np.random.RandomState(0).randint(0, 256, (5, 8, 6, 3)).astype(np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".npy", delete=False) as f:
        np.save(f, frames)
    try:
        fvs = FileVideoStream(f.name, _transform, queue_size=2).start()
        got = []
        while fvs.more():
            fr = fvs.read()
            if fr is None:
                break
            got.append(fr)
        fvs.stop()
    finally:
        os.unlink(f.name)
    assert len(got) == 5 and all(np.array_equal(g, fr[:, :, ::-1]) for g, fr in zip(got, frames))
    with pytest.raises(IOError):
        FileVideoStream("/nonexistent/video.mp4")


def test_reference_import_paths_resolve_without_a_gpu():
    """Every import the reference demo and deep_sort/deep_sort.py:5-9 make resolves to this package (no device touched)."""
    import importlib
    want = {
        "yolo3.models": ["Darknet"], "yolo3.detect.video_detect": ["VideoDetector"], "yolo3.detect.img_detect": ["ImageDetector"],
        "yolo3.utils.model_build": ["soft_non_max_suppression", "xywh2p1p2", "p1p2Toxywh", "resize_boxes", "bbox_iou"],
        "yolo3.utils.parse_config": ["parse_model_config"], "yolo3.utils.helper": ["load_classes"], "yolo3.utils.label_draw": ["LabelDrawer"],
        "deep_sort": ["DeepSort", "build_tracker"], "deep_sort.deep_sort": ["DeepSort"], "deep_sort.deep.feature_extractor": ["Extractor"],
        "deep_sort.sort.detection": ["Detection"], "deep_sort.sort.kalman_filter": ["KalmanFilter", "chi2inv95"],
        "deep_sort.sort.nn_matching": ["NearestNeighborDistanceMetric"], "deep_sort.sort.preprocessing": ["non_max_suppression"],
        "deep_sort.sort.iou_matching": ["iou", "iou_cost"], "deep_sort.sort.track": ["Track", "TrackState"],
        "deep_sort.sort.linear_assignment": ["min_cost_matching", "matching_cascade", "gate_cost_matrix", "INFTY_COST"],
        "deep_sort.sort.tracker": ["Tracker"], "action.action_Identify": ["ActionIdentify"],
        "action.actions": ["TakeOff", "Landing", "Glide", "FastCrossing", "BreakInto"],
    }
    for mod, names in want.items():
        m = importlib.import_module(mod)
        assert m.__file__.startswith(ROOT), mod
        for n in names:
            assert hasattr(m, n), (mod, n)
    # host-only helpers behave like the reference's definitions
    from yolo3.utils.model_build import bbox_iou, resize_boxes, xywh2p1p2
    b = np.array([[10., 20., 4., 6.]], np.float32)
    assert np.array_equal(xywh2p1p2(b), np.array([[8., 17., 12., 23.]], np.float32))
    np.testing.assert_allclose(resize_boxes(np.array([[304., 304., 608., 608.]], np.float32), (608, 608), (1080, 1920)), [[960., 540., 1920., 1080.]])
    np.testing.assert_allclose(bbox_iou(np.array([[0., 0., 9., 9.]]), np.array([[5., 5., 14., 14.]])), [25. / 175.])


@pytest.mark.ref
@pytest.mark.skipif(not has_reference(), reason="reference tree not present")
def test_public_signatures_match_the_reference():
    """Constructor / method parameter names (and defaults) of the drop-in classes vs the reference's, read with inspect from the
    reference sources imported under the oracle's shims."""
    code = r'''
import inspect, sys
sys.path.insert(0, %r)
from oracle import ref_harness
ref_harness.install_shims()
import importlib
def sig(f):
    return [(p.name, p.default if p.default is not inspect._empty else "<req>") for p in inspect.signature(f).parameters.values()]
mine = {}
import yolo3.detect.video_detect as a, yolo3.detect.img_detect as b, deep_sort as c, yolo3.models as d, deep_sort.sort.nn_matching as e, deep_sort.sort.tracker as f
import deep_sort.sort.kalman_filter as k, deep_sort.sort.linear_assignment as la, yolo3.utils.model_build as mb, deep_sort.sort.preprocessing as pp
assert a.__file__.startswith(%r)
def collect():
    import yolo3.detect.video_detect as a, yolo3.detect.img_detect as b, deep_sort as c, yolo3.models as d, deep_sort.sort.nn_matching as e, deep_sort.sort.tracker as f
    import deep_sort.sort.kalman_filter as k, deep_sort.sort.linear_assignment as la, yolo3.utils.model_build as mb, deep_sort.sort.preprocessing as pp
    import deep_sort.sort.detection as dt
    return {
        "VideoDetector.__init__": sig(a.VideoDetector.__init__), "VideoDetector.detect": sig(a.VideoDetector.detect),
        "ImageDetector.__init__": sig(b.ImageDetector.__init__), "ImageDetector.detect": sig(b.ImageDetector.detect),
        "DeepSort.__init__": sig(c.DeepSort.__init__), "DeepSort.update": sig(c.DeepSort.update),
        "Darknet.__init__": sig(d.Darknet.__init__), "Darknet.load_darknet_weights": sig(d.Darknet.load_darknet_weights),
        "Darknet.save_darknet_weights": sig(d.Darknet.save_darknet_weights),
        "NNMetric.__init__": sig(e.NearestNeighborDistanceMetric.__init__), "NNMetric.distance": sig(e.NearestNeighborDistanceMetric.distance),
        "NNMetric.partial_fit": sig(e.NearestNeighborDistanceMetric.partial_fit),
        "Tracker.__init__": sig(f.Tracker.__init__), "Tracker.update": sig(f.Tracker.update), "Tracker.predict": sig(f.Tracker.predict),
        "KalmanFilter.gating_distance": sig(k.KalmanFilter.gating_distance), "KalmanFilter.update": sig(k.KalmanFilter.update),
        "min_cost_matching": sig(la.min_cost_matching), "gate_cost_matrix": sig(la.gate_cost_matrix),
        "soft_non_max_suppression": sig(mb.soft_non_max_suppression), "bbox_iou": sig(mb.bbox_iou), "resize_boxes": sig(mb.resize_boxes),
        "non_max_suppression": sig(pp.non_max_suppression), "Detection.__init__": sig(dt.Detection.__init__),
    }
mine = collect()
for key in [m for m in sys.modules if m.split(".")[0] in ("yolo3", "deep_sort", "action")]:
    del sys.modules[key]
sys.path.insert(0, ref_harness.REF_ROOT)
ref = collect()
import yolo3.models as chk
assert chk.__file__.startswith(ref_harness.REF_ROOT)
extra_ok = {"VideoDetector.__init__": {"batch_frames", "device_overlay"}, "DeepSort.__init__": {"metric"}, "Darknet.__init__": {"batch_max", "cfg_text"},
            "Darknet.load_darknet_weights": {"blob"}}
for name, r in ref.items():
    m = mine[name]
    if name == "VideoDetector.__init__":                      # (cv2.VideoWriter_fourcc('m','p','4','v') in the reference; the string here)
        m = [(n, "mp4v" if n == "fourcc" else d) for n, d in m]
        r = [(n, "mp4v" if n == "fourcc" else d) for n, d in r]
    assert m[:len(r)] == r, (name, m, r)                      # same names, order and defaults; additions only at the end
    assert {n for n, _ in m[len(r):]} <= extra_ok.get(name, set()), (name, m[len(r):])
print("OK", len(ref))
''' % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_checkpoint_reader_refuses_out_of_bounds_tensor_geometry():
    """A pickle whose tensor geometry points outside its storage must be refused (as_strided on unchecked numbers)."""
    import pickle
    import zipfile

    # write a checkpoint with torch, then corrupt the geometry through the reader's own objects
    import torch
    with tempfile.NamedTemporaryFile(suffix=".t7", delete=False) as f:
        pass
    try:
        torch.save({"net_dict": {"w": torch.arange(12, dtype=torch.float32).reshape(3, 4)}}, f.name)
        storages = {}
        with zipfile.ZipFile(f.name) as z:
            pkl = [n for n in z.namelist() if n.endswith("data.pkl")][0]
            with z.open(pkl) as fh:
                obj = loaders._make_unpickler(fh, storages).load()
            for key, st in storages.items():
                st.data = np.frombuffer(z.read(pkl[:-len("data.pkl")] + "data/" + key), dtype=st.dtype, count=st.numel)
        t = obj["net_dict"]["w"]
        assert np.array_equal(t.numpy(), np.arange(12, dtype=np.float32).reshape(3, 4))
        for bad in (dict(offset=1), dict(offset=-1), dict(size=(3, 5)), dict(stride=(8, 1)), dict(stride=(-4, 1)), dict(size=(3,))):
            u = loaders._LazyTensor(t.storage, bad.get("offset", t.offset), bad.get("size", t.size), bad.get("stride", t.stride))
            with pytest.raises(ValueError):
                u.numpy()
        assert loaders._LazyTensor(t.storage, 0, (0, 4), (4, 1)).numpy().shape == (0, 4)       # empty tensors address nothing
    finally:
        os.unlink(f.name)


def test_action_module_matches_committed_reference_fixture(monkeypatch):
    """action/  against the trace oracle/gen_golden.py recorded from the reference (40 frames of scripted
    tracker rows, stepped clock) - runs without /root/reference, i.e. also on the GPU box."""
    import json
    import time
    from conftest import GOLD
    from oracle.gen_golden import action_scene
    from yolo_deepsort_amd import action as mine
    want = json.load(open(os.path.join(GOLD, "action_trace.json")))
    tick = [1000.0]

    def fake_time():
        tick[0] += 0.04
        return tick[0]
    monkeypatch.setattr(time, "time", fake_time)
    acts = [mine.TakeOff(0, (1, 2)), mine.Landing(0, (1, 2)), mine.Glide(0, (2, 4)), mine.FastCrossing(2, 0.05), mine.BreakInto(0, 2),
            mine.BreakInto(2, 1)]
    B = mine.ActionIdentify(acts, max_age=3, max_size=4)
    for t, det in enumerate(action_scene()):
        got = [[int(a), int(b), str(c)] for a, b, c in B.update(det)]
        assert got == want["frames"][t], t
    assert (B.update(None) is None) == want["none_returns_none"]


def test_single_stream_mode_two_ranks_gloo():
    """SURVEY 8e optional mode on CPU (gloo, world 2): frames are 'detected' round-robin, exchanged as fixed-size blocks and
    'tracked' in frame order on rank 0 - with stand-in detect / track callables that make order and content checkable."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from yolo_deepsort_amd.dist import Ranks
from yolo_deepsort_amd import single_stream as ss
r = Ranks("gloo")
frames = list(range(7))                                       # 7 frames on 2 ranks: the last round is ragged
def detect(f):
    if f == 3:
        return None                                           # detector returned None: the tracker must not be called
    d = 70 if f == 5 else f %% 4                              # 0 detections at f = 4: the tracker IS called with D = 0; 70 at f = 5: the block (64) grows
    rng = np.random.RandomState(f)
    return rng.rand(d, 4).astype(np.float32), np.full(d, f, np.float32), rng.rand(d, 512).astype(np.float32)
seen = []
def track(tlwh, payload, feats):
    seen.append((tlwh.shape, payload.tolist(), float(feats.sum())))
    return ("rows", len(tlwh))
S = ss.SingleStream(r, detect, track)
out = S.run(frames)
if r.rank == 0:
    assert len(out) == 7 and out[3] is None and out[4] == ("rows", 0) and out[5] == ("rows", 70) and out[6] == ("rows", 2)
    want = [detect(f) for f in frames if f != 3]
    assert len(seen) == 6
    for (shape, pay, fs), w in zip(seen, want):               # in frame order, bit-identical content
        assert shape == w[0].shape and pay == w[1].tolist() and abs(fs - float(w[2].sum())) < 1e-3
    print("OK")
else:
    assert out == [] and seen == []
assert S.cap == 128                                             # grown alike on both ranks
# round 4: rounds of B frames per rank (rank r owns frames base + r * B .. + B): one exchange per N * B frames, same results in frame order
seen2 = []
def track2(tlwh, payload, feats):
    seen2.append((tlwh.shape, payload.tolist(), float(feats.sum())))
    return ("rows", len(tlwh))
S2 = ss.SingleStream(r, detect, track2, frames_per_rank=2)
out2 = S2.run(frames)
assert out2 == out and seen2 == seen and not S2.on_device
assert S2.exchanges == 3 and S.exchanges == 5                   # 7 frames: 2 rounds of 4 (+ 1 repeat for the growth) against 4 rounds of 2 (+ 1)
r.shutdown()
''' % ROOT
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    try:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                              "--master-port", "29543", f.name], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543"))
    finally:
        os.unlink(f.name)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]
    from yolo_deepsort_amd import single_stream as ss
    small = ss.pack_frame(np.zeros((151, 4), np.float32), np.zeros(151), np.zeros((151, 512)))    # does not fit 64: announces its size
    assert small[0] == -153 and ss.dets_needed(small[None]) == 151 and ss.cap_for(151) == 192
    with pytest.raises(ValueError):
        ss.unpack_frame(small)
    t, pl_, f = ss.unpack_frame(ss.pack_frame(np.ones((151, 4), np.float32), np.arange(151), np.ones((151, 512)), cap=192))
    assert t.shape == (151, 4) and pl_[150] == 150 and f.shape == (151, 512)


class _Clip:
    """cv2.VideoCapture protocol over an in-memory BGR clip."""

    def __init__(self, frames_bgr, fps):
        self.frames, self.fps, self.pos, self.sets = frames_bgr, fps, 0, []

    def isOpened(self):
        return True

    def get(self, prop):
        n, h, w = self.frames.shape[:3]
        return {5: self.fps, 6: 0.0, 3: float(w), 4: float(h), 7: float(n), 1: float(self.pos)}[prop]

    def set(self, prop, value):
        self.sets.append((prop, value))
        self.pos = int(value)

    def read(self):
        if self.pos >= len(self.frames):
            return False, None
        self.pos += 1
        return True, np.array(self.frames[self.pos - 1])


def test_seek_and_writer_rate_truncate_like_the_reference():
    """video_detect.py:92,97,101: video_fps = int(CAP_PROP_FPS); skip_frames = int(skip_secs) * video_fps; no seek past the end."""
    from yolo_deepsort_amd.detect import FileVideoStream
    clip = _Clip(np.zeros((400, 4, 4, 3), np.uint8), 29.97)
    fvs = FileVideoStream(clip)
    assert fvs.fps() == 29
    fvs.seek_secs(10.9)
    assert clip.sets == [(1, 290)]                            # not int(10.9 * 29.97) = 326
    clip2 = _Clip(np.zeros((5, 4, 4, 3), np.uint8), 25.0)
    FileVideoStream(clip2).seek_secs(6)                       # skip_secs > total_frames: "Can't skip over total video!", no seek
    assert clip2.sets == []
    clip3 = _Clip(np.zeros((5, 4, 4, 3), np.uint8), 25.0)
    FileVideoStream(clip3).seek_secs(0)
    assert clip3.sets == [(1, 0)]                             # the reference always issues the set (video_detect.py:101)


@pytest.mark.parametrize("batch_frames", [1, 16])
@pytest.mark.parametrize("case", ["tracker_skip2_mask", "tracker_every_frame", "no_tracker_skip3", "tracker_skip_secs"])
def test_video_generator_control_flow_vs_reference_fixture(case, batch_frames, tmp_path):
    """a9 on the CPU: the generator's control flow (skip gate, detector-None frames, hold, when the action module runs and what is
    yielded when it does not, the skip_secs seek) against what the reference's own VideoDetector.detect yielded
    (tests/golden/video_detect.npz), with stand-in stages that replay the fixture's per-frame results.  The device stages
    themselves are compared in tests/test_gpu_video_detect.py."""
    import json
    from conftest import golden
    from oracle.gen_golden import VIDEO_CASES, VIDEO_SCENE, CountingActions, video_clip
    from yolo_deepsort_amd import detect as D
    g = golden("video_detect")
    c, sc = VIDEO_CASES[case], VIDEO_SCENE
    frames, boxes, _ = video_clip(case)
    key = lambda f: hash(np.ascontiguousarray(f).tobytes())
    index = {key(f): t for t, f in enumerate(frames)}
    served = [int(t) for t in g[f"{case}_served"]]
    n = int(g[f"{case}_n"])
    hold_of = {served[i]: (None if bool(g[f"{case}_f{i}_none"]) else g[f"{case}_f{i}_hold"]) for i in range(n)}

    class Model:
        img_size, batch_max = (sc["img"], sc["img"]), 16

        def eval(self):
            return self

        def parameters(self):
            yield type("P", (), {"device": "cpu"})()

    class Tracker:
        nms_max_overlap = 1.0

        def update(self, boxs, conf, frame, cls):
            return [r for r in hold_of[index[key(frame)]]]

    names = tmp_path / "coco.names"
    names.write_text(cfgs.coco_names_text())
    act = CountingActions() if c["action"] else None
    if batch_frames > 1 and not c["tracker"]:
        pytest.skip("the batched pipeline needs a tracker (tracker=None keeps the frame-by-frame path)")
    vd = D.VideoDetector(Model(), str(names), thres=sc["thres"], nms_thres=sc["nms_thres"], skip_frames=c["skip_frames"],
                         class_mask=c["class_mask"], tracker=Tracker() if c["tracker"] else None, action_id=act, batch_frames=batch_frames)

    def detect(frame):                                        # stand-in detector: None exactly where the scripted frame is empty
        t = index[key(frame)]
        if len(boxes[t]) == 0:
            return None
        return np.zeros((1, 6), np.float32) if c["tracker"] else hold_of[t]
    vd.image_detector.detect = detect
    if batch_frames > 1:
        class Pipe:                                           # stand-in for pipeline.Pipeline.step: the fixture's rows per processed frame
            def __init__(self):
                self.groups = []

            def step(self, buf, h, w, nfr, ahead=None, select_next=None):
                return [None if hold_of[t] is None else np.asarray(hold_of[t], np.int32).reshape(-1, 6) for t in self.groups.pop(0)]
        pipe = Pipe()
        vd._pipe = pipe
        vd._batchable = lambda: True                              # the stand-in tracker plays the package's DeepSort
        orig = vd._processed_batches

        def spy(*a, **k):
            for grp in orig(*a, **k):
                bgr = k.get("transform", True) is False                 # a capture source is staged as the decoder's BGR (swapped on the device)
                pipe.groups.append([index[key(f[..., ::-1] if bgr else f)] for f, proc in grp if proc])
                yield grp
        vd._processed_batches = spy

        class Buf:
            def offset(self, o):
                return self

        # stand-ins for the two device hooks of the batched path: the staged group stays on the host, the output stage is the host form
        def upload(blk, group_frames, h, w, bgr):
            blk.update(dev=Buf(), host=[np.ascontiguousarray(f[..., ::-1]) if bgr else f for f in group_frames])
        vd._upload_group = upload
        vd._render_batch = lambda cur, holds, fps, bgr: [vd._render_host(cur["blk"]["host"][cur["slot_of"][i]], holds[i], None) for i in range(len(holds))]
    try:
        clip = _Clip(frames[..., ::-1], sc["fps"])
        got = list(vd.detect(clip, skip_secs=c["skip_secs"], show_fps=False))
    finally:
        pass
    if batch_frames > 1:
        assert not pipe.groups and vd.host_us["frames"] == n          # the batched generator really ran (every staged group consumed)
    assert len(got) == n
    assert clip.sets == [(1, served[0])]
    for i, (result, hold, actions) in enumerate(got):
        ref = hold_of[served[i]]
        assert (hold is None) == (ref is None), (case, i)
        if ref is not None:
            assert np.array_equal(np.asarray(hold).reshape(-1, 6), np.asarray(ref).reshape(-1, 6)), (case, i)
        assert json.loads(json.dumps(actions)) == json.loads(str(g[f"{case}_f{i}_actions"])), (case, i, actions, str(g[f"{case}_f{i}_actions"]))


def test_batched_generator_stops_cleanly_when_the_consumer_leaves(tmp_path):
    """The batched path of VideoDetector.detect runs a reader and an engine thread next to the generator (detect.py): a consumer that
    breaks out after a few frames (video_deepsort.py's `q` key, an exception in the loop body) must not leave them running or blocked,
    and an exception on either thread must surface in the consumer."""
    import threading
    import time
    from yolo_deepsort_amd import detect as D

    class Model:
        img_size, batch_max = (64, 64), 64

        def eval(self):
            return self

        def parameters(self):
            yield type("P", (), {"device": "cpu"})()

    class Tracker:
        nms_max_overlap = 1.0

    class Buf:
        def offset(self, o):
            return self

    class Pipe:
        def __init__(self, fail_at=None):
            self.n, self.fail_at = 0, fail_at

        def step(self, buf, h, w, n, ahead=None, select_next=None):
            self.n += 1
            if self.fail_at == self.n:
                raise RuntimeError("device fell over")
            time.sleep(0.005)
            return [np.zeros((1, 6), np.int32) for _ in range(n)]

    names = tmp_path / "coco.names"
    names.write_text(cfgs.coco_names_text())

    def make(pipe):
        vd = D.VideoDetector(Model(), str(names), tracker=Tracker(), batch_frames=4)
        vd._pipe = pipe
        vd._batchable = lambda: True                              # the stand-in tracker plays the package's DeepSort
        vd._upload_group = lambda blk, fr, h, w, bgr: blk.update(dev=Buf(), host=list(fr))
        vd._render_batch = lambda cur, holds, fps, bgr: [vd._render_host(cur["blk"]["host"][cur["slot_of"][i]], holds[i], None) for i in range(len(holds))]
        return vd
    before = threading.active_count()
    gen = make(Pipe()).detect((np.zeros((48, 64, 3), np.uint8) for _ in range(10000)), show_fps=False)
    for i, _ in enumerate(gen):
        if i == 5:
            break
    gen.close()
    deadline = time.time() + 3
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == before
    # a failing step: raised where the consumer iterates
    with pytest.raises(RuntimeError, match="device fell over"):
        for _ in make(Pipe(fail_at=3)).detect((np.zeros((48, 64, 3), np.uint8) for _ in range(100)), show_fps=False):
            pass
    # a failing source (frames of two shapes): raised in the consumer too
    def bad():
        for i in range(20):
            yield np.zeros((48, 64 if i < 6 else 65, 3), np.uint8)
    with pytest.raises(ValueError, match="share one"):
        for _ in make(Pipe()).detect(bad(), show_fps=False):
            pass
    deadline = time.time() + 3
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == before


def test_batched_path_is_gated_on_the_package_tracker_and_live_captures():
    """the default (batch_frames=None) must keep the frame-by-frame loop for a tracker that only offers update() - a
    DeepSort around a user callable, a custom tracker - and an already-open capture without a frame count is a live source."""
    from types import SimpleNamespace
    from yolo_deepsort_amd.deep_sort import DeepSort, Extractor
    from yolo_deepsort_amd.detect import VideoDetector

    class Cap:                                             # what cv2.VideoCapture(0) looks like to _is_live
        def __init__(self, n):
            self.n = n

        def isOpened(self):
            return True

        def get(self, prop):
            assert prop == 7                               # cv2.CAP_PROP_FRAME_COUNT
            return self.n

    assert VideoDetector._is_live(Cap(0)) and VideoDetector._is_live(Cap(-1)) and not VideoDetector._is_live(Cap(250))
    assert VideoDetector._is_live(0) and VideoDetector._is_live("rtsp://cam/1") and not VideoDetector._is_live("clip.mp4")

    def vd(tracker):
        v = object.__new__(VideoDetector)
        v.tracker, v.image_detector = tracker, SimpleNamespace(win_size=None)
        return v

    assert not vd(None)._batchable()
    assert not vd(SimpleNamespace(update=lambda *a: []))._batchable()                      # custom tracker: update() only
    ds = object.__new__(DeepSort)                                                          # DeepSort around a user callable
    ds.extractor, ds.tracker, ds.nms_max_overlap = (lambda crops: np.zeros((len(crops), 512), np.float32)), SimpleNamespace(_h=1), 1.0
    assert not vd(ds)._batchable()
    ex = object.__new__(Extractor)
    ex._h = None                                                                           # (keeps __del__ quiet)
    ds.extractor = ex
    assert vd(ds)._batchable()
    ds.nms_max_overlap = 0.5                                                               # tracker-side NMS reorders on the host
    assert not vd(ds)._batchable()
    ds.nms_max_overlap = 1.0
    v = vd(ds)
    v.image_detector.win_size = (416, 416)                                                 # tiled detection: frame by frame
    assert not v._batchable()


def test_auto_batch_follows_the_clip_length_and_sclk_parsing(tmp_path):
    """Round 6: the default read-ahead never exceeds what the source holds (a 6-frame clip must not size every buffer for 68 frames),
    and the roofline record's clock reader parses the driver's pp_dpm_sclk lines."""
    from yolo_deepsort_amd.detect import VideoDetector
    from yolo_deepsort_amd import pipeline as pl

    class Cap:
        def __init__(self, n):
            self.n = n

        def isOpened(self):
            return True

        def get(self, prop):
            return float(self.n)

    assert VideoDetector._source_len(Cap(6)) == 6 and VideoDetector._source_len(Cap(0)) is None
    assert VideoDetector._source_len([1, 2, 3]) == 3 and VideoDetector._source_len(iter(())) is None
    np.save(tmp_path / "clip.npy", np.zeros((5, 4, 4, 3), np.uint8))
    assert VideoDetector._source_len(str(tmp_path / "clip.npy")) == 5 and VideoDetector._source_len("movie.mp4") is None
    s = object.__new__(pl.SclkSampler)
    s.path = str(tmp_path / "pp_dpm_sclk")
    (tmp_path / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 1950Mhz *\n2: 2400Mhz\n")
    assert s._read() == 1950.0
    (tmp_path / "pp_dpm_sclk").write_text("S: 109Mhz *\n0: 500Mhz\n1: 2400Mhz\n")
    assert s._read() == 109.0
    s.samples, s.period = [109.0, 1950.0, 2050.0], 0.01
    assert s.ghz() == 2.0                                   # the sleep state between legs is not a sample of the loaded clock
    s.path = str(tmp_path / "absent")
    assert s._read() is None
