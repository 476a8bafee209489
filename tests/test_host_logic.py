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
assert r.world == 2
scene = synth.PersonScene(5, seed=r.stream_seed())
ids, boxes = scene.boxes(0)
r.barrier()
dt = r.max_over_ranks(1.0 + r.rank)            # slowest rank defines the job time
tot = r.sum_over_ranks(float(boxes[:, 0].sum()))
frames = r.total_frames(10, 8)
assert r.gather_objects(("r", r.rank)) == [("r", 0), ("r", 1)]
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
