"""End-to-end hot path (csrc/pipeline.cpp through the C ABI) vs the oracle pipeline on the same frames."""
import numpy as np
import pytest

from yolo_deepsort_amd import cfgs, synth

pytestmark = pytest.mark.gpu
F32 = np.float32
DS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


@pytest.mark.parametrize("name,size,batch,deep,frames_in", [("yolov3-tiny", 416, 4, False, "hbm"), ("yolov4-tiny", 416, 3, False, "hbm"),
                                                            ("yolov3-tiny", 416, 4, True, "hbm"), ("yolov3-tiny", 416, 4, False, "pinned"),
                                                            ("yolov3-tiny", 416, 4, True, "pageable"), ("yolov3-tiny", 416, 4, False, "pinned3")])
def test_pipeline_matches_oracle_stream(name, size, batch, deep, frames_in, monkeypatch):
    # frames_in: resident in HBM (yds_pipeline_step) or handed over as host memory and uploaded by the pipeline on its
    # copy stream, double buffered (yds_pipeline_step_host; pinned = asynchronous copies, pageable = any numpy array)
    # deep: the crowded-scene schedule (next batch's NMS + ReID started before this batch's association) forced on
    monkeypatch.setenv("YDS_PIPE_DEEP_MIN", "0" if deep else "1000000")
    from oracle.darknet import DarknetOracle
    from oracle.pipeline import run_stream
    from yolo_deepsort_amd import _lib, pipeline as pl
    from yolo_deepsort_amd.deep_sort import DeepSort
    from yolo_deepsort_amd.models import Darknet
    _lib.init(0)
    cfg = cfgs.cfg_text(name, size, size)
    blob = synth.darknet_weights_blob(cfg, 0)
    net = Darknet(None, img_size=(size, size), batch_max=batch, cfg_text=cfg)
    net.load_darknet_weights(None, blob=blob)
    sd = synth.reid_state_dict(0)
    ds = DeepSort(sd, use_cuda=True, **DS)
    scene = synth.PersonScene(10, frame_hw=(480, 640), seed=3, occlude_frac=0.2)
    n = 4 * batch
    frames = np.stack([scene.frame(t) for t in range(n)], 0)
    heads = net.yolo_heads()
    # every third person is a "car" (class 2), one a class the mask drops (class 5)
    inj = []
    for t in range(n):
        ids, tlwh = scene.boxes(t)
        rows = synth.head_injection(tlwh, (480, 640), (size, size), heads)
        rows[:, 8] = np.where(ids % 3 == 0, 2, 0)
        rows[ids == 4, 8] = 5
        inj.append(rows)
    inj[5] = inj[5][:0]                         # a frame where the detector returns None
    pl.load_injection_sets(net, [[inj[s * batch + b] for b in range(batch)] for s in range(n // batch)])
    pipe = pl.Pipeline(net, ds, 0.5, 0.4, class_mask=[0, 2, 4])
    got = []
    steps = n // batch
    pl.select_injection_set(net, 0)
    if frames_in == "hbm":
        dev = _lib.DeviceBuffer.from_array(frames)
        for s in range(steps):
            # all but the last step hand the next batch over early (detector prefetch overlapping the association)
            nxt = dev.offset((s + 1) * batch * frames[0].nbytes) if s + 1 < steps else None
            got += pipe.step(dev.offset(s * batch * frames[0].nbytes), 480, 640, batch, nxt,
                             select_next=(s + 1 if nxt is not None else None))
    elif frames_in == "pinned3":
        # a decoder three batches deep: the batch after next is announced (yds_pipeline_prefetch_host) one step before it is `next`
        hold = [_lib.PinnedArray((batch, 480, 640, 3)) for _ in range(3)]
        bufs = [h.array for h in hold]
        bufs[0][:] = frames[:batch]
        bufs[1][:] = frames[batch:2 * batch]
        for s in range(steps):
            if s + 2 < steps:
                bufs[(s + 2) % 3][:] = frames[(s + 2) * batch:(s + 3) * batch]
                pipe.prefetch_host(bufs[(s + 2) % 3])
            nxt = bufs[(s + 1) % 3] if s + 1 < steps else None
            got += pipe.step_host(bufs[s % 3], nxt, select_next=(s + 1 if nxt is not None else None))
    else:
        # a decoder's two-buffer ring: the batch after next overwrites the buffer of the batch that just finished
        if frames_in == "pinned":
            hold = [_lib.PinnedArray((batch, 480, 640, 3)), _lib.PinnedArray((batch, 480, 640, 3))]
            bufs = [h.array for h in hold]
        else:
            bufs = [np.empty((batch, 480, 640, 3), np.uint8), np.empty((batch, 480, 640, 3), np.uint8)]
        bufs[0][:] = frames[:batch]
        for s in range(steps):
            nxt = None
            if s + 1 < steps:
                nxt = bufs[(s + 1) % 2]
                nxt[:] = frames[(s + 1) * batch:(s + 2) * batch]
            got += pipe.step_host(bufs[s % 2], nxt, select_next=(s + 1 if nxt is not None else None))
    ref_net = DarknetOracle(cfg, size, is_text=True)
    ref_net.load_weights_array(np.frombuffer(blob, dtype=F32, offset=20))
    want = run_stream(ref_net, sd, DS, frames, inj)
    assert len(got) == len(want) == n
    seen_rows = 0
    for t, (g, w) in enumerate(zip(got, want)):
        if w is None:
            assert g is None, t
            continue
        w = np.array(w, np.int32).reshape(-1, 6)
        assert g.shape == w.shape, (t, g, w)
        assert np.array_equal(g[:, 4:], w[:, 4:]), t              # track ids and classes: bit exact
        assert np.abs(g[:, :4] - w[:, :4]).max(initial=0) <= 1, t
        seen_rows += len(w)
    assert got[5] is None and seen_rows > 5 * 8
    st = pipe.stage_us()
    assert st["detector_dev"] > 0


def test_video_detector_batched_lookahead_equals_frame_by_frame():
    """VideoDetector(batch_frames=N) reads N frames ahead and runs the batched pipeline; per-frame results must be the
    ones of the reference-style frame-by-frame loop (ids / classes bit exact, boxes within a pixel)."""
    import os
    import tempfile
    from yolo_deepsort_amd import _lib
    from yolo_deepsort_amd.deep_sort import DeepSort
    from yolo_deepsort_amd.detect import VideoDetector
    from yolo_deepsort_amd.models import Darknet
    _lib.init(0)
    cfg = cfgs.cfg_text("yolov3-tiny", 416, 416)
    blob = synth.darknet_weights_blob(cfg, 0, -1.45)          # a handful of (random) detections per frame
    sd = synth.reid_state_dict(0)
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
    rng = np.random.RandomState(21)
    base = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    frames = [np.roll(base, 3 * t, axis=1) for t in range(11)]
    runs = {}
    for bf in (1, 4):
        net = Darknet(None, img_size=(416, 416), batch_max=max(bf, 1), cfg_text=cfg)
        net.load_darknet_weights(None, blob=blob)
        vd = VideoDetector(net, f.name, thres=0.5, nms_thres=0.4, skip_frames=2, tracker=DeepSort(sd, use_cuda=True, **DS),
                           batch_frames=bf)
        out_npy = f.name + ".npy" if bf == 4 else None
        runs[bf] = [(img.shape, None if d is None else np.array(d, np.int32).reshape(-1, 6)) for img, d, _ in vd.detect(frames, output_path=out_npy)]
        if out_npy:
            saved = np.load(out_npy)
            os.unlink(out_npy)
            assert saved.shape == (len(frames), 480, 640, 3)
            assert (saved[-1] != frames[-1][:, :, ::-1]).any()          # overlay (boxes / FPS text) drawn on the BGR output
    os.unlink(f.name)
    assert len(runs[1]) == len(runs[4]) == len(frames)
    rows = 0
    for t, ((s1, d1), (s4, d4)) in enumerate(zip(runs[1], runs[4])):
        assert s1 == s4 == (480, 640, 3)
        assert (d1 is None) == (d4 is None), t
        if d1 is None:
            continue
        assert d1.shape == d4.shape, t
        assert np.array_equal(d1[:, 4:], d4[:, 4:]), t
        assert np.abs(d1[:, :4] - d4[:, :4]).max(initial=0) <= 1, t
        rows += len(d1)
    assert rows > 0


def test_frame_by_frame_device_crop_path_equals_host_feature_path():
    """VideoDetector.process hands the detector's device copy of the frame to the extractor and the features stay in HBM
    (yds_reid_embed_dev -> yds_tracker_step_sel); the extractor's pass must be complete before the tracker's stream reads
    it.  Same stream through the host-feature path: identical rows, frame after frame, with person
    textures that move so that stale features would change the costs."""
    import os
    import tempfile
    from yolo_deepsort_amd import _lib
    from yolo_deepsort_amd.deep_sort import DeepSort
    from yolo_deepsort_amd.detect import VideoDetector, p1p2Toxywh
    from yolo_deepsort_amd.models import Darknet
    _lib.init(0)
    cfg = cfgs.cfg_text("yolov3-tiny", 416, 416)
    blob = synth.darknet_weights_blob(cfg, 0)
    sd = synth.reid_state_dict(0)
    with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
        f.write(cfgs.coco_names_text())
    net = Darknet(None, img_size=(416, 416), cfg_text=cfg)
    net.load_darknet_weights(None, blob=blob)
    scene = synth.PersonScene(12, frame_hw=(480, 640), seed=8, occlude_frac=0.1)
    vd = VideoDetector(net, f.name, thres=0.5, nms_thres=0.4, tracker=DeepSort(sd, use_cuda=True, **DS), class_mask=[0])
    os.unlink(f.name)
    ds2 = DeepSort(vd.tracker.extractor, use_cuda=True, **DS)            # shares the extractor, own tracker (clone semantics)
    heads = net.yolo_heads()
    rows_seen = 0
    for t in range(12):
        frame = scene.frame(t)
        net.set_injection(0, synth.head_injection(scene.boxes(t)[1], (480, 640), (416, 416), heads))
        got = vd.process(frame)
        det = vd.image_detector.detect(frame)
        det = det.numpy() if hasattr(det, "numpy") else det
        det = det[det[:, 5] == 0]
        tlwh = p1p2Toxywh(det[:, :4]).astype(F32)
        feats = ds2.extractor.embed(frame, tlwh, to_host=True)
        want = ds2.tracker.step(tlwh, feats, det[:, 5])
        got = np.array(got, np.int32).reshape(-1, 6)
        assert np.array_equal(got, want), t
        rows_seen += len(want)
    assert rows_seen > 50


def test_pipeline_nms_workspace_overflow_redoes_the_batch():
    """A threshold so low that a frame has more (box, class) candidates than the pipeline's NMS workspace (4096): the
    workspace grows and the batch is redone instead of failing (the reference has no limit, model_build.py:93-121)."""
    from oracle.darknet import DarknetOracle
    from oracle.pipeline import run_stream
    from yolo_deepsort_amd import _lib, pipeline as pl
    from yolo_deepsort_amd.deep_sort import DeepSort
    from yolo_deepsort_amd.models import Darknet
    _lib.init(0)
    cfg = cfgs.cfg_text("yolov3-tiny", 416, 416)
    blob = synth.darknet_weights_blob(cfg, 0, -1.0)
    net = Darknet(None, img_size=(416, 416), batch_max=2, cfg_text=cfg)
    net.load_darknet_weights(None, blob=blob)
    sd = synth.reid_state_dict(0)
    ds = DeepSort(sd, use_cuda=True, **DS)
    frames = np.random.RandomState(4).randint(0, 256, (4, 480, 640, 3)).astype(np.uint8)
    conf = 0.3
    ref_net = DarknetOracle(cfg, 416, is_text=True)
    ref_net.load_weights_array(np.frombuffer(blob, dtype=F32, offset=20))
    from oracle.resize import resize_bilinear_u8
    x = resize_bilinear_u8(frames[0], (416, 416)).astype(F32).transpose(2, 0, 1)[None] / F32(255.)
    pred = ref_net(x)[0]
    n_cand = int(((pred[:, 5:] * pred[:, 4:5] > conf) & (pred[:, 4:5] > conf)).sum())
    assert n_cand > 4096, n_cand                                    # the case this test is about
    pipe = pl.Pipeline(net, ds, conf, 0.4, class_mask=[0, 2, 4])
    dev = _lib.DeviceBuffer.from_array(frames)
    got = pipe.step(dev.offset(0), 480, 640, 2, dev.offset(2 * frames[0].nbytes)) + pipe.step(dev.offset(2 * frames[0].nbytes), 480, 640, 2)
    want = run_stream(ref_net, sd, DS, frames, None, conf=conf)
    for t, (g, w) in enumerate(zip(got, want)):
        w = np.array(w, np.int32).reshape(-1, 6)
        assert g is not None and g.shape == w.shape, t
        assert np.array_equal(g[:, 4:], w[:, 4:]), t
        assert np.abs(g[:, :4] - w[:, :4]).max(initial=0) <= 1, t
