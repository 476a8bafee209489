"""oracle/fast.py (C + OpenMP form of the oracle used as bench.py's CPU baseline) against the plain numpy oracle."""
import numpy as np

from yolo_deepsort_amd import cfgs, synth

F32 = np.float32


def _pair(cfg, size, seed=1, obj_bias=-1.0):
    from oracle.darknet import DarknetOracle
    from oracle.fast import DarknetFast
    ref = DarknetOracle(cfg, size, is_text=True)
    ref.load_weights_array(np.frombuffer(synth.darknet_weights_blob(cfg, seed, obj_bias), dtype=F32, offset=20))
    return ref, DarknetFast(ref)


def test_fast_detector_equals_numpy_oracle():
    from oracle.gen_golden import MINI_CFG
    rng = np.random.RandomState(0)
    for cfg, size in ((MINI_CFG, (32, 32)), (cfgs.cfg_text("yolov3-tiny", 96, 96), (96, 96)), (cfgs.cfg_text("yolov4-tiny", 96, 64), (96, 64)),
                      (cfgs.cfg_text("yolov4", 64, 64), (64, 64))):
        ref, fast = _pair(cfg, size)
        x = rng.uniform(0, 1, (2, 3) + size).astype(F32)
        a, b = ref(x), fast(x)
        assert a.shape == b.shape
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-4)


def test_fast_reid_and_pipeline_equal_numpy_oracle():
    from oracle import reid as oreid
    from oracle.fast import ReidFast, DarknetFast
    from oracle.pipeline import run_stream
    sd = synth.reid_state_dict(0)
    x = np.random.RandomState(1).randn(5, 3, 128, 64).astype(F32)
    np.testing.assert_allclose(ReidFast(sd)(x), oreid.reid_forward(x, sd), rtol=1e-4, atol=1e-5)
    # whole loop on a small stream: identical rows
    cfg = cfgs.cfg_text("yolov3-tiny", 160, 160)
    ref, fast = _pair(cfg, (160, 160), seed=0, obj_bias=-4.0)
    scene = synth.PersonScene(5, frame_hw=(240, 320), seed=3)
    heads = [(5, 5, [(81, 82), (135, 169), (344, 319)]), (10, 10, [(10, 14), (23, 27), (37, 58)])]
    frames = [scene.frame(t) for t in range(4)]
    inj = [synth.head_injection(scene.boxes(t)[1], (240, 320), (160, 160), heads) for t in range(4)]
    ds = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)
    a = run_stream(ref, sd, ds, frames, inj)
    b = run_stream(fast, sd, ds, frames, inj, reid_fn=ReidFast(sd))
    assert len(a) == len(b) == 4
    for u, v in zip(a, b):
        assert np.array_equal(np.array(u, np.int32).reshape(-1, 6), np.array(v, np.int32).reshape(-1, 6))
    assert len(a[3]) > 0
