import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference (build container only)")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="session")
def gold():
    return golden


def has_reference():
    return os.path.isdir("/root/reference/yolo3")


TRACE_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


def option_trace(name):
    """Inputs of the option traces written by oracle/gen_golden.py gen_track_options:
    (golden file, tracker kwargs, frame -> (ids, tlwh, feats))."""
    from yolo_deepsort_amd import synth
    g = golden(name)
    if name == "track_trace_budget_none":
        scene = synth.PersonScene(10, seed=5, occlude_frac=0.3)
        return g, dict(TRACE_PARAMS, nn_budget=None), lambda t: (*scene.boxes(t), scene.features(t))
    if name == "track_trace_euclidean_adapter":
        scene = synth.PersonScene(20, seed=7, occlude_frac=0.2)
        return g, dict(TRACE_PARAMS, max_dist=0.6, metric="euclidean"), lambda t: (*scene.boxes(t), scene.features(t))
    if name == "track_trace_nms06":
        scene = synth.PersonScene(14, seed=6, occlude_frac=0.1)
        jit = g["jitter"]

        def frame(t):
            ids, tlwh = scene.boxes(t)
            f = scene.features(t)
            dup = (tlwh + jit[t, :len(ids)]).astype(np.float32)
            return np.concatenate([ids, ids]), np.concatenate([tlwh, dup], 0), np.concatenate([f, f], 0)
        return g, dict(TRACE_PARAMS, nms_max_overlap=0.6), frame
    raise KeyError(name)
