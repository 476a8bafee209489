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


def check_int_rows(out, ref, st, stats):
    """int32 output rows (deep_sort.py:85-87 truncates fp32 boxes) against the reference's: ids / classes exact; a box column
    may differ by one ONLY where the pre-truncation float (from the tracker state `st` right after that frame) sits within
    2e-3 of an integer - the consequence of fp32 values that agree with the reference's to 1e-3."""
    assert out.shape == ref.shape
    assert np.array_equal(out[:, 4:], ref[:, 4:])
    shown = (st["state"] == 2) & (st["tsu"] <= 1)
    m = st["mean"][shown].astype(np.float64)
    assert m.shape[0] == out.shape[0]
    w, h = m[:, 2] * m[:, 3], m[:, 3]
    x1, y1 = m[:, 0] - w / 2, m[:, 1] - h / 2
    fl = np.stack([np.maximum(x1, 0), np.maximum(y1, 0), x1 + w, y1 + h], 1)
    bad = out[:, :4] != ref[:, :4]
    assert np.abs(out[:, :4] - ref[:, :4]).max(initial=0) <= 1
    assert (np.abs(fl[bad] - np.rint(fl[bad])) < 2e-3).all(), fl[bad]
    stats[0] += int(bad.sum())
    stats[1] += bad.size
