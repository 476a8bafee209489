#!/usr/bin/env python3
"""Timing of the LSAP kernel through yds_lsap (host call = upload + kernel + download + sync): the 2x2 call gives the
fixed overhead, the rest is kernel time.  Matrices: thresholded appearance costs as the tracker produces them."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import _lib  # noqa: E402

_lib.init()
lib = _lib.load()


def run(c, reps=20):
    c = np.ascontiguousarray(c, np.float32)
    nr, nc = c.shape
    us = C.c_double(0)
    _lib.check(lib.yds_lsap_bench(_lib.ptr(c), nr, nc, reps, C.byref(us)))
    return us.value


rng = np.random.RandomState(0)
base = 0.0
print(f"2x2: {run(rng.rand(2, 2)):.1f} us (launch floor)")
for shape, frac in (((30, 30), 0.05), ((40, 30), 0.05), ((200, 150), 0.02), ((200, 150), 0.006), ((150, 200), 0.02), ((500, 400), 0.01), ((1500, 1200), 0.002)):
    c = np.full(shape, 0.30001, np.float32)
    m = rng.rand(*shape) < frac
    c[m] = rng.rand(m.sum()) * 0.3
    # one good candidate per detection like a real scene
    for j in range(min(shape)):
        c[j % shape[0], j] = rng.rand() * 0.1
    print(f"{shape} thresholded ({frac}): {run(c) - base:.1f} us kernel")
    print(f"{shape} dense random        : {run(rng.rand(*shape)) - base:.1f} us kernel")
