#!/usr/bin/env python3
"""Phase accounting of the 512-thread window kernel (YDS_TIMING_WIN=1 experiment build), default arithmetic and cross8:
tools/phase_prof_win.py shape batch"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from yolo_deepsort_amd import _lib
shape = [int(v) for v in sys.argv[1].split(",")]
batch = int(sys.argv[2])
os.environ["YDS_CONV_FORCE"] = "15"
_lib.init(0)
lib = _lib.load()
out = np.zeros(8, np.uint64)
h, w, cin, cout, k, s, act, res = shape
for x8 in (0, 1):
    lib.yds_set_conv_cross8(x8)
    us, var = C.c_double(), C.c_int()
    _lib.check(lib.yds_conv_bench(batch, h, w, cin, cout, k, s, act, res, 3, C.byref(us), C.byref(var)))
    _lib.check(lib.yds_debug_prof(_lib.ptr(out), 5))
    _lib.check(lib.yds_conv_bench(batch, h, w, cin, cout, k, s, act, res, 20, C.byref(us), C.byref(var)))
    _lib.check(lib.yds_debug_prof(_lib.ptr(out), 5))
    pro, loop, epi, n = (float(v) for v in out[:4])
    steps = 9 * cin // 32
    print(f"cross8={x8} {us.value:.1f} us  per workgroup (wave 0, {n:.0f} samples): prologue {pro / n:.0f}  K loop {loop / n:.0f} ({loop / n / steps:.0f} per step, {steps} steps)  epilogue {epi / n:.0f}  total {(pro + loop + epi) / n:.0f} cycles")
