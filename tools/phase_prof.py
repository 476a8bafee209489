#!/usr/bin/env python3
"""Phase accounting of the LDS-DMA conv kernel (YDS_TIMING=1 experiment build): tools/phase_prof.py shape batch variant"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from yolo_deepsort_amd import _lib
shape = [int(v) for v in sys.argv[1].split(",")]
batch, variant = int(sys.argv[2]), sys.argv[3]
os.environ["YDS_CONV_FORCE"] = variant
_lib.init(0)
lib = _lib.load()
out = np.zeros(8, np.uint64)
us, var = C.c_double(), C.c_int()
h, w, cin, cout, k, s, act, res = shape
_lib.check(lib.yds_conv_bench(batch, h, w, cin, cout, k, s, act, res, 3, C.byref(us), C.byref(var)))
_lib.check(lib.yds_debug_prof(_lib.ptr(out), 1))
_lib.check(lib.yds_conv_bench(batch, h, w, cin, cout, k, s, act, res, 20, C.byref(us), C.byref(var)))
_lib.check(lib.yds_debug_prof(_lib.ptr(out), 1))
wait, bar, body, total, steps, waves = (float(v) for v in out[:6])
print(f"{lib.yds_conv_variant_name(var.value).decode()}  {us.value:.1f} us")
print(f"per wave-step ticks (100 MHz s_memtime? see below): wait_vm {wait / steps:.1f}  barrier {bar / steps:.1f}  body {body / steps:.1f}")
print(f"per wave: loop total {total / waves:.0f} ticks over {steps / waves:.1f} steps")
