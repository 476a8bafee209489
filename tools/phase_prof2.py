#!/usr/bin/env python3
"""Phase accounting of the two-workgroup window kernel (YDS_TIMING2=1 experiment build): tools/phase_prof2.py shape batch"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from yolo_deepsort_amd import _lib
shape = [int(v) for v in sys.argv[1].split(",")]
batch = int(sys.argv[2])
os.environ["YDS_CONV_FORCE"] = "20"
_lib.init(0)
lib = _lib.load()
out = np.zeros(8, np.uint64)
us, var = C.c_double(), C.c_int()
h, w, cin, cout, k, s, act, res = shape
_lib.check(lib.yds_conv_bench(batch, h, w, cin, cout, k, s, act, res, 3, C.byref(us), C.byref(var)))
_lib.check(lib.yds_debug_prof(_lib.ptr(out), 3))
_lib.check(lib.yds_conv_bench(batch, h, w, cin, cout, k, s, act, res, 20, C.byref(us), C.byref(var)))
_lib.check(lib.yds_debug_prof(_lib.ptr(out), 3))
wait, bar, body, pro, epi, total, steps, waves = (float(v) for v in out)
print(f"{lib.yds_conv_variant_name(var.value).decode()}  {us.value:.1f} us   ({waves:.0f} waves, {steps / waves:.1f} steps each)")
print(f"per wave-step cycles: wait(vm+lgkm) {wait / steps:.0f}  barrier {bar / steps:.0f}  body {body / steps:.0f}   (12 MFMAs = 384 cycles of matrix pipe)")
if os.environ.get("YDS_PROF_WALL"):
    print(f"effective shader clock inside the workgroups: {total / wait / 10:.1f} MHz x100 -> {total / wait * 0.1:.3f} GHz  (s_memtime cycles / 100 MHz wall ticks)")
print(f"per wave: prologue {pro / waves:.0f}  loop {(wait + bar + body) / waves:.0f}  epilogue {epi / waves:.0f}  total {total / waves:.0f} cycles")
