"""Side-stream clock probe (csrc/clock_probe.hip) beside back-to-back launches of the dominant window layer; with
YDS_BUILD_TAG=inkernel (a -DYDS_CLOCK_PROBE build) the same call reads the rounds 3-5 in-kernel sampling instead."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import _lib, pipeline as pl       # noqa: E402

_lib.init()
lib = _lib.load()
us, var = C.c_double(), C.c_int()
for label, iters in (("idle", 0), ("window layer x200", 200), ("window layer x200", 200)):
    pl.conv_clock(reset=True)
    t0 = time.perf_counter()
    if iters:
        _lib.check(lib.yds_conv_bench(32, 76, 76, 128, 256, 3, 1, 1, 0, iters, C.byref(us), C.byref(var)))
    else:
        time.sleep(0.2)
    ghz, ms = pl.conv_clock(reset=False)
    print(f"{label}: probe {ghz:.3f} GHz over {ms:.1f} ms (host {1e3 * (time.perf_counter() - t0):.1f} ms), launch {us.value:.1f} us", flush=True)
