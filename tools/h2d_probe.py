"""Pinned host -> device copies: bandwidth and how long hipMemcpyAsync holds the calling thread (tuning aid, GPU box).
    python tools/h2d_probe.py [torch]      # 'torch': import torch first, i.e. run on torch's bundled HIP runtime"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch  # noqa: F401
import numpy as np
from yolo_deepsort_amd import _lib
_lib.init(); lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
ver = C.c_int(0); hip.hipRuntimeGetVersion(C.byref(ver))
n = 16 * 1080 * 1920 * 3
pin = _lib.PinnedArray((n,), np.uint8); pin.array[:] = 1
dev = _lib.DeviceBuffer(n)
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0          # hipStreamNonBlocking
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
for _ in range(3):
    hip.hipMemcpyAsync(dev.ptr, pin.ptr, n, 1, st); hip.hipStreamSynchronize(st)
call = tot = 0.0
for _ in range(20):
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(dev.ptr, pin.ptr, n, 1, st)
    t1 = time.perf_counter()
    hip.hipStreamSynchronize(st)
    t2 = time.perf_counter()
    call += t1 - t0; tot += t2 - t0
print(f"HIP runtime {ver.value}: pinned H2D {n / 1e6:.0f} MB async: call returns after {call / 20 * 1e3:.3f} ms, copy done after {tot / 20 * 1e3:.2f} ms = {n / (tot / 20) / 1e9:.1f} GB/s")
