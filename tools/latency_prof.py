"""Frame-by-frame leg of bench.py on its own (batch_frames = 1; profiling aid, run on the GPU box):
    python tools/latency_prof.py [--config cfg2] [--frames 60] [--lookahead]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import _lib
from yolo_deepsort_amd.workload import Workload

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg2")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--lookahead", action="store_true")
a = ap.parse_args()
_lib.init()
lib = _lib.load()
wl = Workload(a.config, a.batch, seed=0, n_distinct=64)
wl.to_device()
for i in range(20):
    wl.step(i, prefetch=a.lookahead and i < 19)
_lib.check(lib.yds_device_sync())
t0 = time.perf_counter()
for i in range(20, 20 + a.frames):
    wl.step(i, prefetch=a.lookahead and i + 1 < 20 + a.frames)
_lib.check(lib.yds_device_sync())
dt = time.perf_counter() - t0
print(f"batch {a.batch} lookahead={a.lookahead}: {dt / a.frames * 1e3:.3f} ms per step, {a.frames * a.batch / dt:.1f} frames/s; stages {wl.pipe.stage_us()}")
