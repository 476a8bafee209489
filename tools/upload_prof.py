"""The upload-inclusive leg of bench.py on its own, with the pipeline's stage times (tuning aid, GPU box):
    python tools/upload_prof.py [--hbm]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import _lib
from yolo_deepsort_amd.workload import Workload
ap = argparse.ArgumentParser()
ap.add_argument("--hbm", action="store_true")
ap.add_argument("--config", default="cfg2")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--schedule", type=int, default=None, help="yds_pipeline_set_schedule: 0 serialized, -1 two streams")
a = ap.parse_args()
_lib.init(); lib = _lib.load()
wl = Workload(a.config, a.batch, seed=0)
wl.pipe.set_schedule(a.schedule)
wl.to_device()
host = not a.hbm
for i in range(5):
    wl.step(i, prefetch=i < 4, host_frames=host, prefetch2=i < 3)
_lib.check(lib.yds_device_sync())
t0 = time.perf_counter(); acc = {}
N = 20
for i in range(5, 5 + N):
    wl.step(i, prefetch=i + 1 < 5 + N, host_frames=host, prefetch2=i + 2 < 5 + N)
    for k, v in wl.pipe.stage_us().items():
        acc[k] = acc.get(k, 0.0) + v / N
_lib.check(lib.yds_device_sync())
dt = (time.perf_counter() - t0) / N
print(("host frames" if host else "hbm frames"), f"{dt * 1e3:.3f} ms per step, {a.batch / dt:.1f} frames/s ({wl.pipe.last_schedule()}); mean stages", {k: round(v) for k, v in acc.items()})
