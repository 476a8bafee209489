#!/usr/bin/env python3
"""Only the VideoDetector leg of bench.py (tuning aid, run on the GPU box): python tools/vd_leg.py [cfg2] [frames] [host]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yolo_deepsort_amd import _lib  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
_lib.init()
fps, rec = bench.video_detector_leg(cfg, 32, 0, n, device_overlay=not (len(sys.argv) > 3 and sys.argv[3] == "host"))
print(json.dumps(dict(value=round(fps, 1), **rec)))
