"""Detector-only loop (resize + all layers + decode, frames resident in HBM): tools/det_loop.py  (run on the GPU box)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from yolo_deepsort_amd import _lib, cfgs, synth
from yolo_deepsort_amd.models import Darknet
_lib.init(0); lib=_lib.load()
B=int(os.environ.get("DET_LOOP_B", "16"))
cfg=cfgs.cfg_text(os.environ.get("DET_NET", "yolov3"),608,608)
net=Darknet(None,img_size=(608,608),batch_max=B,cfg_text=cfg); net.load_darknet_weights(None,blob=synth.darknet_weights_blob(cfg,0))
frames=np.random.RandomState(0).randint(0,256,(B,1080,1920,3)).astype(np.uint8)
dev=_lib.DeviceBuffer.from_array(frames)
for i in range(3): _lib.check(lib.yds_darknet_forward_u8_dev(net._h, dev.offset(0), 1080,1920,B))
_lib.check(lib.yds_device_sync())
t=time.perf_counter(); N=int(os.environ.get("DET_LOOP_N", "30"))
for i in range(N): _lib.check(lib.yds_darknet_forward_u8_dev(net._h, dev.offset(0), 1080,1920,B))
_lib.check(lib.yds_device_sync())
dt=(time.perf_counter()-t)/N
print("graph" if os.environ.get("YDS_GRAPH") else "plain", "detector pass %.3f ms -> %.1f img/s" % (dt*1e3, B/dt))
