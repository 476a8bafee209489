"""Frame-by-frame latency of the reference-style API (ImageDetector.detect + DeepSort.update, batch of one, host frames):
tools/frame_latency.py   (run on the GPU box)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from yolo_deepsort_amd import _lib, cfgs, synth
from yolo_deepsort_amd.models import Darknet
from yolo_deepsort_amd.detect import VideoDetector
from yolo_deepsort_amd.deep_sort import DeepSort

_lib.init(0)
cfg = cfgs.cfg_text("yolov3", 608, 608)
net = Darknet(None, img_size=(608, 608), batch_max=1, cfg_text=cfg)
net.load_darknet_weights(None, blob=synth.darknet_weights_blob(cfg, 0))
scene = synth.PersonScene(30, frame_hw=(1080, 1920), seed=0)
heads = net.yolo_heads()
with tempfile.NamedTemporaryFile("w", suffix=".names", delete=False) as f:
    f.write(cfgs.coco_names_text())
ds = DeepSort(synth.reid_state_dict(0), use_cuda=True, max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)
vd = VideoDetector(net, f.name, thres=0.5, nms_thres=0.4, tracker=ds)
frames = [scene.frame(t) for t in range(40)]
times = []
for t, fr in enumerate(frames):
    ids, tlwh = scene.boxes(t)
    net.set_injection(0, synth.head_injection(tlwh, (1080, 1920), (608, 608), heads))
    t0 = time.perf_counter()
    out = vd.process(fr)
    times.append(time.perf_counter() - t0)
times = np.array(times[8:]) * 1e3
print(f"frame-by-frame: median {np.median(times):.2f} ms  ({1e3 / np.median(times):.0f} frames/s), rows in last frame: {len(out)}")
