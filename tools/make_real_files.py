"""Writes a Darknet .weights file and a ckpt.t7 in the REAL file formats (synthetic values) for bench.py --weights/--ckpt on a box
that has no downloaded weights (objectness bias -20: random filters must not "detect" - a random net's boxes include empty crops,
on which the reference's cv2.resize raises and so does this package; the leg then times detector + NMS on real-format files): tools/make_real_files.py <dir> [net]  ->  <dir>/<net>.weights, <dir>/ckpt.t7"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_deepsort_amd import cfgs, synth      # noqa: E402

out, net = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "yolov3")
os.makedirs(out, exist_ok=True)
cfg = cfgs.cfg_text(net, 608, 608)
open(os.path.join(out, net + ".weights"), "wb").write(synth.darknet_weights_blob(cfg, seed=0, obj_bias=-20.0))
import torch                                    # noqa: E402
sd = synth.reid_state_dict(0)
torch.save({"net_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "acc": 0.9, "epoch": 40}, os.path.join(out, "ckpt.t7"))
print(out)
