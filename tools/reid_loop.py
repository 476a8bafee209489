import sys, numpy as np
sys.path.insert(0, "/root/repo")
from yolo_deepsort_amd import synth
from yolo_deepsort_amd.deep_sort import Extractor
sd = synth.reid_state_dict(0)
ex = Extractor(sd, max_crops=512)
x = np.random.RandomState(1).randn(480, 3, 128, 64).astype(np.float32)
for i in range(12):
    y = ex.forward(x)
print(float(np.abs(y).sum()))
