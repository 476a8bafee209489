"""numpy restatement of the DeepSORT appearance network (TEST ORACLE).

Net(reid=True).forward: reference deep_sort/deep/model.py:48-95 (BasicBlock :5-37).
Crop + preprocessing: deep_sort/deep_sort.py:116-122,133-146 and
deep_sort/deep/feature_extractor.py:30-58.
"""

import numpy as np

from .darknet import conv2d_nchw, batchnorm_eval, maxpool_nchw
from .resize import resize_bilinear_u8

F32 = np.float32
MEAN = np.array([0.485, 0.456, 0.406], dtype=F32)
STD = np.array([0.229, 0.224, 0.225], dtype=F32)
STAGES = (("layer1", 64, 64, False), ("layer2", 64, 128, True),
          ("layer3", 128, 256, True), ("layer4", 256, 512, True))


def _bn(x, sd, prefix):
    return batchnorm_eval(x, sd[prefix + ".weight"], sd[prefix + ".bias"],
                          sd[prefix + ".running_mean"], sd[prefix + ".running_var"])


def _block(x, sd, pfx, downsample):
    """BasicBlock.forward model.py:28-37"""
    y = conv2d_nchw(x, sd[pfx + ".conv1.weight"], None, 2 if downsample else 1, 1)
    y = np.maximum(_bn(y, sd, pfx + ".bn1"), 0)
    y = conv2d_nchw(y, sd[pfx + ".conv2.weight"], None, 1, 1)
    y = _bn(y, sd, pfx + ".bn2")
    if downsample:
        x = conv2d_nchw(x, sd[pfx + ".downsample.0.weight"], None, 2, 0)
        x = _bn(x, sd, pfx + ".downsample.1")
    return np.maximum(x + y, 0).astype(F32)


def reid_forward(x, sd):
    """x [D,3,128,64] fp32 normalised -> [D,512] unit-norm rows (model.py:81-92)."""
    x = conv2d_nchw(x.astype(F32), sd["conv.0.weight"], sd["conv.0.bias"], 1, 1)
    x = np.maximum(_bn(x, sd, "conv.1"), 0)
    x = maxpool_nchw(x, 3, 2, 1)
    for name, cin, cout, down in STAGES:
        x = _block(x, sd, name + ".0", down)
        x = _block(x, sd, name + ".1", False)
    x = x.mean(axis=(2, 3), dtype=F32)                    # AvgPool2d((8,4),1) on an 8x4 map
    nrm = np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    return (x / nrm).astype(F32)


def crop_boxes(tlwh, frame_h, frame_w):
    """deep_sort.py:116-122  python int() truncation then clip; returns int [D,4] (x1,y1,x2,y2)."""
    out = []
    for x, y, w, h in np.asarray(tlwh, dtype=F32):
        x1 = max(int(x), 0)
        x2 = min(int(F32(x + w)), frame_w - 1)
        y1 = max(int(y), 0)
        y2 = min(int(F32(y + h)), frame_h - 1)
        out.append((x1, y1, x2, y2))
    return np.array(out, dtype=np.int32).reshape(-1, 4)


def preprocess_crops(frame, tlwh):
    """frame uint8 [H,W,3] RGB, tlwh [D,4] -> fp32 [D,3,128,64] (feature_extractor.py:34-51)."""
    H, W = frame.shape[:2]
    batch = []
    for x1, y1, x2, y2 in crop_boxes(tlwh, H, W):
        crop = frame[y1:y2, x1:x2]
        r = resize_bilinear_u8(crop, (64, 128)).astype(F32).transpose(2, 0, 1)
        batch.append(r)
    b = (np.stack(batch, 0) / F32(255.)).astype(F32)
    b = ((b - MEAN[None, :, None, None]) / STD[None, :, None, None]).astype(F32)
    return b
