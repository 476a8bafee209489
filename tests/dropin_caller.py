"""A caller of the drop-in boundary written for this repo (not a transcription of any reference script).

It reaches every class through the REFERENCE'S IMPORT PATHS (`yolo3.models.Darknet`,
`yolo3.detect.video_detect.VideoDetector`, `deep_sort.DeepSort`, `action.action_Identify.ActionIdentify`, `action.actions.*`)
and passes the keyword names of section 8(b)'s signature table; the values live in the tables below.  Run as
`python dropin_caller.py <repo root> <half: 0|1>` from a scratch directory that holds cfg / weights / names / frames.npy;
prints one `ROWS <json>` line: per yielded frame None or the tracker's int rows."""
import importlib
import json
import sys

import numpy as np

# constructor signature table: import path, class, positional arguments, keywords
TRACKER_KW = dict(max_dist=0.3, min_confidence=1, nn_budget=30, max_iou_distance=0.7, max_age=30, n_init=3, use_cuda=True)
DETECTOR_KW = dict(thres=0.5, nms_thres=0.4, skip_frames=2, class_mask=[0, 2, 4], thickness=2)
ACTIONS = [("TakeOff", (4,), dict(delta=(0, 1))), ("Landing", (4,), dict(delta=(2, 2))), ("Glide", (4,), dict(delta=(1, 2))),
           ("FastCrossing", (4,), dict(speed=0.2)), ("BreakInto", (0,), dict(timeout=2))]
FILES = dict(cfg="config/yolov4.cfg", weights="weights/yolov4.weights", ckpt="weights/ckpt.t7", names="config/coco.names",
             source="frames.npy", sink="out.npy")


def at(module, name):
    return getattr(importlib.import_module(module), name)


def build(half):
    net = at("yolo3.models", "Darknet")(FILES["cfg"], img_size=(608, 608))
    net.load_darknet_weights(FILES["weights"])
    net.to("cuda:0")
    tracker = at("deep_sort", "DeepSort")(FILES["ckpt"], **TRACKER_KW)
    act = importlib.import_module("action.actions")
    judge = at("action.action_Identify", "ActionIdentify")(actions=[getattr(act, n)(*a, **k) for n, a, k in ACTIONS],
                                                           max_age=30, max_size=8)
    return at("yolo3.detect.video_detect", "VideoDetector")(net, FILES["names"], tracker=tracker, action_id=judge, half=half,
                                                            **DETECTOR_KW)


def main(root, half):
    sys.path.insert(0, root)
    vd = build(bool(int(half)))
    rows = []
    for image, detections, actions in vd.detect(FILES["source"], output_path=FILES["sink"], real_show=False, skip_secs=0):
        if image.dtype != np.uint8 or image.shape != (270, 480, 3):
            raise SystemExit(f"unexpected frame {image.dtype} {image.shape}")
        rows.append(None if detections is None else np.asarray(detections, np.int32).reshape(-1, 6).tolist())
    print("ROWS", json.dumps(rows))


if __name__ == "__main__":
    main(*sys.argv[1:3])
