"""Import-path shim: the module-level names of the reference's yolo3/utils/model_build.py that the detect path uses."""
from yolo_deepsort_amd.model_build import (bbox_iou, epsilon, p1p2Toxywh, resize_boxes,  # noqa: F401
                                           soft_non_max_suppression, xywh2p1p2)
