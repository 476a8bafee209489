from yolo_deepsort_amd.sort_api import iou, iou_cost  # noqa: F401
