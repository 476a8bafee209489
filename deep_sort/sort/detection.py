from yolo_deepsort_amd.sort_api import Detection  # noqa: F401
