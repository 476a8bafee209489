from yolo_deepsort_amd.sort_api import non_max_suppression  # noqa: F401
