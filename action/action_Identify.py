from yolo_deepsort_amd.action import ActionIdentify  # noqa: F401
