from yolo_deepsort_amd.action import Orbit  # noqa: F401
