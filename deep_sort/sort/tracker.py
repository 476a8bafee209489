from yolo_deepsort_amd.sort_api import Tracker  # noqa: F401
