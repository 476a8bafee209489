from yolo_deepsort_amd.models import Darknet  # noqa: F401
