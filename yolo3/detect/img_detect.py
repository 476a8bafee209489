from yolo_deepsort_amd.detect import ImageDetector  # noqa: F401
