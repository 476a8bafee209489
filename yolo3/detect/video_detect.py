from yolo_deepsort_amd.detect import VideoDetector  # noqa: F401
