from yolo_deepsort_amd.detect import p1p2Toxywh  # noqa: F401
