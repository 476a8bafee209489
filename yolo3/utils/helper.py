from yolo_deepsort_amd.loaders import load_classes  # noqa: F401
