from yolo_deepsort_amd.loaders import parse_model_config  # noqa: F401
