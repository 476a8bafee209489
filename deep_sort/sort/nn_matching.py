from yolo_deepsort_amd.sort_api import NearestNeighborDistanceMetric  # noqa: F401
