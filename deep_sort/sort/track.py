from yolo_deepsort_amd.sort_api import Track, TrackState  # noqa: F401
