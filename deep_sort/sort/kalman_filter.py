from yolo_deepsort_amd.sort_api import KalmanFilter, chi2inv95  # noqa: F401
