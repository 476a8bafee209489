"""Import-path shim: deep_sort/deep/feature_extractor.py:12 ``Extractor`` on libydsort."""
from yolo_deepsort_amd.deep_sort import Extractor  # noqa: F401
