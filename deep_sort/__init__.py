"""Import-path shim: ``from deep_sort import DeepSort`` resolves to the MI355X-native tracker."""
from yolo_deepsort_amd.deep_sort import DeepSort, Extractor, build_tracker  # noqa: F401

__all__ = ["DeepSort", "build_tracker"]
