"""Import-path shim: ``from deep_sort.deep_sort import DeepSort`` (deep_sort/deep_sort.py:15)."""
from yolo_deepsort_amd.deep_sort import DeepSort  # noqa: F401

__all__ = ["DeepSort"]
