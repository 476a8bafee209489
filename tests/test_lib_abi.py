"""CPU-side checks of the C ABI: the library builds, loads and exports every declared symbol."""
import os
import re

from yolo_deepsort_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    build.build()
    lib = _lib.load()
    assert _lib.MISSING == []
    header = open(os.path.join(ROOT, "include", "ydsort.h")).read()
    declared = set(re.findall(r"\b(yds_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes found"
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, f"{name} lacks a ctypes prototype"
    assert set(_lib.SIGNATURES) <= declared


def test_no_cpu_fallback_without_gpu():
    import pytest
    lib = _lib.load()
    if lib.yds_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.YdsError):
        _lib.init(0)
    assert "no CPU path" in _lib.last_error()
