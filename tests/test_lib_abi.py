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


class _FakeLib:
    """Records yds_init calls (host-logic test of the one-process-one-GPU rule, no device needed)."""

    def __init__(self):
        self.calls = []

    def yds_init(self, d):
        self.calls.append(d)
        return 0


def test_init_binds_one_device_per_process(monkeypatch):
    """constructors used to call init(0) and re-select GPU 0 on every rank.  Now the first init
    binds the process (LOCAL_RANK by default), later init() calls keep it and a different explicit device raises."""
    import pytest
    fake = _FakeLib()
    monkeypatch.setattr(_lib, "load", lambda: fake)
    monkeypatch.setattr(_lib, "_bound", None)
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.delenv("YDS_DEVICE", raising=False)
    assert _lib.init() == 1                      # rank 1 of torch.distributed.run -> GPU 1
    assert _lib.init() == 1 and _lib.init(1) == 1
    assert fake.calls == [1]                     # what Darknet/Extractor/DeepSort constructors do: no second yds_init
    with pytest.raises(_lib.YdsError):
        _lib.init(0)                             # the round-1 bug would have moved the process to GPU 0 here
    assert _lib.current_device() == 1
    monkeypatch.setattr(_lib, "_bound", None)
    monkeypatch.setenv("YDS_DEVICE", "3")
    assert _lib.init() == 3


def test_constructors_do_not_name_a_device():
    """No product module hard-codes a device ordinal in an init call."""
    pkg = os.path.join(ROOT, "yolo_deepsort_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f != "_lib.py":
            assert not re.search(r"_lib\.init\(\s*\d", open(os.path.join(pkg, f)).read()), f
