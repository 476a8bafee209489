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


def test_default_device_under_per_rank_visibility_masks(monkeypatch):
    """with HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = one GPU per rank every rank sees ONE device: ordinal 0."""
    monkeypatch.delenv("YDS_DEVICE", raising=False)
    for k in _lib.VISIBILITY_MASKS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert _lib.default_device(n_visible=8) == 5            # no mask: LOCAL_RANK is the ordinal
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "5")
    assert _lib.default_device(n_visible=1) == 0            # masked to one device
    assert _lib.default_device(n_visible=8) == 5            # a mask that still shows all eight
    monkeypatch.setenv("YDS_DEVICE", "2")
    assert _lib.default_device(n_visible=1) == 2            # an explicit choice always wins


def test_exchange_block_round_trip():
    """dist.pack_rows / unpack_rows: {header, rows[R][6]} per frame; -1 = the detector returned None, -(2 + n) = n rows did not
    fit and the block has to grow (it does, in steps of 64 rows)."""
    import numpy as np
    import pytest
    from yolo_deepsort_amd import dist
    outs = [np.arange(18, dtype=np.int32).reshape(3, 6), None, np.zeros((0, 6), np.int32), []]
    blk = dist.pack_rows(outs)
    assert blk.shape == (4, 1 + 64 * 6) and blk[:, 0].tolist() == [3, -1, 0, 0]
    back = dist.unpack_rows(blk)
    assert np.array_equal(back[0], outs[0]) and back[1] is None and back[2].shape == (0, 6) and back[3].shape == (0, 6)
    big = [np.arange(257 * 6, dtype=np.int32).reshape(257, 6), None]
    small = dist.pack_rows(big, 64)
    assert small[:, 0].tolist() == [-259, -1] and dist.rows_needed(small) == 257
    with pytest.raises(ValueError):
        dist.unpack_rows(small)
    assert dist.rows_for(257) == 320 and dist.rows_for(0) == 64 and dist.rows_for(64) == 64
    grown = dist.pack_rows(big, dist.rows_for(257))
    assert grown.shape == (2, 1 + 320 * 6) and np.array_equal(dist.unpack_rows(grown)[0], big[0])


def test_constructors_do_not_name_a_device():
    """No product module hard-codes a device ordinal in an init call."""
    pkg = os.path.join(ROOT, "yolo_deepsort_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f != "_lib.py":
            assert not re.search(r"_lib\.init\(\s*\d", open(os.path.join(pkg, f)).read()), f
