"""Host-side loaders kept drop-in with the reference.

parse_model_config: yolo3/utils/parse_config.py:1-19
load_classes:       yolo3/utils/helper.py:8-14
load_reid_checkpoint: Extractor.__init__, deep_sort/deep/feature_extractor.py:13-17
"""

from __future__ import annotations

import numpy as np


def parse_model_config(path, text=None):
    """Parses a Darknet cfg into a list of dicts, exactly like the reference: blank and
    '#' lines dropped, fringe whitespace stripped, values stay strings and
    ``convolutional`` blocks default to ``batch_normalize = 0`` (int)."""
    if text is None:
        with open(path, "r") as f:
            text = f.read()
    module_defs = []
    for line in text.split("\n"):
        if not line or line.startswith("#"):
            continue
        line = line.strip()
        if line.startswith("["):
            block = {"type": line[1:-1].rstrip()}
            if block["type"] == "convolutional":
                block["batch_normalize"] = 0
            module_defs.append(block)
        else:
            key, value = line.split("=")
            module_defs[-1][key.rstrip()] = value.strip()
    return module_defs


def load_classes(path):
    """Class labels, one per line; the file must end with a newline (last split element dropped)."""
    with open(path, "r", encoding="utf-8") as fp:
        return fp.read().split("\n")[:-1]


# ---------------------------------------------------------------------------------------------------
# ckpt.t7 reader without torch.  ``torch.save`` writes either a zip archive (torch >= 1.6: ``<name>/data.pkl`` plus one
# raw little-endian file per storage under ``<name>/data/``) or the legacy stream (three header pickles, the object
# pickle, the list of storage keys, then per key an int64 element count followed by the raw data).  In both, tensors
# are pickled as ``torch._utils._rebuild_tensor_v2(storage, offset, size, stride, ...)`` with the storage given by a
# persistent id ``('storage', <torch.XStorage>, key, location, numel[, view])``.  The unpickler below resolves exactly
# those names (plus OrderedDict) to numpy builders and refuses every other global, so reading a checkpoint cannot
# execute code - unlike ``torch.load(..., weights_only=False)``.
_STORAGE_DTYPES = {
    "FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16, "LongStorage": np.int64,
    "IntStorage": np.int32, "ShortStorage": np.int16, "CharStorage": np.int8, "ByteStorage": np.uint8, "BoolStorage": np.bool_,
}
_LEGACY_MAGIC = 0x1950A86A20F9469CFC6C


class _StorageType:
    def __init__(self, name):
        self.dtype = np.dtype(_STORAGE_DTYPES[name])


class _LazyStorage:
    """Storage named by a persistent id; ``data`` is filled when its bytes are read."""

    def __init__(self, dtype, key, numel):
        self.dtype, self.key, self.numel, self.data = dtype, key, int(numel), None


class _LazyTensor:
    def __init__(self, storage, offset, size, stride):
        self.storage, self.offset, self.size, self.stride = storage, int(offset), tuple(size), tuple(stride)

    def numpy(self):
        flat = self.storage.data
        if flat is None:
            raise ValueError(f"storage {self.storage.key} was never read")
        item = flat.dtype.itemsize
        # the geometry comes straight from the (untrusted) pickle: every element it addresses must lie inside the storage
        size, stride = tuple(int(n) for n in self.size), tuple(int(st) for st in self.stride)
        if len(size) != len(stride) or self.offset < 0 or any(n < 0 for n in size) or any(st < 0 for st in stride):
            raise ValueError(f"tensor geometry of storage {self.storage.key} is malformed (offset {self.offset}, size {size}, stride {stride})")
        if all(n > 0 for n in size):        # (an empty tensor addresses nothing)
            last = self.offset + sum((n - 1) * st for n, st in zip(size, stride))
            if last >= flat.size:
                raise ValueError(f"tensor of storage {self.storage.key} reaches element {last} of {flat.size}: refused")
        else:
            return np.zeros(size, dtype=flat.dtype)
        v = np.lib.stride_tricks.as_strided(flat[self.offset:], shape=size, strides=tuple(st * item for st in stride))
        return np.array(v)            # contiguous copy (0-dim tensors included)


def _rebuild_tensor(storage, storage_offset, size, stride, *unused):
    return _LazyTensor(storage, storage_offset, size, stride)


def _rebuild_parameter(data, *unused):
    return data


def _make_unpickler(file, storages):
    import collections
    import pickle

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == "collections" and name == "OrderedDict":
                return collections.OrderedDict
            if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return _rebuild_tensor
            if module == "torch._utils" and name in ("_rebuild_parameter", "_rebuild_parameter_with_state"):
                return _rebuild_parameter
            if module == "torch" and name in _STORAGE_DTYPES:
                return _StorageType(name)
            raise pickle.UnpicklingError(f"checkpoint references {module}.{name}: refused (only tensors and containers are read)")

        def persistent_load(self, pid):
            if not isinstance(pid, tuple) or pid[0] != "storage":
                raise pickle.UnpicklingError(f"unsupported persistent id {pid!r}")
            stype, key, numel = pid[1], str(pid[2]), pid[4]
            if len(pid) > 5 and pid[5] is not None:
                raise pickle.UnpicklingError("storage views are not supported")
            if key not in storages:
                storages[key] = _LazyStorage(stype.dtype, key, numel)
            return storages[key]

    return Unpickler(file)


def _materialise(obj):
    if isinstance(obj, _LazyTensor):
        return obj.numpy()
    if isinstance(obj, dict):
        return type(obj)((k, _materialise(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_materialise(v) for v in obj)
    return obj


def read_torch_checkpoint(path):
    """The object stored by ``torch.save`` with tensors as numpy arrays; both on-disk formats, no torch import."""
    import pickle
    import zipfile
    storages = {}
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            pkl = [n for n in z.namelist() if n.endswith("/data.pkl") or n == "data.pkl"]
            if len(pkl) != 1:
                raise ValueError(f"{path}: not a torch zip checkpoint")
            root = pkl[0][:-len("data.pkl")]
            if root + "byteorder" in z.namelist() and z.read(root + "byteorder").strip() != b"little":
                raise ValueError(f"{path}: big-endian checkpoints are not supported")
            with z.open(pkl[0]) as f:
                obj = _make_unpickler(f, storages).load()
            for key, st in storages.items():
                raw = z.read(f"{root}data/{key}")
                st.data = np.frombuffer(raw, dtype=st.dtype, count=st.numel)
        return _materialise(obj)
    with open(path, "rb") as f:
        if pickle.load(f) != _LEGACY_MAGIC:                    # three plain header pickles: ints and a dict of ints/bools
            raise ValueError(f"{path}: neither a zip nor a legacy torch checkpoint")
        pickle.load(f)                                           # protocol version
        pickle.load(f)                                           # sys info
        obj = _make_unpickler(f, storages).load()
        keys = pickle.load(f)
        for key in keys:
            st = storages.get(str(key))
            numel = int(np.frombuffer(f.read(8), dtype="<i8")[0])
            if st is None:
                raise ValueError(f"{path}: storage {key} is not referenced by the object")
            if numel != st.numel or numel < 0:
                raise ValueError(f"{path}: storage {key} holds {numel} elements, the object expects {st.numel}")
            st.data = np.frombuffer(f.read(numel * st.dtype.itemsize), dtype=st.dtype, count=numel)
    return _materialise(obj)


def load_reid_checkpoint(path):
    """``torch.load(path)['net_dict']`` (deep_sort/deep/feature_extractor.py:16) as {name: ndarray}.

    Read by the torch-free reader above.  If that fails and torch is installed, ``torch.load(weights_only=True)`` is
    tried; the code-executing ``weights_only=False`` load is only used when YDS_ALLOW_UNSAFE_PICKLE=1 is set."""
    import os
    try:
        sd = read_torch_checkpoint(path)["net_dict"]
        return {k: np.asarray(v) for k, v in sd.items()}
    except Exception as first:
        try:
            import torch
        except ImportError:
            raise first
        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
        except Exception:
            if os.environ.get("YDS_ALLOW_UNSAFE_PICKLE") != "1":
                raise first
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
        return {k: v.detach().cpu().numpy() for k, v in ckpt["net_dict"].items()}
