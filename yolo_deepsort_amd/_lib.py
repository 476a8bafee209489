"""ctypes binding of libydsort.so (the C ABI declared in include/ydsort.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception (``YdsError``) carrying ``yds_last_error()``.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_TAG = os.environ.get("YDS_BUILD_TAG", "")          # experiment builds (see build.py)
LIB_PATH = os.path.join(HERE, "libydsort" + ("_" + _TAG if _TAG else "") + ".so")


class YdsError(RuntimeError):
    pass


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_SZ = C.c_size_t
_I64 = C.c_int64

# name -> (restype, argtypes); mirrors include/ydsort.h one to one
SIGNATURES = {
    "yds_init": (_I, [_I]),
    "yds_last_error": (C.c_char_p, []),
    "yds_device_count": (_I, []),
    "yds_current_device": (_I, []),
    "yds_device_pci_bus_id": (_I, [C.c_char_p, _I]),
    "yds_build_info": (C.c_char_p, []),
    "yds_host_alloc": (_P, [_SZ]),
    "yds_host_free": (_I, [_P]),
    "yds_dev_alloc": (_P, [_SZ]),
    "yds_dev_free": (_I, [_P]),
    "yds_memcpy_h2d": (_I, [_P, _P, _SZ]),
    "yds_memcpy_d2h": (_I, [_P, _P, _SZ]),
    "yds_memcpy_d2d": (_I, [_P, _P, _SZ]),
    "yds_device_sync": (_I, []),
    "yds_darknet_create": (_P, [C.c_char_p, _I, _I, _I]),
    "yds_darknet_destroy": (None, [_P]),
    "yds_darknet_load_weights": (_I, [_P, _P, _SZ, _I]),
    "yds_darknet_set_batch_max": (_I, [_P, _I]),
    "yds_darknet_batch_max": (_I, [_P]),
    "yds_darknet_set_half": (_I, [_P, _I]),
    "yds_darknet_layer_format": (_I, [_P, _I]),
    "yds_darknet_num_boxes": (_I, [_P]),
    "yds_darknet_num_attrs": (_I, [_P]),
    "yds_darknet_num_layers": (_I, [_P]),
    "yds_darknet_layer_shape": (_I, [_P, _I, _P, _P, _P]),
    "yds_darknet_conv_flops": (_I64, [_P]),
    "yds_darknet_forward_f32": (_I, [_P, _P, _I, _P]),
    "yds_darknet_forward_u8": (_I, [_P, _P, _I, _I, _I, _P]),
    "yds_darknet_forward_u8_dev": (_I, [_P, _P, _I, _I, _I]),
    "yds_darknet_last_frames_dev": (_P, [_P, _P, _P, _P]),
    "yds_darknet_layer_output": (_I, [_P, _I, _I, _P]),
    "yds_darknet_get_input": (_I, [_P, _I, _P]),
    "yds_darknet_set_injection": (_I, [_P, _I, _P, _I, _F]),
    "yds_darknet_load_injection_sets": (_I, [_P, _P, _P, _I, _F]),
    "yds_darknet_select_injection_set": (_I, [_P, _I]),
    "yds_conv_variant_name": (C.c_char_p, [_I]),
    "yds_conv_timing_ex": (_I, [_P, _I, _P, _P, _P, _P, _P]),
    "yds_conv_num_variants": (_I, []),
    "yds_conv_clock": (_I, [_P, _P, _I]),
    "yds_set_conv_math": (_I, [_I]),
    "yds_get_conv_math": (_I, []),
    "yds_conv_bench": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "yds_conv_run": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "yds_nms": (_I, [_P, _I, _F, _F, _I, _I, _P, _I, _P]),
    "yds_nms_pred": (_I, [_P, _I, _I, _F, _F, _P, _I, _P]),
    "yds_nms_merge_pred": (_I, [_P, _I, _I, _F, _F, _P, _I, _P]),
    "yds_detect_tiled": (_I, [_P, _P, _I, _I, _P, _I, _F, _F, _P, _I, _P]),
    "yds_reid_create": (_P, [_I]),
    "yds_reid_destroy": (None, [_P]),
    "yds_reid_load_tensor": (_I, [_P, C.c_char_p, _P, _P, _I]),
    "yds_reid_finalize": (_I, [_P]),
    "yds_reid_flops_per_crop": (_I64, []),
    "yds_reid_embed": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "yds_reid_embed_dev": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "yds_reid_features_dev": (_P, [_P]),
    "yds_reid_preprocess": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "yds_reid_forward_f32": (_I, [_P, _P, _I, _P]),
    "yds_tracker_create": (_P, [C.c_double, C.c_double, _I, _I, _I]),
    "yds_tracker_create_ex": (_P, [C.c_double, C.c_double, _I, _I, _I, _I]),
    "yds_tracker_step_sel": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P]),
    "yds_tracker_nms": (_I, [_P, _P, _I, C.c_double, _P, _P]),
    "yds_euclidean_min_cost": (_I, [_P, _P, _I, _P, _I, _I, _P]),
    "yds_tracker_destroy": (None, [_P]),
    "yds_tracker_step": (_I, [_P, _P, _P, _P, _I, _P, _I, _P, _P, _I, _P]),
    "yds_tracker_step_dev": (_I, [_P, _P, _P, _P, _I, _P, _I, _P]),
    "yds_tracker_num_tracks": (_I, [_P]),
    "yds_tracker_get_state": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "yds_comm_unique_id": (_I, [_P]),
    "yds_comm_create": (_P, [_P, _I, _I]),
    "yds_comm_destroy": (None, [_P]),
    "yds_comm_world": (_I, [_P]),
    "yds_comm_rank": (_I, [_P]),
    "yds_comm_rccl_version": (_I, []),
    "yds_comm_allgather": (_I, [_P, _P, _SZ, _P]),
    "yds_comm_allgather_dev": (_I, [_P, _P, _SZ, _P]),
    "yds_comm_allgather_rows": (_I, [_P, _P, _I, _P, _I, _I, _P, _P]),
    "yds_comm_preflight": (_I, []),
    "yds_comm_allreduce_f64": (_I, [_P, _P, _I, _I]),
    "yds_comm_barrier": (_I, [_P]),
    "yds_tracker_get_payload": (_I, [_P, _P, _I]),
    "yds_tracker_get_age": (_I, [_P, _P, _I]),
    "yds_tracker_gallery_rows": (_I, [_P]),
    "yds_kalman_gating_ex": (_I, [_P, _P, _I, _P, _I, _I, _P]),
    "yds_kalman_initiate": (_I, [_P, _I, _P, _P]),
    "yds_kalman_project": (_I, [_P, _P, _I, _P, _P]),
    "yds_tracker_last_unmatched": (_I, [_P, _P, _I, _P, _P, _I, _P]),
    "yds_lsap": (_I, [_P, _I, _I, _P, _P, _P]),
    "yds_lsap_bench": (_I, [_P, _I, _I, _I, _P]),
    "yds_kalman_predict": (_I, [_P, _P, _I]),
    "yds_kalman_update": (_I, [_P, _P, _P, _I]),
    "yds_kalman_gating": (_I, [_P, _P, _I, _P, _I, _P]),
    "yds_iou_cost": (_I, [_P, _I, _P, _I, _P]),
    "yds_cosine_min_cost": (_I, [_P, _P, _I, _P, _I, _I, _P]),
    "yds_pipeline_create": (_P, [_P, _P, _P, _F, _F, _P, _I]),
    "yds_pipeline_destroy": (None, [_P]),
    "yds_pipeline_step": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "yds_pipeline_step_host": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "yds_pipeline_prefetch_host": (_I, [_P, _P, _I, _I, _I]),
    "yds_pipeline_set_next_injection": (_I, [_P, _I]),
    "yds_pipeline_stage_us": (_I, [_P, _P]),
    "yds_pipeline_set_schedule": (_I, [_P, _I]),
    "yds_pipeline_last_schedule": (_I, [_P]),
    "yds_pipeline_schedule_trial": (_I, [_P, _I, _P, _P, _P]),
    "yds_conv_timing": (_I, [_P, _I, _P, _P, _P]),
    "yds_overlay_tracks": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "yds_swap_rb": (_I, [_P, _SZ]),
    "yds_overlay_tracks_bgr": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "yds_pipeline_set_frame_order": (_I, [_P, _I]),
}

_lib = None
MISSING = []


def load():
    """Load libydsort.so and declare every prototype.  Raises YdsError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YdsError(f"{LIB_PATH} not found: build it with `python -m yolo_deepsort_amd.build` "
                       "(the product has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:       # stale .so: calling the symbol raises, tests assert MISSING == []
            MISSING.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().yds_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise YdsError(last_error())


def check_ptr(p):
    if not p:
        raise YdsError(last_error())
    return p


_bound = None


VISIBILITY_MASKS = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL")


def default_device(n_visible=None):
    """Device of this process when the caller names none: YDS_DEVICE, else LOCAL_RANK (one process per GPU under
    torch.distributed.run), else 0.  A launcher that masks the devices per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES =
    one GPU each) leaves every rank with exactly one visible device: its ordinal is 0 whatever LOCAL_RANK says."""
    v = os.environ.get("YDS_DEVICE")
    if v not in (None, ""):
        return int(v)
    v = os.environ.get("LOCAL_RANK")
    if v in (None, ""):
        return 0
    masked = any(os.environ.get(k) not in (None, "") for k in VISIBILITY_MASKS)
    if masked:
        n = load().yds_device_count() if n_visible is None else n_visible
        if n == 1:
            return 0
    return int(v)


def init(device=None):
    """Bind this process to ONE GPU (yds_init); raises when no MI355X is visible.  ``device=None`` keeps the device
    that is already bound, or picks ``default_device()`` on the first call.  Asking for a different device after the
    first call raises: every handle of the process lives on the bound GPU (multi-GPU = one process per GPU)."""
    global _bound
    if _bound is not None:
        if device is not None and int(device) != _bound:
            raise YdsError(f"this process is bound to device {_bound}; one process drives one GPU (requested {int(device)})")
        return _bound
    dev = default_device() if device is None else int(device)
    check(load().yds_init(dev))
    _bound = dev
    return dev


def current_device():
    return _bound


def pci_bus_id():
    buf = C.create_string_buffer(64)
    check(load().yds_device_pci_bus_id(buf, 64))
    return buf.value.decode()


def ptr(a):
    """Data pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class PinnedArray:
    """numpy view of pinned host memory (yds_host_alloc): uploads from it are asynchronous (decoders write frames here)."""

    def __init__(self, shape, dtype=np.uint8):
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = check_ptr(load().yds_host_alloc(self.nbytes))
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    @classmethod
    def from_array(cls, a):
        a = np.asarray(a)
        p = cls(a.shape, a.dtype)
        np.copyto(p.array, a)
        return p

    def free(self):
        if self.ptr:
            self.array = None
            load().yds_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """Raw HBM allocation owned by Python (frames kept resident for the bench)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = check_ptr(load().yds_dev_alloc(self.nbytes))

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        check(load().yds_memcpy_h2d(buf.ptr, ptr(a), a.nbytes))
        return buf

    def offset(self, nbytes):
        return C.c_void_p(self.ptr + int(nbytes))

    def free(self):
        if self.ptr:
            load().yds_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
