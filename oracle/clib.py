"""ctypes loader for the oracle's C helpers (TEST ORACLE; built with plain gcc)."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "oracle.c")
_OUT_DIR = os.path.join(_HERE, "_build")
_SO = os.path.join(_OUT_DIR, "liboracle.so")
_lib = None


def build(force=False):
    os.makedirs(_OUT_DIR, exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c11",
                               _SRC, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.lsap_f64.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p]
        _lib.lsap_f64.restype = ctypes.c_int
    return _lib


def lsap(cost):
    """cost [nr,nc] (any float dtype; converted to fp64 like scipy) -> (rows, cols) int64."""
    c = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = c.shape
    n = min(nr, nc)
    rows = np.zeros(n, np.int64)
    cols = np.zeros(n, np.int64)
    if n:
        rc = lib().lsap_f64(nr, nc, c.ctypes.data, rows.ctypes.data, cols.ctypes.data)
        if rc != 0:
            raise ValueError("cost matrix is infeasible")
    return rows, cols
