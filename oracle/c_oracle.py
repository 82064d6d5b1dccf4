"""ctypes view of oracle/wdbx_oracle.c (test infrastructure only)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = _DIR / "libwdbx_oracle.so"
        if not so.exists():
            subprocess.run(["make", "-C", str(_DIR), "all"], check=True)
        _lib = C.CDLL(str(so))
        _lib.wdbx_oracle_flat_search.restype = C.c_int
        _lib.wdbx_oracle_flat_search.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.wdbx_oracle_normalize_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        _lib.wdbx_oracle_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p]
    return _lib


def flat_search(rows, query, k, metric=0, allowed=None):
    rows = np.ascontiguousarray(rows, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    idx = np.empty(k, np.int64)
    score = np.empty(k, np.float32)
    al = None if allowed is None else np.ascontiguousarray(allowed, np.uint8)
    n = lib().wdbx_oracle_flat_search(rows.ctypes.data, rows.shape[0], rows.shape[1], query.ctypes.data, int(k), int(metric),
                                      None if al is None else al.ctypes.data, idx.ctypes.data, score.ctypes.data)
    return idx[:n], score[:n]


def normalize_rows(rows):
    out = np.array(rows, np.float32, order="C", copy=True)
    lib().wdbx_oracle_normalize_rows(out.ctypes.data, out.shape[0], out.shape[1])
    return out


def synth_rows(seed, row0, n, d):
    out = np.empty((n, d), np.float32)
    lib().wdbx_oracle_synth_rows(int(seed), int(row0), int(n), int(d), out.ctypes.data)
    return out
