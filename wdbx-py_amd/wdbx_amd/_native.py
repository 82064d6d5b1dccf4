"""ctypes binding of ``libwdbx_hip.so`` (C ABI declared in ``include/wdbx_hip.h``).

No PyTorch and no CPU fallback: if the library is missing, or no AMD GPU is
visible, the backend raises ``HipBackendError`` -- it never computes a result
any other way.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

METRIC_COSINE = 0
METRIC_L2 = 1
MAX_K = 2048
UNIQUE_ID_BYTES = 128

_LIB_NAME = "libwdbx_hip.so"


class HipBackendError(RuntimeError):
    """Raised for every failure of the HIP backend (code + library message)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"wdbx_hip error {code}: {message}")
        self.code = code
        self.message = message


_lib: Optional[C.CDLL] = None

_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_dblp = C.POINTER(C.c_double)

# name -> (restype, argtypes); the one place the ABI is spelled out for Python.
SIGNATURES = {
    "wdbx_hip_version": (C.c_int, []),
    "wdbx_last_error": (C.c_char_p, []),
    "wdbx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "wdbx_index_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "wdbx_index_destroy": (None, [C.c_void_p]),
    "wdbx_index_dim": (C.c_int, [C.c_void_p]),
    "wdbx_index_row_pitch": (C.c_int, [C.c_void_p]),
    "wdbx_index_size": (C.c_int, [C.c_void_p, _u64p]),
    "wdbx_index_capacity": (C.c_int, [C.c_void_p, _u64p]),
    "wdbx_index_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "wdbx_index_clear": (C.c_int, [C.c_void_p]),
    "wdbx_index_add": (C.c_int, [C.c_void_p, _f32p, C.c_uint64, C.c_int, _u64p]),
    "wdbx_index_set_rows": (C.c_int, [C.c_void_p, C.c_uint64, _f32p, C.c_uint64, C.c_int]),
    "wdbx_index_get_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _f32p]),
    "wdbx_index_fill_synthetic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _u64p]),
    "wdbx_index_compact": (C.c_int, [C.c_void_p, _u64p, C.c_uint64]),
    "wdbx_index_search": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _i64p, _f32p]),
    "wdbx_index_search_masked": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), _i64p,
                                           _f32p]),
    "wdbx_index_search_masked_n": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_uint64,
                                             _i64p, _f32p]),
    "wdbx_device_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "wdbx_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wdbx_device_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "wdbx_device_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "wdbx_device_fill_synthetic": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]),
    "wdbx_index_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wdbx_index_synchronize": (C.c_int, [C.c_void_p]),
    "wdbx_index_search_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wdbx_index_batch_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_int)]),
    "wdbx_index_profile_read_gemm": (C.c_int, [C.c_void_p, _u64p, _dblp]),
    "wdbx_index_profile_read_sample": (C.c_int, [C.c_void_p, _u64p, _dblp]),
    "wdbx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "wdbx_index_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64]),
    "wdbx_index_comm_destroy": (C.c_int, [C.c_void_p]),
    "wdbx_index_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u64p]),
    "wdbx_index_comm_set_row_base": (C.c_int, [C.c_void_p, C.c_uint64]),
    "wdbx_index_search_sharded_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wdbx_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "wdbx_group_destroy": (None, [C.c_void_p]),
    "wdbx_group_add": (C.c_int, [C.c_void_p, _f32p, C.c_uint64, C.c_int, _u64p]),
    "wdbx_group_size": (C.c_int, [C.c_void_p, _u64p]),
    "wdbx_group_search": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _i64p, _f32p]),
    "wdbx_group_attach": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "wdbx_group_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u64p]),
    "wdbx_group_stat": (C.c_int, [C.c_void_p, C.c_char_p, _i64p]),
    "wdbx_group_search_merged": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _i64p, _f32p]),
    "wdbx_group_search_merged_masked": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(C.POINTER(C.c_uint32)), _i64p, _f32p]),
    "wdbx_group_search_merged_masked_n": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(C.POINTER(C.c_uint32)), _u64p, _i64p, _f32p]),
    "wdbx_group_attach_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wdbx_group_set_row_bases": (C.c_int, [C.c_void_p, _u64p, C.c_int]),
    "wdbx_group_queries_upload": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int]),
    "wdbx_group_queries_synthetic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
    "wdbx_group_search_resident": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "wdbx_group_synchronize": (C.c_int, [C.c_void_p]),
    "wdbx_group_results": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _i64p, _f32p]),
    "wdbx_index_comm_allgather_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "wdbx_index_search_sharded_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                         C.c_void_p]),
    "wdbx_index_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "wdbx_index_profile_read": (C.c_int, [C.c_void_p, _u64p, _dblp, _u64p, _dblp]),
    "wdbx_index_probe_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dblp]),
    "wdbx_index_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "wdbx_index_get_option": (C.c_int, [C.c_void_p, C.c_char_p, _i64p]),
}


def library_path() -> Path:
    env = os.environ.get("WDBX_HIP_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / _LIB_NAME


def load_library() -> C.CDLL:
    """Load the shared library and declare every entry point.  Raises
    ``HipBackendError`` when it is absent (build it with
    ``python __graft_entry__.py`` or ``make -C wdbx-py_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise HipBackendError(-100, f"{path} not built: run `make -C wdbx-py_amd/csrc` (needs hipcc)")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # missing ROCm runtime etc.
        raise HipBackendError(-101, f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().wdbx_last_error()
        raise HipBackendError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    rc = load_library().wdbx_device_count(C.byref(n))
    if rc != 0:
        return 0
    return n.value


def _as_f32(a, shape_last: int) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if arr.ndim == 1:
        arr = arr.reshape(1, -1)
    if arr.ndim != 2 or arr.shape[1] != shape_last:
        raise ValueError(f"expected rows of length {shape_last}, got array of shape {arr.shape}")
    return arr


def pack_row_mask(allowed: np.ndarray) -> np.ndarray:
    """bool[n] -> uint32 words for ``wdbx_index_search_masked`` (bit r%32 of word r//32)."""
    allowed = np.asarray(allowed, dtype=bool)
    padded = np.zeros(((allowed.size + 31) // 32) * 32, dtype=bool)
    padded[: allowed.size] = allowed
    return np.packbits(padded, bitorder="little").view(np.uint32)


class DeviceBuffer:
    """A raw HBM allocation owned through the library (no torch)."""

    def __init__(self, index: "NativeIndex", nbytes: int):
        self._index = index
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(index._lib.wdbx_device_alloc(index._h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host: np.ndarray) -> None:
        host = np.ascontiguousarray(host)
        if host.nbytes > self.nbytes:
            raise ValueError("upload larger than the device buffer")
        _check(self._index._lib.wdbx_device_upload(self._index._h, self.ptr, host.ctypes.data, host.nbytes))

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than the device buffer")
        _check(self._index._lib.wdbx_device_download(self._index._h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr and self._index._h:
            _check(self._index._lib.wdbx_device_free(self._index._h, self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class NativeIndex:
    """Thin object wrapper over one ``wdbx_index*`` (one shard in one GPU's HBM)."""

    def __init__(self, dim: int, metric: int = METRIC_COSINE, device_id: int = 0, capacity_rows: int = 1024):
        self._lib = load_library()
        self._h = None
        h = C.c_void_p()
        _check(self._lib.wdbx_index_create(int(device_id), int(dim), int(metric), int(capacity_rows), C.byref(h)))
        self._h = h.value
        self.dim = int(dim)
        self.metric = int(metric)
        self.device_id = int(device_id)
        self.pitch = int(self._lib.wdbx_index_row_pitch(self._h))

    # -- lifecycle --
    def close(self) -> None:
        if self._h:
            self._lib.wdbx_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state --
    def size(self) -> int:
        n = C.c_uint64(0)
        _check(self._lib.wdbx_index_size(self._h, C.byref(n)))
        return n.value

    def capacity(self) -> int:
        n = C.c_uint64(0)
        _check(self._lib.wdbx_index_capacity(self._h, C.byref(n)))
        return n.value

    def reserve(self, rows: int) -> None:
        _check(self._lib.wdbx_index_reserve(self._h, int(rows)))

    def clear(self) -> None:
        _check(self._lib.wdbx_index_clear(self._h))

    # -- ingest --
    def add(self, rows, normalize: bool = False) -> int:
        arr = _as_f32(rows, self.dim) if np.size(rows) else np.empty((0, self.dim), np.float32)
        first = C.c_uint64(0)
        _check(self._lib.wdbx_index_add(self._h, arr.ctypes.data_as(_f32p), arr.shape[0], int(normalize),
                                        C.byref(first)))
        return first.value

    def set_rows(self, first_row: int, rows, normalize: bool = False) -> None:
        arr = _as_f32(rows, self.dim)
        _check(self._lib.wdbx_index_set_rows(self._h, int(first_row), arr.ctypes.data_as(_f32p), arr.shape[0],
                                             int(normalize)))

    def get_rows(self, first_row: int, n: int) -> np.ndarray:
        out = np.empty((int(n), self.dim), np.float32)
        _check(self._lib.wdbx_index_get_rows(self._h, int(first_row), int(n), out.ctypes.data_as(_f32p)))
        return out

    def compact(self, src_rows) -> None:
        """Keep exactly the rows ``src_rows`` (strictly increasing), moved down to rows 0 .. len - 1 in that order."""
        src = np.ascontiguousarray(src_rows, dtype=np.uint64)
        _check(self._lib.wdbx_index_compact(self._h, src.ctypes.data_as(_u64p), src.size))

    def fill_synthetic(self, seed: int, counter_row0: int, n: int, normalize: bool) -> int:
        first = C.c_uint64(0)
        _check(self._lib.wdbx_index_fill_synthetic(self._h, int(seed), int(counter_row0), int(n), int(normalize),
                                                   C.byref(first)))
        return first.value

    # -- search (host buffers, blocking) --
    def search(self, queries, k: int, normalize_queries: bool = False,
               mask_words: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """``mask_words``: optional uint32 bit mask over the stored rows (bit r%32 of word r//32 set =
        row r may be returned), see :func:`pack_row_mask`."""
        q = _as_f32(queries, self.dim)
        nq = q.shape[0]
        idx = np.empty((nq, int(k)), np.int64)
        score = np.empty((nq, int(k)), np.float32)
        if mask_words is None:
            _check(self._lib.wdbx_index_search(self._h, q.ctypes.data_as(_f32p), nq, int(k), int(normalize_queries),
                                               idx.ctypes.data_as(_i64p), score.ctypes.data_as(_f32p)))
        else:
            m = np.ascontiguousarray(mask_words, dtype=np.uint32)
            if m.size < (self.size() + 31) // 32:
                raise ValueError("mask has fewer bits than the index has rows")
            # (the library checks the length again under the handle's lock: rows may have been added since the line above)
            _check(self._lib.wdbx_index_search_masked_n(self._h, q.ctypes.data_as(_f32p), nq, int(k),
                                                        int(normalize_queries), m.ctypes.data_as(C.POINTER(C.c_uint32)), m.size,
                                                        idx.ctypes.data_as(_i64p), score.ctypes.data_as(_f32p)))
        return idx, score

    # -- device-resident path --
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def device_queries_synthetic(self, seed: int, counter_row0: int, n: int, normalize: bool) -> DeviceBuffer:
        buf = self.alloc(int(n) * self.pitch * 4)
        _check(self._lib.wdbx_device_fill_synthetic(self._h, buf.ptr, int(seed), int(counter_row0), int(n),
                                                    int(normalize)))
        return buf

    def device_queries(self, queries) -> DeviceBuffer:
        q = _as_f32(queries, self.dim)
        padded = np.zeros((q.shape[0], self.pitch), np.float32)
        padded[:, : self.dim] = q
        buf = self.alloc(padded.nbytes)
        buf.upload(padded)
        return buf

    def search_device(self, d_queries: DeviceBuffer, nq: int, k: int, d_idx: DeviceBuffer, d_score: DeviceBuffer,
                      query_offset: int = 0, sharded: bool = False) -> None:
        """Enqueue ``nq`` scans (asynchronous); ``query_offset`` = first query row in ``d_queries``."""
        fn = self._lib.wdbx_index_search_sharded_device if sharded else self._lib.wdbx_index_search_device
        qptr = d_queries.ptr + int(query_offset) * self.pitch * 4
        _check(fn(self._h, qptr, int(nq), int(k), d_idx.ptr, d_score.ptr))

    def synchronize(self) -> None:
        _check(self._lib.wdbx_index_synchronize(self._h))

    # -- batched queries on the MFMA path (extension) --
    def search_batch_device(self, d_queries: DeviceBuffer, nq: int, k: int, d_idx: DeviceBuffer,
                            d_score: DeviceBuffer, query_offset: int = 0, sharded: bool = False) -> None:
        qptr = d_queries.ptr + int(query_offset) * self.pitch * 4
        fn = self._lib.wdbx_index_search_sharded_batch_device if sharded else self._lib.wdbx_index_search_batch_device
        _check(fn(self._h, qptr, int(nq), int(k), d_idx.ptr, d_score.ptr))

    def batch_status(self, nq: int):
        counts = np.zeros(max(int(nq), 1), np.uint32)
        cap, over = C.c_uint32(0), C.c_int(0)
        _check(self._lib.wdbx_index_batch_status(self._h, counts.ctypes.data_as(C.POINTER(C.c_uint32)), int(nq),
                                                 C.byref(cap), C.byref(over)))
        return {"counts": counts[: int(nq)], "capacity": cap.value, "overflowed": over.value}

    def profile_read_gemm(self):
        n, ms = C.c_uint64(0), C.c_double(0)
        _check(self._lib.wdbx_index_profile_read_gemm(self._h, C.byref(n), C.byref(ms)))
        return {"gemm_launches": n.value, "gemm_ms": ms.value}

    def profile_read_sample(self):
        n, ms = C.c_uint64(0), C.c_double(0)
        _check(self._lib.wdbx_index_profile_read_sample(self._h, C.byref(n), C.byref(ms)))
        return {"sample_launches": n.value, "sample_ms": ms.value}

    # -- shard group (RCCL) --
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        _check(load_library().wdbx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, unique_id: bytes, global_row_base: int) -> None:
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        _check(self._lib.wdbx_index_comm_init(self._h, int(nranks), int(rank), buf, int(global_row_base)))

    def comm_destroy(self) -> None:
        _check(self._lib.wdbx_index_comm_destroy(self._h))

    def comm_info(self):
        """What RCCL reports for this handle's communicator: ranks, this rank, and the handle's global row base."""
        n, r, base = C.c_int(0), C.c_int(-1), C.c_uint64(0)
        _check(self._lib.wdbx_index_comm_info(self._h, C.byref(n), C.byref(r), C.byref(base)))
        return {"rccl_nranks": n.value, "rccl_rank": r.value, "row_base": base.value}

    def comm_set_row_base(self, global_row_base: int) -> None:
        _check(self._lib.wdbx_index_comm_set_row_base(self._h, int(global_row_base)))

    def comm_allgather_host(self, payload: bytes, nranks: int) -> list:
        """All-gather ``payload`` (same length on every rank) through the handle's RCCL communicator; returns the
        ranks' payloads in rank order.  Launcher-side plumbing only (barrier, max of a time, cross-checks)."""
        send = C.create_string_buffer(payload, len(payload))
        recv = C.create_string_buffer(len(payload) * int(nranks))
        _check(self._lib.wdbx_index_comm_allgather_host(self._h, send, recv, len(payload)))
        raw = recv.raw
        return [raw[i * len(payload):(i + 1) * len(payload)] for i in range(int(nranks))]

    # -- measurement / knobs --
    def profile(self, enable: bool) -> None:
        _check(self._lib.wdbx_index_profile(self._h, int(enable)))

    def profile_read(self):
        sl, ml = C.c_uint64(0), C.c_uint64(0)
        sm, mm = C.c_double(0), C.c_double(0)
        _check(self._lib.wdbx_index_profile_read(self._h, C.byref(sl), C.byref(sm), C.byref(ml), C.byref(mm)))
        return {"scan_launches": sl.value, "scan_ms": sm.value, "merge_launches": ml.value, "merge_ms": mm.value}

    def probe_read_ms(self, nontemporal: bool = True, blocks: int = 0, reps: int = 20) -> float:
        ms = C.c_double(0)
        _check(self._lib.wdbx_index_probe_read(self._h, int(nontemporal), int(blocks), int(reps), C.byref(ms)))
        return ms.value

    def set_option(self, name: str, value: int) -> None:
        _check(self._lib.wdbx_index_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        _check(self._lib.wdbx_index_get_option(self._h, name.encode(), C.byref(v)))
        return v.value


class NativeGroup:
    """S shards on S distinct devices in one process (``wdbx_group_*``): contiguous row ranges, RCCL
    all-gather of the per-shard key lists, merge on the first device."""

    def __init__(self, device_ids, dim: int, metric: int = METRIC_COSINE, cap_per_shard: int = 1 << 20):
        self._lib = load_library()
        self._h = None
        self._attached = ()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        _check(self._lib.wdbx_group_create(ids, len(device_ids), int(dim), int(metric), int(cap_per_shard), C.byref(h)))
        self._h = h.value
        self.dim = int(dim)

    EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_COPY = 0, 1, 2

    @classmethod
    def attach(cls, shards, exchange: int = 0) -> "NativeGroup":
        """A group over existing :class:`NativeIndex` shards (``wdbx_group_attach_ex``): the shards stay owned by their
        creators; merged results number the rows ``stride * shard + local_row`` (see :meth:`info`) unless
        :meth:`set_row_bases` says otherwise.  ``exchange``: 0 = RCCL when every shard has its own device (device copies
        otherwise), 1 = RCCL or fail, 2 = device copies."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = None
        self._attached = tuple(shards)  # keep the handles alive as long as the group
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        h = C.c_void_p()
        _check(self._lib.wdbx_group_attach_ex(arr, len(shards), int(exchange), C.byref(h)))
        self._h = h.value
        self.dim = shards[0].dim
        return self

    def set_row_bases(self, bases) -> None:
        arr = (C.c_uint64 * len(bases))(*[int(b) for b in bases])
        _check(self._lib.wdbx_group_set_row_bases(self._h, arr, len(bases)))

    # -- device-resident form: queries live on every shard's device, results on the first shard's --
    def queries_upload(self, queries, normalize_queries: bool = False) -> int:
        q = _as_f32(queries, self.dim)
        _check(self._lib.wdbx_group_queries_upload(self._h, q.ctypes.data_as(_f32p), q.shape[0], int(normalize_queries)))
        return q.shape[0]

    def queries_synthetic(self, seed: int, counter_row0: int, nq: int, normalize: bool) -> None:
        _check(self._lib.wdbx_group_queries_synthetic(self._h, int(seed), int(counter_row0), int(nq), int(normalize)))

    def search_resident(self, first_query: int, nq: int, k: int, k_out: Optional[int] = None) -> None:
        """Asynchronous: enqueue the search of resident queries ``[first_query, first_query + nq)`` on every shard."""
        _check(self._lib.wdbx_group_search_resident(self._h, int(first_query), int(nq), int(k), int(k_out or k)))

    def synchronize(self) -> None:
        _check(self._lib.wdbx_group_synchronize(self._h))

    def results(self, nq: int, k_out: int) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.empty((int(nq), int(k_out)), np.int64)
        score = np.empty((int(nq), int(k_out)), np.float32)
        _check(self._lib.wdbx_group_results(self._h, int(nq), int(k_out), idx.ctypes.data_as(_i64p),
                                            score.ctypes.data_as(_f32p)))
        return idx, score

    def info(self):
        n, r, stride = C.c_int(0), C.c_int(0), C.c_uint64(0)
        _check(self._lib.wdbx_group_info(self._h, C.byref(n), C.byref(r), C.byref(stride)))
        return {"shards": n.value, "rccl_nranks": r.value, "row_stride": stride.value}

    def stat(self, name: str) -> int:
        """A counter of the group: ``exchanges`` (exchange + merge steps enqueued so far), ``dispatches``, ``unusable``."""
        v = C.c_int64(0)
        _check(self._lib.wdbx_group_stat(self._h, name.encode(), C.byref(v)))
        return v.value

    def search_merged(self, queries, k: int, k_out: int, normalize_queries: bool = False,
                      mask_words=None) -> Tuple[np.ndarray, np.ndarray]:
        """Per-shard top-``k``, merged into the ``k_out`` best of their union (``k <= k_out <= shards * k``).
        ``mask_words``: optional list with one uint32 mask (see :func:`pack_row_mask`) or ``None`` per shard -- only rows whose
        bit is set compete (metadata filter pushed down into every shard's scan)."""
        q = _as_f32(queries, self.dim)
        idx = np.empty((q.shape[0], int(k_out)), np.int64)
        score = np.empty((q.shape[0], int(k_out)), np.float32)
        if mask_words is None:
            _check(self._lib.wdbx_group_search_merged(self._h, q.ctypes.data_as(_f32p), q.shape[0], int(k), int(k_out),
                                                      int(normalize_queries), idx.ctypes.data_as(_i64p),
                                                      score.ctypes.data_as(_f32p)))
        else:
            keep = [None if m is None else np.ascontiguousarray(m, dtype=np.uint32) for m in mask_words]
            if self._attached and len(keep) != len(self._attached):
                raise ValueError(f"{len(keep)} masks for {len(self._attached)} shards")
            for s, m in enumerate(keep):  # as NativeIndex.search does; the library checks again under the group's locks
                if m is not None and self._attached and m.size < (self._attached[s].size() + 31) // 32:
                    raise ValueError(f"shard {s}: mask has fewer bits than the shard has rows")
            u32p = C.POINTER(C.c_uint32)
            arr = (u32p * len(keep))(*[C.cast(None, u32p) if m is None else m.ctypes.data_as(u32p) for m in keep])
            counts = (C.c_uint64 * len(keep))(*[0 if m is None else m.size for m in keep])
            _check(self._lib.wdbx_group_search_merged_masked_n(self._h, q.ctypes.data_as(_f32p), q.shape[0], int(k), int(k_out),
                                                               int(normalize_queries), arr, counts, idx.ctypes.data_as(_i64p),
                                                               score.ctypes.data_as(_f32p)))
        return idx, score

    def close(self) -> None:
        if self._h:
            self._lib.wdbx_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, rows, normalize: bool = False) -> int:
        arr = _as_f32(rows, self.dim)
        first = C.c_uint64(0)
        _check(self._lib.wdbx_group_add(self._h, arr.ctypes.data_as(_f32p), arr.shape[0], int(normalize), C.byref(first)))
        return first.value

    def size(self) -> int:
        n = C.c_uint64(0)
        _check(self._lib.wdbx_group_size(self._h, C.byref(n)))
        return n.value

    def search(self, queries, k: int, normalize_queries: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        q = _as_f32(queries, self.dim)
        idx = np.empty((q.shape[0], int(k)), np.int64)
        score = np.empty((q.shape[0], int(k)), np.float32)
        _check(self._lib.wdbx_group_search(self._h, q.ctypes.data_as(_f32p), q.shape[0], int(k), int(normalize_queries),
                                           idx.ctypes.data_as(_i64p), score.ctypes.data_as(_f32p)))
        return idx, score
