"""The ``VectorIndex`` backend seam (reference: wdbx/core/indexing.py:18-217) and its
MI355X backend ``HipFlatIndex``.

``HipFlatIndex`` has the contract of the reference's exact backend
(``FaissIndex`` with ``FAISS_INDEX_TYPE="Flat"``, indexing.py:657-1183): rows are
unit-normalised at add time, the query is normalised the same way, scores are
inner products, results come best first as ``[(id, float)]``.  The scan itself
runs in HBM through the C ABI (``include/wdbx_hip.h``); this class only keeps the
string-id <-> row maps the reference keeps (indexing.py:697-700).
"""

from __future__ import annotations

import asyncio
import bisect
import json
import logging
import os
import threading
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import _native

logger = logging.getLogger(__name__)


class VectorIndex(ABC):
    """Backend contract, method for method the reference's ABC (indexing.py:18-217)."""

    @abstractmethod
    def __init__(self, vector_dim: int, index_path: Path, config: Any = None):
        ...

    @abstractmethod
    async def initialize(self):
        ...

    @abstractmethod
    async def shutdown(self):
        ...

    @abstractmethod
    def add(self, vector_id: str, vector: np.ndarray) -> bool:
        ...

    @abstractmethod
    async def add_async(self, vector_id: str, vector: np.ndarray) -> bool:
        ...

    @abstractmethod
    def batch_add(self, vectors: Dict[str, np.ndarray]) -> bool:
        ...

    @abstractmethod
    async def batch_add_async(self, vectors: Dict[str, np.ndarray]) -> bool:
        ...

    @abstractmethod
    def search(self, query_vector: np.ndarray, limit: int = 10) -> List[Tuple[str, float]]:
        ...

    @abstractmethod
    async def search_async(self, query_vector: np.ndarray, limit: int = 10) -> List[Tuple[str, float]]:
        ...

    @abstractmethod
    def remove(self, vector_id: str) -> bool:
        ...

    @abstractmethod
    async def remove_async(self, vector_id: str) -> bool:
        ...

    @abstractmethod
    def clear(self) -> bool:
        ...

    @abstractmethod
    async def clear_async(self) -> bool:
        ...

    @abstractmethod
    def optimize(self) -> bool:
        ...

    @abstractmethod
    async def optimize_async(self) -> bool:
        ...

    @abstractmethod
    def size(self) -> int:
        ...

    @abstractmethod
    def get_stats(self) -> Dict[str, Any]:
        ...


def normalize_vector(vector: np.ndarray) -> np.ndarray:
    """Unit-normalise exactly as the reference does (indexing.py:851-856): float32
    ``v / np.linalg.norm(v)``; a zero vector is returned unchanged."""
    # (np.linalg.norm of a 1-D real array IS sqrt(x.dot(x)) in the array's own precision -- numpy/linalg/_linalg.py; spelled
    # out here it skips ~2 us of argument handling per query.  Bit-exact against the reference: tests/golden/normalize.json)
    if vector.ndim == 1 and vector.dtype.kind == "f":
        norm = np.sqrt(vector.dot(vector))
    else:
        norm = np.linalg.norm(vector)
    if norm > 0:
        return vector / norm
    return vector


_METRICS = {"cosine": _native.METRIC_COSINE, "ip": _native.METRIC_COSINE, "l2": _native.METRIC_L2}


class HipFlatIndex(VectorIndex):
    """Exact flat index resident in one MI355X's HBM (one shard)."""

    _warned_no_cpu_path = False

    def __init__(self, vector_dim: int, index_path: Path, use_gpu: bool = False, config: Any = None,
                 device_id: int = 0):
        self.vector_dim = int(vector_dim)
        self.index_path = Path(index_path)
        # The reference's switch (wdbx.py:44,124 -> vector_store.py:43,128 -> indexing.py:741-748): True moves the flat
        # index to the GPU, False (its default) keeps faiss on the CPU, and a failed move is a warning.  This backend
        # IS that GPU branch and has no CPU path: the flag is recorded and reported as given, True is confirmed with the
        # reference's info line, False with one warning that the index is served from the GPU all the same.
        self.use_gpu = bool(use_gpu)
        self.config = config or {}
        self.device_id = int(device_id)
        metric_name = str(self.config.get("HIP_METRIC", "cosine")).lower()
        if metric_name not in _METRICS:
            raise ValueError(f"Unsupported HIP_METRIC: {metric_name}")
        self.metric_name = metric_name
        self.metric = _METRICS[metric_name]
        # backend errors in add / search: the reference logs them and returns False / [] (indexing.py:901-905,
        # :1028-1030); HIP_SWALLOW_ERRORS=False raises them instead.  A missing library or GPU always raises (below).
        self.swallow_errors = bool(self.config.get("HIP_SWALLOW_ERRORS", True))
        capacity = int(self.config.get("HIP_CAPACITY_ROWS", 4096) or 4096)
        self.autosave_rows = int(self.config.get("HIP_AUTOSAVE_ROWS", 1000) or 0)  # the reference's cadence (indexing.py:898)
        self.persist = bool(self.config.get("HIP_PERSIST_INDEX", True))  # False: scratch corpora (benchmarks) write no row files

        # same worker count as the reference's per-index pool (indexing.py:692)
        self.thread_pool = ThreadPoolExecutor(max_workers=4)

        # raises HipBackendError when the library or the GPU is missing: no fallback
        self._native = _native.NativeIndex(self.vector_dim, self.metric, self.device_id, capacity)
        if self.use_gpu:
            logger.info("HIP index using GPU acceleration (device %d)", self.device_id)
        elif not HipFlatIndex._warned_no_cpu_path:
            HipFlatIndex._warned_no_cpu_path = True
            logger.warning("use_gpu=False: the HIP backend has no CPU path; indices are served from the GPU (device %d)",
                           self.device_id)
        if not bool(self.config.get("HIP_BF16_SHADOW", True)):
            self._native.set_option("gemm_bf16", 1)  # bf16 selection tiles on the fp32 rows
        if not bool(self.config.get("HIP_U8_SHADOW", True)):
            self._native.set_option("scan_shadow", 1)  # single queries on the bf16 tile path (or fp32 scans without it)

        self.id_to_index: Dict[str, int] = {}
        self.index_to_id: Dict[int, str] = {}
        self.next_index = 0
        # ingest (row numbering + id maps) is serialised; the reference mutates these maps unguarded
        # from its 4-worker pool (indexing.py:381-383 under run_in_executor :407)
        self._ingest_lock = threading.RLock()
        # bulk-ingested row ranges with IMPLICIT ids "<prefix><label>" (no per-row Python objects):
        # (first_row, count, prefix, first_label); removed rows of such ranges are remembered
        self._implicit: List[Tuple[int, int, str, int]] = []
        self._implicit_removed: set = set()
        self._implicit_key = None  # (lookup tables of the list above are rebuilt when this no longer matches it)
        # persistence state: rows [0, _saved_rows) are in the rows file, _dirty_rows were overwritten since
        self._saved_rows = 0
        self._dirty_rows: set = set()
        self._unsaved_adds = 0
        self._rows_gen = 0        # generation of the committed row file (see _rows_path)
        self._rewrite = False     # the next save must write a whole new generation (rows moved or dropped under the mapping)
        self._load_index()

    # ---- persistence: flat [n, d] fp32 rows + id table ----
    # index.rows.npy is a .npy file with a FIXED-SIZE header (128 bytes, shape padded with spaces), so rows are
    # appended and the header is rewritten in place: a save costs the rows added since the last one, not the corpus
    # (the reference rewrites its whole index every 1000 adds, indexing.py:898-899, :805-838).  Rows overwritten since
    # (replace, remove) are rewritten at their offsets.  index.mapping.json goes through a temporary file + fsync +
    # os.replace and names the row count it describes: a crash between the two leaves extra rows behind the mapping's
    # count, which the loader ignores.
    # A save that MOVES or DROPS rows the committed mapping names (after optimize(), after clear(), a file in another
    # layout) never touches the committed row file: the rows go to a file of the NEXT GENERATION (index.rows.g<N>.npy), the
    # new mapping names that generation ("rows_gen"), and the one os.replace of the mapping commits both at once -- a crash
    # before it leaves the old pair intact, a crash after it the new pair; the other generation's file is swept by the
    # next load or save (rows shifted under the old mapping made ids resolve to the wrong vectors).
    _HEADER_BYTES = 128

    def _rows_path(self, gen: int) -> Path:
        return self.index_path.with_suffix(".rows.npy" if gen == 0 else f".rows.g{gen}.npy")

    def _files(self):
        return self._rows_path(self._rows_gen), self.index_path.with_suffix(".mapping.json")

    def _sweep_other_generations(self) -> None:
        """Row files no committed mapping names (an interrupted rewrite, or the generation a rewrite just replaced)."""
        keep = self._rows_path(self._rows_gen).name
        stem = self.index_path.with_suffix("").name
        for f in self.index_path.parent.glob(stem + ".rows*.npy*"):
            if f.name != keep:
                try:
                    f.unlink()
                except OSError:
                    pass

    def _npy_header(self, rows: int) -> bytes:
        body = "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d), }" % (rows, self.vector_dim)
        pad = self._HEADER_BYTES - 10 - len(body) - 1
        if pad < 0:
            raise ValueError("npy header does not fit its fixed size")
        text = body + " " * pad + "\n"
        return b"\x93NUMPY\x01\x00" + len(text).to_bytes(2, "little") + text.encode("latin1")

    def _load_index(self) -> None:
        map_file = self.index_path.with_suffix(".mapping.json")
        if not map_file.exists():
            return
        try:
            with open(map_file, "r") as f:
                mapping = json.load(f)
            self._rows_gen = int(mapping.get("rows_gen", 0))
            rows_file = self._rows_path(self._rows_gen)
            if not rows_file.exists():
                self._rows_gen = 0
                return
            rows = np.load(rows_file, mmap_mode="r")
            n = int(mapping["next_index"])
            if rows.ndim != 2 or rows.shape[1] != self.vector_dim or n > rows.shape[0]:
                raise ValueError("index files do not match this index")
            step = 1 << 18
            for r0 in range(0, n, step):  # stream: never hold the corpus twice
                self._native.add(np.asarray(rows[r0:min(r0 + step, n)], dtype=np.float32), normalize=False)
            self.id_to_index = {k: int(v) for k, v in mapping["id_to_index"].items()}
            self.index_to_id = {v: k for k, v in self.id_to_index.items()}
            self.next_index = n
            self._implicit = [tuple(x) for x in mapping.get("implicit", [])]
            self._implicit_removed = set(mapping.get("implicit_removed", []))
            self._implicit_key = None
            # a file in the plain np.save layout (another header size) is rewritten whole by the next save
            with open(rows_file, "rb") as f:
                fixed = f.read(10)[8:10] == (self._HEADER_BYTES - 10).to_bytes(2, "little")
            self._saved_rows = n if fixed else 0
            self._dirty_rows.clear()
            self._unsaved_adds = 0
            self._rewrite = False
            del rows
            self._sweep_other_generations()
        except Exception as e:
            logger.error("Error loading HIP index: %s", e)
            self._native.clear()
            self.id_to_index, self.index_to_id, self.next_index = {}, {}, 0
            self._implicit, self._implicit_removed, self._implicit_key = [], set(), None
            self._saved_rows = 0

    def _save_index(self) -> bool:
        """Bring the files up to date with the index (incremental; see above).  Serialised with ingest."""
        if not self.persist:
            return True
        with self._ingest_lock:
            rows_file, map_file = self._files()
            try:
                rows_file.parent.mkdir(parents=True, exist_ok=True)
                n, row_bytes, step = self.next_index, self.vector_dim * 4, 1 << 18
                fresh = self._rewrite or self._saved_rows == 0 or not rows_file.exists()
                # a whole rewrite (first save, rows moved or dropped since the committed mapping, a file in another layout)
                # goes to the NEXT generation's file: the file the committed mapping names is not touched, and the
                # os.replace of the mapping below commits the new rows and the new mapping together
                gen = self._rows_gen
                if fresh and (rows_file.exists() or map_file.exists()):
                    gen += 1
                target = self._rows_path(gen) if fresh else rows_file
                with open(target, "wb" if fresh else "r+b") as f:
                    if fresh:
                        f.write(self._npy_header(0))
                        first = 0
                    else:
                        first = self._saved_rows
                        for r in sorted(r for r in self._dirty_rows if r < first):  # overwritten rows, in place
                            f.seek(self._HEADER_BYTES + r * row_bytes)
                            f.write(self._native.get_rows(r, 1).tobytes())
                    f.seek(self._HEADER_BYTES + first * row_bytes)
                    for r0 in range(first, n, step):
                        f.write(self._native.get_rows(r0, min(step, n - r0)).tobytes())
                    f.truncate(self._HEADER_BYTES + n * row_bytes)
                    f.flush()
                    os.fsync(f.fileno())
                    f.seek(0)
                    f.write(self._npy_header(n))  # the rows are durable before the header names them
                    f.flush()
                    os.fsync(f.fileno())
                tmp = map_file.with_suffix(".json.tmp")
                with open(tmp, "w") as f:
                    json.dump({"id_to_index": self.id_to_index, "next_index": n, "rows_gen": gen,
                               "implicit": self._implicit, "implicit_removed": sorted(self._implicit_removed)}, f)
                    f.flush()
                    os.fsync(f.fileno())
                os.replace(tmp, map_file)  # THE commit point: the mapping and the generation of rows it names
                try:  # (make the rename itself durable)
                    dfd = os.open(str(map_file.parent), os.O_RDONLY)
                    try:
                        os.fsync(dfd)
                    finally:
                        os.close(dfd)
                except OSError:
                    pass
                self._rows_gen, self._rewrite = gen, False
                if fresh:
                    self._sweep_other_generations()
                self._saved_rows, self._unsaved_adds = n, 0
                self._dirty_rows.clear()
                return True
            except Exception as e:
                logger.error("Error saving HIP index: %s", e)
                return False

    def _note_added(self, count: int) -> None:
        """Autosave cadence of the per-vector ingest paths (the reference saves every 1000 adds, indexing.py:898,
        :961); bulk ``add_rows`` is saved by ``save()`` / shutdown only."""
        self._unsaved_adds += count
        if self.autosave_rows > 0 and self._unsaved_adds >= self.autosave_rows:
            self._save_index()

    def save(self) -> bool:
        return self._save_index()

    def unsaved(self) -> bool:
        return self.persist and (self._saved_rows != self.next_index or bool(self._dirty_rows) or self._rewrite)

    async def initialize(self):
        pass

    async def shutdown(self):
        loop = asyncio.get_event_loop()
        await loop.run_in_executor(self.thread_pool, self._save_index)
        self.thread_pool.shutdown()
        self._native.close()

    # ---- id table: explicit dicts (reference style) + implicit ranges (bulk ingest) ----
    # The implicit ranges are looked up by bisection (a compaction can split one bulk range into tens of thousands of
    # runs): by first row for row -> id, and per prefix by first label for id -> row.  Tables rebuilt when the list changes.
    def _implicit_tables(self):
        key = (id(self._implicit), len(self._implicit))
        if self._implicit_key != key:
            runs = self._implicit
            order = sorted(range(len(runs)), key=lambda i: runs[i][0])
            self._runs_by_row = ([runs[i][0] for i in order], order)
            by_prefix: Dict[str, Tuple[List[int], List[int]]] = {}
            for i in sorted(range(len(runs)), key=lambda i: (runs[i][2], runs[i][3])):
                labels, idxs = by_prefix.setdefault(runs[i][2], ([], []))
                labels.append(runs[i][3])
                idxs.append(i)
            self._runs_by_prefix = by_prefix
            self._implicit_key = key
        return self._runs_by_row, self._runs_by_prefix

    def _id_of(self, row: int) -> str:
        vid = self.index_to_id.get(row)
        if vid is not None:
            return vid
        if self._implicit and row not in self._implicit_removed:
            (firsts, order), _ = self._implicit_tables()
            pos = bisect.bisect_right(firsts, row) - 1
            if pos >= 0:
                first, count, prefix, label0 = self._implicit[order[pos]]
                if row < first + count:
                    return f"{prefix}{label0 + row - first}"
        return str(row)  # unmapped row: the reference's fallback (indexing.py:1021)

    def _row_of(self, vector_id: str) -> Optional[int]:
        row = self.id_to_index.get(vector_id)
        if row is not None:
            return row
        if not self._implicit:
            return None
        _, by_prefix = self._implicit_tables()
        for prefix, (labels, idxs) in by_prefix.items():
            if vector_id.startswith(prefix):
                tail = vector_id[len(prefix):]
                if tail.isdigit() and str(int(tail)) == tail:
                    pos = bisect.bisect_right(labels, int(tail)) - 1
                    if pos >= 0:
                        first, count, _, label0 = self._implicit[idxs[pos]]
                        if int(tail) < label0 + count:
                            row = first + int(tail) - label0
                            return None if row in self._implicit_removed else row
        return None

    def mapped_rows(self):
        """Iterate (row, id) over every row that still has an id."""
        yield from self.index_to_id.items()
        for first, count, prefix, label0 in self._implicit:
            for i in range(count):
                if first + i not in self._implicit_removed:
                    yield first + i, f"{prefix}{label0 + i}"

    # ---- ingest ----
    def _prepare(self, vector: np.ndarray) -> np.ndarray:
        v = np.asarray(vector)
        if v.dtype != np.float32:
            v = v.astype(np.float32)
        if v.shape != (self.vector_dim,):
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {v.shape}")
        return normalize_vector(v) if self.metric == _native.METRIC_COSINE else v

    def _add_unlocked(self, vector_id: str, vector: np.ndarray) -> bool:
        """Append one row (indexing.py:858-905).  An id that is already stored is
        overwritten in place, as the reference's default backend does
        (``replace_vector``, indexing.py:370-375)."""
        try:
            row = self._prepare(vector)
            existing = self._row_of(vector_id)
            if existing is not None:
                self._native.set_rows(existing, row)
                self._dirty_rows.add(existing)
                return True
            first = self._native.add(row)
            assert first == self.next_index
            self.id_to_index[vector_id] = self.next_index
            self.index_to_id[self.next_index] = vector_id
            self.next_index += 1
            logger.debug("Added vector %s to HIP index", vector_id)
            self._note_added(1)
            return True
        except Exception as e:
            logger.error("Error adding vector to HIP index: %s", e)
            if not self.swallow_errors:
                raise
            return False

    async def add_async(self, vector_id: str, vector: np.ndarray) -> bool:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.thread_pool, self.add, vector_id, vector)

    def _batch_add_unlocked(self, vectors: Dict[str, np.ndarray]) -> bool:
        """Append many rows with one upload (indexing.py:921-968)."""
        if not vectors:
            return True
        try:
            fresh_ids, fresh_rows = [], []
            for vector_id, vector in vectors.items():
                row = self._prepare(vector)
                existing = self._row_of(vector_id)
                if existing is not None:
                    self._native.set_rows(existing, row)
                    self._dirty_rows.add(existing)
                elif vector_id in fresh_ids:
                    fresh_rows[fresh_ids.index(vector_id)] = row
                else:
                    fresh_ids.append(vector_id)
                    fresh_rows.append(row)
            if fresh_rows:
                first = self._native.add(np.stack(fresh_rows))
                assert first == self.next_index
                for i, vector_id in enumerate(fresh_ids):
                    self.id_to_index[vector_id] = self.next_index + i
                    self.index_to_id[self.next_index + i] = vector_id
                self.next_index += len(fresh_ids)
                self._note_added(len(fresh_ids))
            logger.debug("Batch added %d vectors to HIP index", len(vectors))
            return True
        except Exception as e:
            logger.error("Error batch adding vectors to HIP index: %s", e)
            if not self.swallow_errors:
                raise
            return False

    async def batch_add_async(self, vectors: Dict[str, np.ndarray]) -> bool:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.thread_pool, self.batch_add, vectors)

    def _add_rows_unlocked(self, vector_ids: Optional[List[str]], rows: np.ndarray, id_prefix: str = "row_",
                 first_label: Optional[int] = None, exact_normalize: bool = False) -> Tuple[int, int]:
        """Bulk ingest of a contiguous ``[n, d]`` array: ONE host-to-HBM copy
        (measured 56 GB/s) and the device row-normalise kernel instead of n Python calls.
        ``vector_ids=None`` gives the rows implicit ids ``f"{id_prefix}{first_label + i}"`` with no
        per-row Python objects (the reference's dict maps cannot hold 10 M+ rows).
        ``exact_normalize=True`` normalises on the host exactly as the reference does (bit-exact
        rows, slow); the device kernel differs by at most 1 ulp per element.
        Returns (first_row, n)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.vector_dim:
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {rows.shape}")
        n = rows.shape[0]
        if vector_ids is not None and len(vector_ids) != n:
            raise ValueError("len(vector_ids) != number of rows")
        cosine = self.metric == _native.METRIC_COSINE
        if cosine and exact_normalize and n:
            rows = np.stack([normalize_vector(r) for r in rows])
        first = self._native.add(rows, normalize=cosine and not exact_normalize)
        assert first == self.next_index
        if vector_ids is None:
            if n:
                self._implicit.append((first, n, id_prefix, first if first_label is None else int(first_label)))
        else:
            for i, vector_id in enumerate(vector_ids):
                self.id_to_index[vector_id] = first + i
                self.index_to_id[first + i] = vector_id
        self.next_index = first + n
        return first, n

    # public ingest entry points: one at a time per index
    def add(self, vector_id: str, vector: np.ndarray) -> bool:
        with self._ingest_lock:
            return self._add_unlocked(vector_id, vector)

    def batch_add(self, vectors: Dict[str, np.ndarray]) -> bool:
        with self._ingest_lock:
            return self._batch_add_unlocked(vectors)

    def add_rows(self, vector_ids: Optional[List[str]], rows: np.ndarray, id_prefix: str = "row_",
                 first_label: Optional[int] = None, exact_normalize: bool = False) -> Tuple[int, int]:
        with self._ingest_lock:
            return self._add_rows_unlocked(vector_ids, rows, id_prefix, first_label, exact_normalize)

    def add_synthetic_rows(self, seed: int, counter_row0: int, n: int, id_prefix: str = "row_",
                           first_label: Optional[int] = None) -> Tuple[int, int]:
        """Append ``n`` rows of the counter-based benchmark corpus, generated and -- for cosine
        -- normalised on the device (``wdbx_index_fill_synthetic``); implicit ids as in ``add_rows``."""
        with self._ingest_lock:
            first = self._native.fill_synthetic(seed, counter_row0, n, normalize=self.metric == _native.METRIC_COSINE)
            assert first == self.next_index
            if n:
                self._implicit.append((first, n, id_prefix, first if first_label is None else int(first_label)))
            self.next_index = first + n
            return first, n

    def remove(self, vector_id: str) -> bool:
        with self._ingest_lock:
            return self._remove_unlocked(vector_id)

    def clear(self) -> bool:
        with self._ingest_lock:
            return self._clear_unlocked()

    # ---- search ----
    def _map(self, idx_row: np.ndarray, score_row: np.ndarray) -> List[Tuple[str, float]]:
        """(row, score) slots of one result -> [(id, similarity)] as the reference builds it (indexing.py:1020-1024):
        unused slots (-1) dropped, unmapped rows under their ``str(row)`` fallback.  The lookup tables are fetched once per
        result, not once per id (this runs under the GIL for every query of every thread)."""
        rows, scores = idx_row.tolist(), score_row.tolist()
        cosine = self.metric == _native.METRIC_COSINE
        explicit = self.index_to_id
        implicit = self._implicit
        if implicit:
            (firsts, order), _ = self._implicit_tables()
            removed = self._implicit_removed
        out = []
        for row, s in zip(rows, scores):
            if row == -1:  # unused slot (indexing.py:1023)
                continue
            vid = explicit.get(row) if explicit else None
            if vid is None:
                vid = None
                if implicit and not (removed and row in removed):
                    pos = bisect.bisect_right(firsts, row) - 1
                    if pos >= 0:
                        first, count, prefix, label0 = implicit[order[pos]]
                        if row < first + count:
                            vid = f"{prefix}{label0 + row - first}"
                if vid is None:
                    vid = str(row)  # unmapped row: the reference's fallback (indexing.py:1021)
            out.append((vid, s if cosine else -s))
        return out

    def search(self, query_vector: np.ndarray, limit: int = 10,
               row_mask: Optional[np.ndarray] = None) -> List[Tuple[str, float]]:
        """Exact top-``limit`` of this shard, best first (indexing.py:983-1030).
        ``row_mask`` (extension): bool per stored row; only allowed rows compete."""
        try:
            if self.next_index == 0:
                return []
            actual_limit = min(int(limit), self.next_index, _native.MAX_K)
            if actual_limit <= 0:
                return []
            q = self._prepare(query_vector)
            words = None
            if row_mask is not None:
                words = row_mask if row_mask.dtype == np.uint32 else _native.pack_row_mask(row_mask)
            idx, score = self._native.search(q, actual_limit, mask_words=words)
            return self._map(idx[0], score[0])
        except Exception as e:
            logger.error("Error searching HIP index: %s", e)
            if self.swallow_errors:
                return []  # reference convention (indexing.py:1028-1030)
            raise

    def row_mask_for(self, predicate) -> np.ndarray:
        """uint32 mask words of the rows whose id satisfies ``predicate(id)``; unmapped (removed)
        rows are excluded."""
        allowed = np.zeros(self.next_index, dtype=bool)
        for row, vector_id in self.mapped_rows():
            if predicate(vector_id):
                allowed[row] = True
        return _native.pack_row_mask(allowed)

    def search_batch(self, queries: np.ndarray, limit: int = 10) -> List[List[Tuple[str, float]]]:
        """Extension: many queries in one call."""
        raw = self.search_batch_raw(queries, limit)
        if raw is None:
            return [[] for _ in range(len(queries))]
        return [self._map(i, s) for i, s in zip(*raw)]

    def search_batch_raw(self, queries: np.ndarray, limit: int = 10):
        """``search_batch`` without the id mapping: (rows int64[nq, k], scores f32[nq, k]) or None for "no results" (empty
        index, swallowed backend error).  The coalescing front of ``VectorStore`` hands each waiting caller ITS row of
        these, and the caller maps ids and merges on its own thread while the next batch is already on the GPU."""
        queries = np.asarray(queries, dtype=np.float32)
        if queries.ndim != 2 or queries.shape[1] != self.vector_dim:
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {queries.shape}")
        if self.next_index == 0 or queries.shape[0] == 0:
            return None
        actual_limit = min(int(limit), self.next_index, _native.MAX_K)
        if actual_limit <= 0:
            return None
        try:
            q = np.stack([self._prepare(r) for r in queries])
            return self._native.search(q, actual_limit)
        except Exception as e:
            logger.error("Error searching HIP index: %s", e)
            if self.swallow_errors:
                return None
            raise

    async def search_async(self, query_vector: np.ndarray, limit: int = 10) -> List[Tuple[str, float]]:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.thread_pool, self.search, query_vector, limit)

    # ---- removal / maintenance ----
    def _remove_unlocked(self, vector_id: str) -> bool:
        """Unmap the id and overwrite the row so it "will never match anything" (indexing.py:538-560); the row number
        is not reused.  The reference writes zeros, which still score 0 (cosine) or |q|^2 (L2) and can come back under
        the ``str(row)`` fallback id; here the row becomes NaN: a NaN score is never a result on any path
        (include/wdbx_hip.h), and the selection scans skip such rows outright, so a removed row is gone for good."""
        row = self._row_of(vector_id)
        if row is None:
            return False
        explicit = vector_id in self.id_to_index
        if explicit:
            self.id_to_index.pop(vector_id)
            self.index_to_id.pop(row, None)
        else:
            self._implicit_removed.add(row)
        try:
            self._native.set_rows(row, np.full(self.vector_dim, np.nan, np.float32))
            self._dirty_rows.add(row)
            return True
        except Exception as e:
            logger.error("Error removing vector from HIP index: %s", e)
            if explicit:
                self.id_to_index[vector_id] = row
                self.index_to_id[row] = vector_id
            else:
                self._implicit_removed.discard(row)
            if not self.swallow_errors:
                raise
            return False

    async def remove_async(self, vector_id: str) -> bool:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.thread_pool, self.remove, vector_id)

    def _clear_unlocked(self) -> bool:
        try:
            self._native.clear()
            self.id_to_index, self.index_to_id, self.next_index = {}, {}, 0
            self._implicit, self._implicit_removed, self._implicit_key = [], set(), None
            self._saved_rows, self._unsaved_adds = 0, 0
            self._dirty_rows.clear()
            self._rewrite = True
            self._save_index()
            return True
        except Exception as e:
            logger.error("Error clearing HIP index: %s", e)
            if not self.swallow_errors:
                raise
            return False

    async def clear_async(self) -> bool:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.thread_pool, self.clear)

    def optimize(self, min_dead_fraction: Optional[float] = None) -> bool:
        """The reference's rebuild hook (indexing.py:1124-1149 re-trains / rebuilds its faiss index).  A flat scan has
        nothing to re-train, but it streams every stored row: removed rows stay behind as NaN tombstones that every scan
        reads and can never return.  ``optimize`` compacts them away ON THE DEVICE (``wdbx_index_compact``: surviving rows
        move down in row order, the derived copies are rebuilt for the moved tail only), renumbers the id maps and the
        implicit ranges, and lets the next save rewrite the row file from the first moved row.  Row order -- hence tie
        order -- is preserved.  Compacts when the dead share of the stored rows reaches ``min_dead_fraction`` (default:
        config ``HIP_COMPACT_MIN_FRACTION``, 0.0 = whenever there is a dead row)."""
        with self._ingest_lock:
            try:
                n = self.next_index
                dead = n - self.size()
                if min_dead_fraction is None:
                    min_dead_fraction = float(self.config.get("HIP_COMPACT_MIN_FRACTION", 0.0) or 0.0)
                if dead <= 0 or dead < min_dead_fraction * n:
                    return True
                live = np.zeros(n, dtype=bool)
                if self.index_to_id:
                    live[np.fromiter(self.index_to_id.keys(), dtype=np.int64, count=len(self.index_to_id))] = True
                for first, count, _, _ in self._implicit:
                    live[first:first + count] = True
                if self._implicit_removed:
                    live[np.fromiter(self._implicit_removed, dtype=np.int64, count=len(self._implicit_removed))] = False
                src = np.flatnonzero(live).astype(np.uint64)
                new_of_old = np.cumsum(live) - 1
                moved = np.flatnonzero(src != np.arange(src.size, dtype=np.uint64))
                first_moved = int(moved[0]) if moved.size else int(src.size)
                self._native.compact(src)
                self.id_to_index = {vid: int(new_of_old[row]) for vid, row in self.id_to_index.items()}
                self.index_to_id = {row: vid for vid, row in self.id_to_index.items()}
                runs: List[Tuple[int, int, str, int]] = []
                for first, count, prefix, label0 in self._implicit:
                    alive = live[first:first + count]
                    edges = np.flatnonzero(np.diff(np.concatenate(([False], alive, [False])).astype(np.int8)))
                    for b, e in zip(edges[0::2].tolist(), edges[1::2].tolist()):  # maximal runs of surviving rows
                        runs.append((int(new_of_old[first + b]), e - b, prefix, label0 + b))
                self._implicit = runs
                self._implicit_removed = set()
                self._implicit_key = None
                self.next_index = int(src.size)
                # The row file: the committed mapping names rows that have now moved or gone, so the next save must not
                # write into that file (a crash half way would leave the OLD mapping over SHIFTED rows: ids silently resolving
                # to other vectors, or a file shorter than the mapping says and the loader wiping the index).  It writes a
                # whole new generation instead and commits it with the mapping's rename (_save_index).
                self._dirty_rows = set()
                self._saved_rows = min(self._saved_rows, first_moved)
                self._rewrite = True
                logger.info("HIP index compacted: %d dead rows dropped, %d rows stored", dead, self.next_index)
                return True
            except Exception as e:
                logger.error("Error optimizing HIP index: %s", e)
                if not self.swallow_errors:
                    raise
                return False

    async def optimize_async(self) -> bool:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.thread_pool, self.optimize)

    def size(self) -> int:
        return len(self.id_to_index) + sum(c for _, c, _, _ in self._implicit) - len(self._implicit_removed)

    def get_stats(self) -> Dict[str, Any]:
        return {
            "type": "hip_flat",
            "size": self.size(),
            "dimension": self.vector_dim,
            "gpu_enabled": self.use_gpu,  # the flag as given (reference: indexing.py:1170-1183); "device" says where it runs
            "device": self.device_id,
            "metric": self.metric_name,
            "stored_rows": self.next_index,
            "capacity_rows": self._native.capacity(),
        }
