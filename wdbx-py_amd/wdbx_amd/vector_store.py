"""Shard fan-out, merge, threshold, metadata post-filter (reference:
wdbx/core/vector_store.py:22-815).  Same constructor, methods and result shapes;
every shard is a ``HipFlatIndex`` living in one GPU's HBM.

Differences a maintainer should know (all documented in INTEGRATION.md):
* the object is callable -- ``wdbx.vector_store(vector, metadata)`` works although
  ``wdbx.vector_store`` is this object (the reference's method/attribute clash,
  wdbx.py:120 vs :241, makes that call raise ``TypeError``);
* shard placement is a deterministic FNV-1a hash of the id instead of Python's
  salted ``hash`` (vector_store.py:178-190), so it survives a restart;
* ``index_type`` is "hip" only;
* the fan-out of ``search`` over several shards can be ONE call into the library: every shard's launches are enqueued by
  its own host thread, the per-shard (row, score) lists are exchanged (RCCL all-gather over xGMI with one shard per GPU,
  device copies when shards share a GPU) and merged on the device (``wdbx_group_search_merged``) -- the reference's
  loop + ``list.sort`` (vector_store.py:323-345) with the same candidate set and the same order (``HIP_GROUP_SEARCH``).
  Row masks of a pushed-down filter travel with the call (one mask per shard).
"""

from __future__ import annotations

import array
import asyncio
import json
import logging
import os
import pickle
import threading
import time
import uuid
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .config import WDBXConfig
from .indexing import HipFlatIndex

logger = logging.getLogger(__name__)

Result = Tuple[str, float, Dict[str, Any]]


def fnv1a_64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def matches_filter(metadata: Dict[str, Any], filter_metadata: Dict[str, Any]) -> bool:
    """Metadata predicate with the reference's grammar (vector_store.py:414-463):
    equality, or a ``{"$op": arg}`` dict of which only the first key counts;
    ``$gt $lt $gte $lte $in`` fail on a missing key, ``$nin`` and
    ``$exists: False`` pass on it, unknown operators are ignored."""
    for key, wanted in filter_metadata.items():
        if isinstance(wanted, dict) and list(wanted.keys())[0].startswith("$"):
            op = list(wanted.keys())[0]
            arg = wanted[op]
            have = key in metadata
            val = metadata.get(key)
            if op == "$exists":
                ok = have if arg else not have
            elif op == "$nin":
                ok = not (have and val in arg)
            elif op == "$in":
                ok = have and val in arg
            elif op == "$gt":
                ok = have and not (val <= arg)
            elif op == "$lt":
                ok = have and not (val >= arg)
            elif op == "$gte":
                ok = have and not (val < arg)
            elif op == "$lte":
                ok = have and not (val > arg)
            else:
                ok = True
            if not ok:
                return False
        elif key not in metadata or metadata[key] != wanted:
            return False
    return True


def _as_query(query_vector) -> np.ndarray:
    """list -> float32 array (vector_store.py:321 ``np.array(query_vector, dtype=np.float32)``): for a plain list of Python
    numbers ``array('f')`` makes the same double -> float casts in two thirds of the time; anything else goes the numpy way."""
    if type(query_vector) is list:
        try:
            return np.frombuffer(array.array("f", query_vector), dtype=np.float32)
        except (TypeError, OverflowError):
            pass
    return np.array(query_vector, dtype=np.float32)


class _SyncRequest:
    """One synchronous caller waiting in ``VectorStore._search_coalesced``.  ``gate`` is a plain lock used as a one-shot
    signal (created held; the server releases it): a tenth of the cost of ``threading.Event`` under the GIL."""
    __slots__ = ("query", "limit", "threshold", "flt", "gate", "result", "error", "promoted")

    def __init__(self, query, limit, threshold, flt):
        self.query, self.limit, self.threshold, self.flt = query, limit, threshold, flt
        self.gate = threading.Lock()
        self.gate.acquire()
        self.result = None
        self.error = None
        self.promoted = False


class VectorStore:
    _sync_coalesce = False  # (instances switch it on in __init__ from SYNC_COALESCE, once the queue and its lock exist)

    def __init__(
        self,
        vector_dim: int,
        data_dir: Path,
        num_shards: int = 1,
        use_gpu: bool = False,
        index_type: str = "hip",
        config: Optional[WDBXConfig] = None,
    ):
        self.vector_dim = vector_dim
        self.data_dir = Path(data_dir)
        self.num_shards = num_shards
        self.use_gpu = use_gpu
        self.index_type = index_type
        self.config = config or WDBXConfig({})

        self.vectors: Dict[str, np.ndarray] = {}
        self.metadata: Dict[str, Dict[str, Any]] = {}
        self.indices: List[HipFlatIndex] = []
        self._mask_cache: Dict[str, Any] = {}
        self._meta_version = 0
        self._pending: List[Any] = []      # coalescing queue of search_async (one event loop)
        self._drain_task = None
        # coalescing of SYNCHRONOUS callers on several threads (the reference's per-index pools call search from 4 workers,
        # indexing.py:692, :1045-1048): whoever arrives while a search is in flight queues up, and the next leader answers the
        # whole queue with ONE batched pass per shard (see _search_coalesced)
        self._sync_lock = threading.Lock()
        self._sync_pending: List[Any] = []
        self._sync_busy = False
        self._sync_last_batch = 0
        self._group_verified = False       # an RCCL group is checked once against the per-shard path before it is trusted
        self._sync_coalesce = bool(self.config.get("SYNC_COALESCE", True))
        # bulk-ingested rows: (prefix, first_label, count, shard) ranges with implicit ids, and the
        # shard of explicitly named bulk rows (placed by row range, not by hash)
        self._bulk_ranges: List[Tuple[str, int, int, int]] = []
        self._bulk_id_shard: Dict[str, int] = {}
        self._bulk_rows = 0        # labels handed out to implicit-id bulk rows so far (the next label)
        # shard group (one library call per search, RCCL merge): created lazily, None = not tried, False = unavailable
        self._group: Any = None
        self._group_lock = threading.Lock()  # (the group is built lazily by whichever searching thread comes first)
        self.last_search_path = ""  # "rccl_group" / "copy_group" / "threads": which fan-out served the last search (diagnostics)
        self._group_path = "rccl_group"

        self.thread_pool = ThreadPoolExecutor(
            max_workers=self.config.get("VECTOR_STORE_THREADS", os.cpu_count() or 4))
        # the per-shard calls of one fan-out run on a pool of their own: a fan-out that is itself running on
        # ``thread_pool`` (search_async) must never wait for workers of the pool it occupies (with
        # VECTOR_STORE_THREADS=1 and two shards that is a deadlock on the first call)
        self._shard_pool = ThreadPoolExecutor(max_workers=max(1, num_shards), thread_name_prefix="wdbx-shard")
        self._create_dirs()
        self._init_indices()
        self._load_data()
        logger.info("VectorStore initialized with %d vectors, %d shards, index_type=%s", self.count(),
                    self.num_shards, self.index_type)

    # ---- construction ----
    def _create_dirs(self) -> None:
        for sub in ["vectors", "metadata", "indices"] + [f"shard_{s}" for s in range(self.num_shards)]:
            (self.data_dir / sub).mkdir(parents=True, exist_ok=True)

    def _devices(self) -> List[int]:
        wanted = self.config.get("HIP_DEVICES")
        if wanted:
            return [int(d) for d in wanted]
        n = _native.device_count()
        if n < 1:
            raise _native.HipBackendError(-4, "no HIP device visible: the WDBX HIP backend needs an AMD GPU")
        return list(range(n))

    def _init_indices(self) -> None:
        if self.index_type not in ("hip", "hip_flat"):
            raise ValueError(f"Unsupported index type: {self.index_type}")
        devices = self._devices()
        self.indices = []
        for shard in range(self.num_shards):
            self.indices.append(HipFlatIndex(
                vector_dim=self.vector_dim,
                index_path=self.data_dir / f"shard_{shard}" / "index",
                use_gpu=self.use_gpu,
                config=self.config,
                device_id=devices[shard % len(devices)],
            ))

    def _load_data(self) -> None:
        meta_path = self.data_dir / "metadata" / "metadata.json"
        if meta_path.exists():
            try:
                with open(meta_path, "r") as f:
                    self.metadata = json.load(f)
            except Exception as e:
                logger.error("Error loading metadata: %s", e)
        bulk_path = self.data_dir / "metadata" / "bulk.json"
        if bulk_path.exists():
            try:
                with open(bulk_path, "r") as f:
                    b = json.load(f)
                self._bulk_ranges = [tuple(r) for r in b.get("ranges", [])]
                self._bulk_id_shard = {k: int(v) for k, v in b.get("id_shard", {}).items()}
                self._bulk_rows = int(b.get("rows", 0))
            except Exception as e:
                logger.error("Error loading bulk table: %s", e)
        vec_path = self.data_dir / "vectors" / "vectors.pickle"
        if vec_path.exists():
            try:
                with open(vec_path, "rb") as f:
                    self.vectors = pickle.load(f)
            except Exception as e:
                logger.error("Error loading vectors: %s", e)
        self._reconcile()

    def _reconcile(self) -> None:
        """Make the tables loaded above agree with what the shards' own files brought back into HBM (they are saved
        at different moments; after an unclean exit the index files can be older than vectors.pickle / bulk.json):
        every id of the id -> vector table that its shard does not hold is added again, and bulk ranges / bulk ids
        that no shard holds any more are dropped (their rows existed only in HBM)."""
        missing: Dict[int, Dict[str, np.ndarray]] = {}
        for vid, vec in self.vectors.items():
            shard = self._get_shard_for_id(vid)
            if self.indices[shard]._row_of(vid) is None:
                missing.setdefault(shard, {})[vid] = vec
        for shard, vecs in missing.items():
            logger.warning("shard %d: re-adding %d vectors its index files did not hold", shard, len(vecs))
            self.indices[shard].batch_add(vecs)
        # a bulk range survives if its shard still holds at least one run of its labels (a compaction splits a range
        # into the runs of its surviving rows, indexing.py ``optimize``)
        def held(prefix, label0, count, shard):
            return any(p == prefix and l0 < label0 + count and label0 < l0 + c for _, c, p, l0 in self.indices[shard]._implicit)

        kept = [r for r in self._bulk_ranges if held(*r)]
        if len(kept) != len(self._bulk_ranges):
            logger.warning("dropping %d bulk ranges that no shard holds any more", len(self._bulk_ranges) - len(kept))
            self._bulk_ranges = kept
        gone = [vid for vid, s in self._bulk_id_shard.items() if self.indices[s]._row_of(vid) is None]
        for vid in gone:
            del self._bulk_id_shard[vid]
        if gone:
            logger.warning("dropping %d bulk ids that no shard holds any more", len(gone))

    def _save_metadata(self) -> None:
        try:
            with open(self.data_dir / "metadata" / "metadata.json", "w") as f:
                json.dump(self.metadata, f)
            with open(self.data_dir / "metadata" / "bulk.json", "w") as f:
                json.dump({"ranges": self._bulk_ranges, "id_shard": self._bulk_id_shard, "rows": self._bulk_rows}, f)
        except Exception as e:
            logger.error("Error saving metadata: %s", e)

    def _save_vectors(self) -> None:
        try:
            with open(self.data_dir / "vectors" / "vectors.pickle", "wb") as f:
                pickle.dump(self.vectors, f)
        except Exception as e:
            logger.error("Error saving vectors: %s", e)

    def _save_indices(self) -> None:
        for ix in self.indices:
            if ix.unsaved():
                ix.save()

    def _save_now(self) -> None:
        """Everything durable at once (VECTOR_STORE_SAVE_IMMEDIATELY, shutdown, clear): the shards' rows first -- an
        incremental append, indexing.py -- then the tables that refer to them."""
        self._save_indices()
        self._save_metadata()
        self._save_vectors()

    def _get_shard_for_id(self, vector_id: str) -> int:
        shard = self._bulk_id_shard.get(vector_id)
        if shard is not None:
            return shard
        for prefix, label0, count, s in self._bulk_ranges:
            if vector_id.startswith(prefix):
                tail = vector_id[len(prefix):]
                if tail.isdigit() and label0 <= int(tail) < label0 + count:
                    return s
        return fnv1a_64(vector_id) % self.num_shards

    def _is_bulk(self, vector_id: str) -> bool:
        return vector_id not in self.vectors and self.indices[self._get_shard_for_id(vector_id)]._row_of(
            vector_id) is not None

    def bulk_store(self, rows, ids: Optional[Sequence[str]] = None,
                   metadata: Optional[Dict[str, Dict[str, Any]]] = None, id_prefix: str = "row_",
                   exact_normalize: bool = False, persist: Optional[bool] = None) -> int:
        """Bulk ingest of an ``[N, d]`` float32 array (the reference's per-vector
        ``batch_store`` builds N Python arrays and dict entries, vector_store.py:720-763).
        Rows are placed in CONTIGUOUS ranges (row r -> shard r // ceil(N/S)), one host-to-HBM copy
        per shard, normalised on the device; with ``ids=None`` they get implicit ids
        ``f"{id_prefix}{n}"`` (n counts bulk rows of this store) and cost no per-row host memory.
        The original vectors are not retained on the host: ``get`` returns the stored (normalised)
        row read back from HBM.  Bulk rows are written to the shards' files by ``shutdown`` (or at once with
        ``persist=True`` / VECTOR_STORE_SAVE_IMMEDIATELY); ranges that were never saved are dropped when the store is
        reopened (``_reconcile``)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.vector_dim:
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {rows.shape}")
        n = rows.shape[0]
        if ids is not None and len(ids) != n:
            raise ValueError("len(ids) != number of rows")
        per = -(-n // self.num_shards) if n else 0
        for s in range(self.num_shards):
            b, e = min(s * per, n), min((s + 1) * per, n)
            if e <= b:
                continue
            label0 = self._bulk_rows + b
            if ids is None:
                self.indices[s].add_rows(None, rows[b:e], id_prefix=id_prefix, first_label=label0,
                                         exact_normalize=exact_normalize)
                self._bulk_ranges.append((id_prefix, label0, e - b, s))
            else:
                self.indices[s].add_rows(list(ids[b:e]), rows[b:e], exact_normalize=exact_normalize)
                for vid in ids[b:e]:
                    self._bulk_id_shard[vid] = s
        self._bulk_rows += n
        for vid, meta in (metadata or {}).items():
            self.metadata[vid] = meta
        self._meta_version += 1
        if persist or (persist is None and self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False)):
            self._save_now()
        return n

    def bulk_store_synthetic(self, n: int, seed: int, counter_row0: int = 0, id_prefix: str = "row_") -> int:
        """Benchmark / test corpora of BASELINE.md section 3 generated ON THE DEVICE (``wdbx_index_fill_synthetic``:
        element (r, c) = ((splitmix64(seed ^ (r * d + c)) >> 40) - 2^23) * 2^-23, rows unit-normalised for cosine),
        placed like ``bulk_store`` (contiguous ranges, implicit ids ``f"{id_prefix}{label}"``)."""
        per = -(-n // self.num_shards) if n else 0
        for s in range(self.num_shards):
            b, e = min(s * per, n), min((s + 1) * per, n)
            if e <= b:
                continue
            label0 = self._bulk_rows + b
            self.indices[s].add_synthetic_rows(seed, counter_row0 + b, e - b, id_prefix=id_prefix, first_label=label0)
            self._bulk_ranges.append((id_prefix, label0, e - b, s))
        self._bulk_rows += n
        self._meta_version += 1
        return n

    async def initialize(self):
        await asyncio.gather(*[ix.initialize() for ix in self.indices])

    async def shutdown(self):
        self._save_now()
        if self._group:
            self._group.close()
        self._group = False
        await asyncio.gather(*[ix.shutdown() for ix in self.indices])
        self.thread_pool.shutdown()
        self._shard_pool.shutdown()

    # ---- the facade's ``vector_store(vector, metadata, id)`` (wdbx.py:241-270) ----
    def _check_dim(self, vector: Sequence[float]) -> None:
        if len(vector) != self.vector_dim:
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {len(vector)}")

    def __call__(self, vector: List[float], metadata: Optional[Dict[str, Any]] = None,
                 id: Optional[str] = None) -> str:
        self._check_dim(vector)
        vector_id = id or str(uuid.uuid4())
        self.store(vector_id, vector, metadata)
        return vector_id

    # ---- ingest ----
    def _remember(self, vector_id: str, vector, metadata) -> np.ndarray:
        vec = np.array(vector, dtype=np.float32)
        self.vectors[vector_id] = vec
        self.metadata[vector_id] = metadata or {}
        self._meta_version += 1
        return vec

    def store(self, vector_id: str, vector: List[float], metadata: Optional[Dict[str, Any]] = None) -> bool:
        try:
            vec = self._remember(vector_id, vector, metadata)
            self.indices[self._get_shard_for_id(vector_id)].add(vector_id, vec)
            if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
                self._save_now()
            return True
        except Exception as e:
            logger.error("Error storing vector: %s", e)
            return False

    async def store_async(self, vector_id: str, vector: List[float],
                          metadata: Optional[Dict[str, Any]] = None) -> bool:
        try:
            vec = self._remember(vector_id, vector, metadata)
            await self.indices[self._get_shard_for_id(vector_id)].add_async(vector_id, vec)
            if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
                await asyncio.get_event_loop().run_in_executor(self.thread_pool, self._save_now)
            return True
        except Exception as e:
            logger.error("Error storing vector asynchronously: %s", e)
            return False

    def _group_by_shard(self, vectors, metadata) -> Dict[int, Dict[str, np.ndarray]]:
        metadata = metadata or {}
        groups: Dict[int, Dict[str, np.ndarray]] = {}
        for vector_id, vector in vectors.items():
            vec = self._remember(vector_id, vector, metadata.get(vector_id, {}))
            groups.setdefault(self._get_shard_for_id(vector_id), {})[vector_id] = vec
        return groups

    def batch_store(self, vectors: Dict[str, List[float]],
                    metadata: Optional[Dict[str, Dict[str, Any]]] = None) -> int:
        for shard, vecs in self._group_by_shard(vectors, metadata).items():
            self.indices[shard].batch_add(vecs)
        if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
            self._save_now()
        return len(vectors)

    async def batch_store_async(self, vectors: Dict[str, List[float]],
                                metadata: Optional[Dict[str, Dict[str, Any]]] = None) -> int:
        groups = self._group_by_shard(vectors, metadata)
        if groups:
            await asyncio.gather(*[self.indices[s].batch_add_async(v) for s, v in groups.items()])
        if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
            await asyncio.get_event_loop().run_in_executor(self.thread_pool, self._save_now)
        return len(vectors)

    # ---- search ----
    def _matches_filter(self, vector_id: str, filter_metadata: Dict[str, Any]) -> bool:
        return matches_filter(self.metadata.get(vector_id, {}), filter_metadata)

    def _merge(self, shard_results, limit: int, threshold: float,
               filter_metadata: Optional[Dict[str, Any]]) -> List[Result]:
        """vector_store.py:323-351: shard order concat, stable sort by score
        descending, ``threshold > 0`` keeps ``score >= threshold``, metadata
        post-filter, cut to ``limit``, attach metadata."""
        merged: List[Tuple[str, float]] = []
        for results in shard_results:
            merged.extend(results)
        merged.sort(key=lambda r: r[1], reverse=True)
        if threshold > 0:
            merged = [r for r in merged if r[1] >= threshold]
        if filter_metadata:
            merged = [r for r in merged if self._matches_filter(r[0], filter_metadata)]
        return [(vid, score, self.metadata.get(vid, {})) for vid, score in merged[:limit]]

    def _row_masks(self, filter_metadata: Dict[str, Any]):
        """Per-shard row masks of the metadata filter (push-down), cached until the
        store changes."""
        key = json.dumps(filter_metadata, sort_keys=False, default=str)
        version = (sum(ix.next_index for ix in self.indices), len(self.metadata), self._meta_version)
        cache = self._mask_cache
        if cache.get("version") != version:
            cache.clear()
            cache["version"] = version
        if key not in cache:
            cache[key] = [ix.row_mask_for(lambda vid: self._matches_filter(vid, filter_metadata))
                          for ix in self.indices]
        return cache[key]

    # ---- fan-out over the shards ----
    def _shard_group(self):
        """The in-library shard group (``wdbx_group_attach`` over this store's per-shard handles): every shard's
        launches enqueued by its own host thread inside ONE library call, the per-shard lists exchanged by RCCL
        all-gather (one shard per GPU) or by device copies (shards sharing a GPU), merged on the device.
        ``HIP_GROUP_SEARCH``: "auto" (default) and True = on for any layout of several shards; "always" = also for a single
        shard; False = off.  A group whose exchange is RCCL (one shard per GPU) is VERIFIED on its first query against the
        per-shard calls + Python merge (``_fan_out``): a differing answer or an error switches it off for this store with one
        log line (tools/preflight_multigpu.py and tests/test_gpu_parity.py::test_rccl_group_over_distinct_devices_* exercise
        the same exchange stage by stage).  Any failure to build it (e.g. RCCL initialisation) is logged once and the
        per-shard calls stay in use."""
        if self._group is not None:
            return self._group or None
        with self._group_lock:
            if self._group is not None:
                return self._group or None
            devices = [ix.device_id for ix in self.indices]
            mode = self.config.get("HIP_GROUP_SEARCH", "auto")
            wanted = mode == "always" or ((mode is True or mode == "auto") and len(self.indices) > 1)
            group: Any = False
            if wanted:
                try:
                    group = _native.NativeGroup.attach([ix._native for ix in self.indices])
                    info = group.info()
                    self._group_stride = info["row_stride"]
                    self._group_path = "rccl_group" if info["rccl_nranks"] > 0 else "copy_group"
                except Exception as e:
                    logger.warning("shard group unavailable, searching shard by shard: %s", e)
                    group = False
            self._group = group
        return self._group or None

    def _group_search(self, queries: np.ndarray, limit: int, keep_all: bool,
                      masks=None) -> Optional[List[List[Tuple[str, float]]]]:
        """One library call for all shards; returns, per query, the merged per-shard candidate list best first
        (the top ``limit`` of it, or with ``keep_all`` the whole union of the shards' top-``limit`` lists, which is what
        the reference's threshold / post-filter see) -- or None when the group cannot serve this call."""
        group = self._shard_group()
        if group is None:
            return None
        shards = len(self.indices)
        if int(limit) > _native.MAX_K:
            return None  # (the per-shard path returns up to MAX_K per shard and merges them: same answer only there)
        k = min(int(limit), max(ix.next_index for ix in self.indices))
        if k <= 0:
            return [[] for _ in range(queries.shape[0])]
        k_out = min(shards * k, sum(min(k, ix.next_index) for ix in self.indices)) if keep_all else k
        k_out = max(k_out, k)
        if k_out > _native.MAX_K:
            return None
        try:
            prepared = np.stack([self.indices[0]._prepare(q) for q in queries])
            idx, score = group.search_merged(prepared, k, k_out, mask_words=masks)
        except Exception as e:
            logger.error("Error searching the shard group: %s", e)
            if all(ix.swallow_errors for ix in self.indices):
                return [[] for _ in range(queries.shape[0])]
            raise
        stride = self._group_stride
        cosine = self.indices[0].metric == _native.METRIC_COSINE
        out = []
        for irow, srow in zip(idx.tolist(), score.tolist()):
            res = []
            for g, sc in zip(irow, srow):
                if g == -1:
                    continue
                res.append((self.indices[g // stride]._id_of(g % stride), float(sc if cosine else -sc)))
            out.append(res)
        return out

    def _fan_out(self, query: np.ndarray, limit: int, masks, post_filtered: bool) -> List[List[Tuple[str, float]]]:
        """Per-shard candidate lists of one query, in shard order (or ONE already merged list from the shard group:
        the stable sort of ``_merge`` leaves it as it is)."""
        # (a pushed-down filter travels with the call: every shard applies its own row mask inside its scan)
        merged = self._group_search(query[None, :], limit, keep_all=post_filtered,
                                    masks=None if all(m is None for m in masks) else masks)
        if merged is not None and self._group_path == "rccl_group" and not self._group_verified:
            merged = self._verify_group_once(query, limit, masks, post_filtered, merged)
        if merged is not None:
            self.last_search_path = self._group_path
            return merged
        self.last_search_path = "threads"
        return self._per_shard(query, limit, masks)

    def _verify_group_once(self, query, limit, masks, post_filtered, merged):
        """First query through an RCCL group: the same query through the per-shard calls, merged as the reference does; ids
        must agree.  On a mismatch the group is closed and the store stays on the per-shard path."""
        per_shard = self._per_shard(query, limit, masks)
        flat = [r for res in per_shard for r in res]
        flat.sort(key=lambda r: r[1], reverse=True)
        want = [vid for vid, _ in flat[: len(merged[0])]]
        got = [vid for vid, _ in merged[0]]
        if got == want:
            self._group_verified = True
            return merged
        logger.error("shard group (RCCL exchange) disagrees with the per-shard path on its first query (%s vs %s): "
                     "switched off for this store", got[:5], want[:5])
        with self._group_lock:
            try:
                if self._group:
                    self._group.close()
            finally:
                self._group = False
        return None

    def _per_shard(self, query: np.ndarray, limit: int, masks) -> List[List[Tuple[str, float]]]:
        if len(self.indices) > 1:
            # the reference loops over its shards one after the other (vector_store.py:325-327); here every
            # shard is a GPU-resident index behind a GIL-releasing call, so the fan-out runs concurrently
            # (one worker per shard) and the results are gathered in shard order -- same answer
            return list(self._shard_pool.map(lambda a: a[0].search(query, limit=limit, row_mask=a[1]),
                                             zip(self.indices, masks)))
        return [ix.search(query, limit=limit, row_mask=m) for ix, m in zip(self.indices, masks)]

    def _masks_for(self, filter_metadata, prefilter: Optional[bool]):
        if prefilter is None:
            prefilter = bool(self.config.get("FILTER_PUSHDOWN", False))
        return self._row_masks(filter_metadata) if (prefilter and filter_metadata) else [None] * len(self.indices)

    def search(self, query_vector: List[float], limit: int = 10, threshold: float = 0.0,
               filter_metadata: Optional[Dict[str, Any]] = None, prefilter: Optional[bool] = None) -> List[Result]:
        """``prefilter=True`` (or config ``FILTER_PUSHDOWN``) evaluates the metadata filter BEFORE the
        scan, so a filtered query returns a full ``limit`` whenever enough rows match; the default keeps
        the reference's post-filter (vector_store.py:337-342), which can under-return."""
        query = _as_query(query_vector)
        masks = self._masks_for(filter_metadata, prefilter)
        if self._sync_coalesce and all(m is None for m in masks) and query.shape == (self.vector_dim,):
            return self._search_coalesced(query, int(limit), threshold, filter_metadata)
        shard_results = self._fan_out(query, limit, masks, post_filtered=bool(filter_metadata))
        return self._merge(shard_results, limit, threshold, filter_metadata)

    # ---- synchronous callers on several threads: leader / follower coalescing ----
    def _search_coalesced(self, query: np.ndarray, limit: int, threshold: float, flt) -> List[Result]:
        """The reference serves ``search`` from thread pools (indexing.py:692, :1045-1048); N threads calling a GPU-resident
        index one query at a time would be N serial corpus scans behind the handle's mutex.  Instead: a caller that finds no
        search in flight becomes the LEADER and runs at once (a lone caller pays a lock and a list append: nothing waits for
        company).  Callers that arrive meanwhile queue up; when the leader is done it hands over to the first of them, which
        answers EVERYTHING queued by then -- itself included -- with ONE batched pass per shard (the matrix-core kernels from
        4 queries up, per-query scans inside one call below that) and wakes the others.  Each caller keeps its own limit,
        threshold and filter; a shard's top-kmax list cut to ``limit`` is its top-``limit`` list, so the answers are those of
        one-at-a-time calls (config ``SYNC_COALESCE=False`` restores them)."""
        req = _SyncRequest(query, limit, threshold, flt)
        with self._sync_lock:
            self._sync_pending.append(req)
            lead = not self._sync_busy
            if lead:
                self._sync_busy = True
        if not lead:
            req.gate.acquire()            # woken by the leader that served (or promoted) this request
            if not req.promoted:
                return self._sync_finish(req)
        # leader: everything queued so far, own request included.  A PROMOTED leader (so: callers are contending) first gives
        # the callers of the batch that has just been answered a moment to come back -- they are re-entering search() right
        # now, and a batch taken this instant would hold one or two queries where a few microseconds later it holds them all
        # (a first leader never waits: a lone caller is served at once)
        batch: List[Any] = []
        try:
            if req.promoted and self._sync_last_batch > 2:
                want, deadline = self._sync_last_batch - 1, time.perf_counter() + 40e-6
                while len(self._sync_pending) < want and time.perf_counter() < deadline:
                    time.sleep(0)         # (yields the GIL to the callers on their way in)
            with self._sync_lock:
                batch, self._sync_pending = self._sync_pending, []
            self._sync_last_batch = len(batch)
            self._serve_sync_batch(batch)
        except BaseException as e:  # every waiter of this batch sees the failure
            for r in batch:
                if r.result is None and r.error is None:
                    r.error = e if isinstance(e, Exception) else RuntimeError(str(e))
            if not batch or not isinstance(e, Exception):
                raise
        finally:
            # whatever happened above (an interrupt in the wait included), leadership is handed on or given up: a leader that
            # left with _sync_busy still set would park every later caller for ever
            with self._sync_lock:
                if self._sync_pending:
                    nxt = self._sync_pending[0]
                    nxt.promoted = True
                    nxt.gate.release()
                else:
                    self._sync_busy = False
            for r in batch:
                if r is not req:
                    r.gate.release()
        return self._sync_finish(req)

    @staticmethod
    def _sync_finish(req) -> List[Result]:
        if req.error is not None:
            raise req.error
        return req.result

    def _serve_sync_batch(self, batch) -> None:
        if len(batch) == 1:
            r = batch[0]
            res = self._fan_out(r.query, r.limit, [None] * len(self.indices), bool(r.flt))
            r.result = self._merge(res, r.limit, r.threshold, r.flt)
            return
        kmax = max(r.limit for r in batch)
        queries = np.stack([r.query for r in batch])
        merged = None
        if all(r.limit == kmax for r in batch) and self._shard_group() is not None and (
                self._group_path != "rccl_group" or self._group_verified):
            merged = self._group_search(queries, kmax, keep_all=any(bool(r.flt) for r in batch))
        if merged is not None:
            self.last_search_path = self._group_path
            for m, r in zip(merged, batch):
                r.result = self._merge([m], r.limit, r.threshold, r.flt)
            return
        self.last_search_path = "threads"
        raw_capable = all(hasattr(ix, "search_batch_raw") for ix in self.indices)
        if raw_capable:
            if len(self.indices) > 1:
                raws = list(self._shard_pool.map(lambda ix: ix.search_batch_raw(queries, limit=kmax), self.indices))
            else:
                raws = [self.indices[0].search_batch_raw(queries, limit=kmax)]
            # (the leader maps and merges for everybody before it wakes anybody: the callers then come back together, and the
            # next batch is as large as this one -- with the tails on the callers' own threads the arrivals spread out and the
            # batches shrank: 11.7 k vs 14.5 k q/s from 8 threads on 1 M rows)
            for i, r in enumerate(batch):
                lists = [[] if raw is None else ix._map(raw[0][i][: r.limit], raw[1][i][: r.limit])
                         for ix, raw in zip(self.indices, raws)]
                r.result = self._merge(lists, r.limit, r.threshold, r.flt)
            return
        if len(self.indices) > 1:
            per_shard = list(self._shard_pool.map(lambda ix: ix.search_batch(queries, limit=kmax), self.indices))
        else:
            per_shard = [self.indices[0].search_batch(queries, limit=kmax)]
        for i, r in enumerate(batch):
            r.result = self._merge([res[i][: r.limit] for res in per_shard], r.limit, r.threshold, r.flt)

    async def search_async(self, query_vector: List[float], limit: int = 10, threshold: float = 0.0,
                           filter_metadata: Optional[Dict[str, Any]] = None,
                           prefilter: Optional[bool] = None) -> List[Result]:
        """Same contract as the reference (vector_store.py:355-412).  Concurrent callers on one event
        loop are COALESCED: whatever is queued while the previous batch runs is answered by one
        batched pass per shard (the matrix-core kernels from 4 queries up) instead of one corpus scan
        per caller.  No waiting window: a lone caller is served at once, exactly as before.  Every
        shard is still asked for each query's own top-``limit``; results are the exact ones
        (config ``ASYNC_COALESCE=False`` restores one call per query).  ``prefilter`` / FILTER_PUSHDOWN as in
        ``search`` (such callers are served one by one: the batched pass takes no row masks)."""
        query = np.array(query_vector, dtype=np.float32)
        if query.shape != (self.vector_dim,):
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {query.shape}")
        loop = asyncio.get_running_loop()
        masks = self._masks_for(filter_metadata, prefilter)
        if not self.config.get("ASYNC_COALESCE", True) or any(m is not None for m in masks):
            shard_results = await loop.run_in_executor(self.thread_pool, self._fan_out, query, limit, masks,
                                                       bool(filter_metadata))
            return self._merge(shard_results, limit, threshold, filter_metadata)
        fut = loop.create_future()
        self._pending.append((query, int(limit), threshold, filter_metadata, fut))
        if self._drain_task is None or self._drain_task.done():
            self._drain_task = loop.create_task(self._drain_pending())
        return await fut

    async def _drain_pending(self) -> None:
        loop = asyncio.get_running_loop()
        nomask = [None] * len(self.indices)
        while self._pending:
            batch, self._pending = self._pending, []
            try:
                if len(batch) == 1:
                    query, limit, threshold, flt, fut = batch[0]
                    res = await loop.run_in_executor(self.thread_pool, self._fan_out, query, limit, nomask, bool(flt))
                    if not fut.done():
                        fut.set_result(self._merge(res, limit, threshold, flt))
                    continue
                kmax = max(b[1] for b in batch)
                queries = np.stack([b[0] for b in batch])
                if all(b[1] == kmax for b in batch) and self._shard_group() is not None:
                    # same limit everywhere: ONE group call answers the whole batch (a caller with a filter needs the
                    # union of the shards' lists, as in ``search``)
                    keep_all = any(bool(b[3]) for b in batch)
                    merged = await loop.run_in_executor(self.thread_pool, self._group_search, queries, kmax, keep_all)
                    if merged is not None:
                        self.last_search_path = self._group_path
                        for m, (_, limit, threshold, flt, fut) in zip(merged, batch):
                            if not fut.done():
                                fut.set_result(self._merge([m], limit, threshold, flt))
                        continue
                per_shard = await asyncio.gather(*[
                    loop.run_in_executor(ix.thread_pool, ix.search_batch, queries, kmax) for ix in self.indices])
                for i, (_, limit, threshold, flt, fut) in enumerate(batch):
                    if not fut.done():
                        # a shard's top-kmax list cut to `limit` IS its top-`limit` list
                        fut.set_result(self._merge([res[i][:limit] for res in per_shard], limit, threshold, flt))
            except Exception as e:  # deliver the failure to every waiter of this batch
                for b in batch:
                    if not b[4].done():
                        b[4].set_exception(e)

    def search_batch(self, queries, limit: int = 10, threshold: float = 0.0,
                     filter_metadata: Optional[Dict[str, Any]] = None) -> List[List[Result]]:
        """Extension: one corpus pass per shard for a whole query batch -- through the shard group when there is
        one (every shard runs its matrix-core pass, or per-query scans for a few queries, inside ONE library call, the
        lists are exchanged and merged on the device), else shard by shard on the shard pool."""
        queries = np.asarray(queries, dtype=np.float32)
        if queries.ndim != 2 or queries.shape[1] != self.vector_dim:
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {queries.shape}")
        if queries.shape[0]:
            merged = self._group_search(queries, limit, keep_all=bool(filter_metadata))
            if merged is not None:
                self.last_search_path = self._group_path
                return [self._merge([m], limit, threshold, filter_metadata) for m in merged]
        self.last_search_path = "threads"
        if len(self.indices) > 1:  # shards run concurrently (GIL-releasing calls), gathered in shard order
            per_shard = list(self._shard_pool.map(lambda ix: ix.search_batch(queries, limit=limit), self.indices))
        else:
            per_shard = [ix.search_batch(queries, limit=limit) for ix in self.indices]
        return [self._merge([res[q] for res in per_shard], limit, threshold, filter_metadata)
                for q in range(queries.shape[0])]

    # ---- row management ----
    def _known(self, vector_id: str) -> bool:
        return vector_id in self.vectors or self._is_bulk(vector_id)

    def _forget(self, vector_id: str) -> None:
        self.vectors.pop(vector_id, None)
        self.metadata.pop(vector_id, None)
        self._bulk_id_shard.pop(vector_id, None)
        self._meta_version += 1  # cached push-down masks name rows by position: rebuild them

    def delete(self, vector_id: str) -> bool:
        if not self._known(vector_id):
            return False
        self.indices[self._get_shard_for_id(vector_id)].remove(vector_id)
        self._forget(vector_id)
        if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
            self._save_now()
        return True

    async def delete_async(self, vector_id: str) -> bool:
        if not self._known(vector_id):
            return False
        await self.indices[self._get_shard_for_id(vector_id)].remove_async(vector_id)
        self._forget(vector_id)
        if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
            await asyncio.get_event_loop().run_in_executor(self.thread_pool, self._save_now)
        return True

    def _set_metadata(self, vector_id: str, metadata: Dict[str, Any]) -> bool:
        if not self._known(vector_id):
            return False
        self.metadata[vector_id] = metadata
        self._meta_version += 1
        return True

    def update_metadata(self, vector_id: str, metadata: Dict[str, Any]) -> bool:
        if not self._set_metadata(vector_id, metadata):
            return False
        if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
            self._save_metadata()
        return True

    async def update_metadata_async(self, vector_id: str, metadata: Dict[str, Any]) -> bool:
        if not self._set_metadata(vector_id, metadata):
            return False
        if self.config.get("VECTOR_STORE_SAVE_IMMEDIATELY", False):
            await asyncio.get_event_loop().run_in_executor(self.thread_pool, self._save_metadata)
        return True

    def get(self, vector_id: str) -> Optional[Tuple[List[float], Dict[str, Any]]]:
        if vector_id not in self.vectors:
            ix = self.indices[self._get_shard_for_id(vector_id)]
            row = ix._row_of(vector_id)
            if row is None:
                return None
            return ix._native.get_rows(row, 1)[0].tolist(), self.metadata.get(vector_id, {})
        return self.vectors[vector_id].tolist(), self.metadata.get(vector_id, {})

    async def get_async(self, vector_id: str) -> Optional[Tuple[List[float], Dict[str, Any]]]:
        return self.get(vector_id)

    def count(self) -> int:
        """Rows that still have an id, over all shards (explicit ids + implicit bulk rows - removed ones): what a search
        can return.  Equals the reference's ``len(self.vectors)`` (vector_store.py:651) for per-vector ingest."""
        return sum(ix.size() for ix in self.indices)

    def _forget_bulk(self) -> None:
        self._bulk_ranges, self._bulk_id_shard, self._bulk_rows = [], {}, 0
        self._meta_version += 1

    def clear(self) -> int:
        removed = self.count()
        for ix in self.indices:
            ix.clear()
        self.vectors, self.metadata = {}, {}
        self._forget_bulk()
        self._save_now()
        return removed

    async def clear_async(self) -> int:
        removed = self.count()
        await asyncio.gather(*[ix.clear_async() for ix in self.indices])
        self.vectors, self.metadata = {}, {}
        self._forget_bulk()
        await asyncio.get_event_loop().run_in_executor(self.thread_pool, self._save_now)
        return removed

    def get_stats(self) -> Dict[str, Any]:
        return {
            "vector_count": self.count(),
            "metadata_count": len(self.metadata),
            "index_type": self.index_type,
            "num_shards": self.num_shards,
            "vector_dim": self.vector_dim,
            "use_gpu": self.use_gpu,
            "indices": [{"shard": i, "type": self.index_type, "size": ix.size(), "stats": ix.get_stats()}
                        for i, ix in enumerate(self.indices)],
        }

    def _after_optimize(self) -> None:
        self._meta_version += 1  # cached push-down masks name rows by position: rebuild them
        self._mask_cache.clear()

    def optimize(self) -> bool:
        """vector_store.py:696-713: every shard's ``optimize`` (here: compaction of removed rows, indexing.py)."""
        ok = all([ix.optimize() for ix in self.indices])
        self._after_optimize()
        return ok

    async def optimize_async(self) -> bool:
        res = await asyncio.gather(*[ix.optimize_async() for ix in self.indices])
        self._after_optimize()
        return all(res)
