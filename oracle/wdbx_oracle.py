"""CPU oracle for the WDBX ``vector_search`` hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of what the reference does on the path
``WDBX.vector_search -> VectorStore.search -> VectorIndex.search``.  It is the
checker for the HIP implementation; nothing under ``wdbx-py_amd/`` imports it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.

Parity status
-------------
* merge / threshold / metadata filter / limit, row normalisation and facade
  validation are PINNED: ``oracle/gen_golden.py`` drives the importable parts of
  the reference (``/root/reference``) and commits the resulting vectors under
  ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this module against
  them.
* the distance + top-k arithmetic itself is PARITY UNPINNED by reference code:
  the reference delegates it to third-party ``faiss-cpu>=1.7.0``
  (``IndexFlatIP``) / ``hnswlib>=0.7.0`` (requirements.txt:18,20 -- version
  floors only, no lock file), neither vendored under ``/root/reference`` nor
  installed.  The restatement follows faiss' published ``IndexFlatIP``
  semantics (exact inner product over all rows, k best by descending score) and
  is anchored on the reference's call sites and its three coarse known-answer
  tests (tests/test_core.py:113-142, :146-192, :196-235).

Reference lines each function follows are cited in the docstrings
(paths relative to /root/reference).
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

METRIC_COSINE = 0  # inner product over unit-normalised rows (reference behaviour)
METRIC_L2 = 1  # extension (the reference has no L2 metric)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


# --------------------------------------------------------------------------- #
# normalisation  (wdbx/core/indexing.py:851-856)
# --------------------------------------------------------------------------- #
def normalize_vector(vector: np.ndarray) -> np.ndarray:
    """``FaissIndex._normalize_vector`` (indexing.py:851-856).

    ``n = np.linalg.norm(v); v / n if n > 0 else v`` in float32; a zero vector is
    returned unchanged.  Same semantics as utils/data_utils.py:291-306.
    """
    vector = np.asarray(vector, dtype=np.float32)
    norm = np.linalg.norm(vector)
    if norm > 0:
        return vector / norm
    return vector


def normalize_rows(rows: np.ndarray) -> np.ndarray:
    """Row-wise ``normalize_vector`` as ``FaissIndex.batch_add`` applies it
    (indexing.py:937-942: one ``_normalize_vector`` per row)."""
    rows = np.asarray(rows, dtype=np.float32)
    out = np.empty_like(rows)
    for i in range(rows.shape[0]):
        out[i] = normalize_vector(rows[i])
    return out


def normalize_rows_fast(rows: np.ndarray) -> np.ndarray:
    """Vectorised row normalisation for large corpora (not bit-pinned: the
    summation order of the norm differs from ``np.linalg.norm`` per row by at
    most an ulp; used only where scores are compared with a tolerance)."""
    rows = np.asarray(rows, dtype=np.float32)
    n = np.sqrt(np.einsum("ij,ij->i", rows, rows, dtype=np.float32)).astype(np.float32)
    n = np.where(n > 0, n, np.float32(1.0)).astype(np.float32)
    return (rows / n[:, None]).astype(np.float32)


# --------------------------------------------------------------------------- #
# exact flat search  (wdbx/core/indexing.py:983-1030, faiss IndexFlatIP)
# --------------------------------------------------------------------------- #
def _topk_desc(scores: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k best entries ordered (score desc, row asc).

    NaN scores are never returned (faiss' heap comparison is false for NaN, so
    such rows never enter the result heap and come back as -1 padding, which
    indexing.py:1023 drops)."""
    n = scores.shape[0]
    valid = ~np.isnan(scores)
    rows = np.nonzero(valid)[0]
    s = scores[rows]
    k = min(k, rows.shape[0])
    if k <= 0:
        return np.empty(0, dtype=np.int64)
    if k < rows.shape[0]:
        # keep everything >= the k-th value so ties at the boundary are ranked
        # by row index, then cut
        kth = np.partition(s, rows.shape[0] - k)[rows.shape[0] - k]
        keep = s >= kth
        rows, s = rows[keep], s[keep]
    order = np.lexsort((rows, -s.astype(np.float64)))
    return rows[order][:k].astype(np.int64)


def flat_scores(rows: np.ndarray, query: np.ndarray, metric: int = METRIC_COSINE) -> np.ndarray:
    """fp32 scores of one query against every stored row.

    cosine: ``s = C_hat @ q_hat`` -- what ``faiss.IndexFlatIP.search`` computes
    at indexing.py:1013 on rows normalised at add time (:886, :939) and a query
    normalised at :1002.  The caller passes rows ALREADY normalised.
    L2 (extension): squared euclidean distance, direct ``sum((c-q)^2)`` form.
    """
    rows = np.asarray(rows, dtype=np.float32)
    query = np.asarray(query, dtype=np.float32)
    if metric == METRIC_COSINE:
        return (rows @ query).astype(np.float32)
    # (rows in chunks that stay in cache: the difference matrix of a 250 k-row slab is 0.8 GB; per row the arithmetic --
    # float32 differences, float32 sum of squares over the row -- does not depend on how many rows share a call)
    out = np.empty(rows.shape[0], np.float32)
    step = max(1, (4 << 20) // max(1, rows.shape[1] * 4))
    for r0 in range(0, rows.shape[0], step):
        diff = rows[r0:r0 + step] - query[None, :]
        out[r0:r0 + step] = np.einsum("ij,ij->i", diff, diff, dtype=np.float32)
    return out


def flat_search(
    rows: np.ndarray,
    query: np.ndarray,
    k: int,
    metric: int = METRIC_COSINE,
    normalize_query: bool = True,
    allowed: Optional[np.ndarray] = None,
) -> Tuple[np.ndarray, np.ndarray]:
    """Exact brute-force top-k over stored rows: (row_index int64[k'], score f32[k']).

    Follows ``FaissIndex.search`` (indexing.py:983-1030): empty index -> nothing
    (:998), query normalised (:1002), ``k' = min(limit, n)`` (:1005), exact
    search (:1013), best first.  Ordering is (score desc, row asc) for cosine and
    (distance asc, row asc) for L2; scores are returned as the backend reports
    them (inner product, or squared distance for L2).
    """
    rows = np.asarray(rows, dtype=np.float32)
    n = rows.shape[0]
    if n == 0 or k <= 0:
        return np.empty(0, np.int64), np.empty(0, np.float32)
    q = np.asarray(query, dtype=np.float32)
    if metric == METRIC_COSINE and normalize_query:
        q = normalize_vector(q)
    s = flat_scores(rows, q, metric)
    rank = s if metric == METRIC_COSINE else -s
    if allowed is not None:  # extension (filter push-down): disallowed rows never compete
        rank = np.where(np.asarray(allowed, bool)[:n], rank, np.float32(np.nan))
    idx = _topk_desc(rank, min(k, n))
    return idx, s[idx]


def slab_search(get_rows, n: int, queries: np.ndarray, k: int, metric: int = METRIC_COSINE,
                slab: int = 500_000) -> List[Tuple[np.ndarray, np.ndarray]]:
    """``flat_search`` for corpora too large to hold on the host at once: the stored rows are
    fetched slab by slab through ``get_rows(first_row, count) -> float32[count, d]`` (e.g. the bytes an
    index reads back from HBM), every slab's exact top-k is kept with global row numbers, and the
    survivors are ranked with the same total order as ``_topk_desc``.  The union of per-slab top-k
    lists contains the global top-k, so the result equals ``flat_search`` on the whole matrix
    (same arithmetic per row: indexing.py:1013 / the direct L2 form).  One (rows, scores) pair per query."""
    queries = np.asarray(queries, dtype=np.float32)
    if queries.ndim == 1:
        queries = queries.reshape(1, -1)
    keep_rows = [[] for _ in queries]
    keep_scores = [[] for _ in queries]
    for r0 in range(0, n, slab):
        rows = get_rows(r0, min(slab, n - r0))
        if metric == METRIC_COSINE:
            s_all = (rows @ queries.T).astype(np.float32)  # one sgemm per slab
        for qi, q in enumerate(queries):
            s = s_all[:, qi] if metric == METRIC_COSINE else flat_scores(rows, q, metric)
            top = _topk_desc(s if metric == METRIC_COSINE else -s, min(k, rows.shape[0]))
            keep_rows[qi].append(top + r0)
            keep_scores[qi].append(s[top])
        del rows
    out = []
    for qi in range(len(queries)):
        r = np.concatenate(keep_rows[qi])
        s = np.concatenate(keep_scores[qi])
        rank = -s.astype(np.float64) if metric == METRIC_COSINE else s.astype(np.float64)
        order = np.lexsort((r, rank))[:k]
        out.append((r[order].astype(np.int64), s[order]))
    return out


def slab_search_screened(get_rows, n: int, queries: np.ndarray, k: int, metric: int = METRIC_COSINE,
                         slab: int = 500_000, extra: int = 16) -> List[Tuple[np.ndarray, np.ndarray, float]]:
    """``slab_search`` plus the near-tie screen of SURVEY 7.2: per query ``(rows[k], scores[k], gap)`` where ``gap`` is the
    smallest difference between neighbours among the best ``k + 1`` FLOAT64 scores.  Exact-id parity between two fp32
    implementations with different summation orders (this oracle's sgemv, the HIP kernels' lane-group trees) can only be
    promised where the true scores are further apart than fp32 rounding; a query with ``gap >= 1e-5`` (ten times the
    worst fp32 error on unit vectors) is one whose ids every correct implementation must return identically.
    The fp64 scores are taken from the same bytes (``get_rows``) for the best ``k + extra`` fp32 candidates of every slab:
    a row outside them cannot be among the best ``k + 1`` in fp64 unless ``extra`` near-ties separate it, in which case
    the reported gap is already tiny."""
    queries = np.asarray(queries, dtype=np.float32)
    if queries.ndim == 1:
        queries = queries.reshape(1, -1)
    kk = k + extra
    keep_rows = [[] for _ in queries]
    keep_s32 = [[] for _ in queries]
    keep_s64 = [[] for _ in queries]
    q64 = queries.astype(np.float64)
    for r0 in range(0, n, slab):
        rows = get_rows(r0, min(slab, n - r0))
        if metric == METRIC_COSINE:
            s_all = (rows @ queries.T).astype(np.float32)
        for qi, q in enumerate(queries):
            s = s_all[:, qi] if metric == METRIC_COSINE else flat_scores(rows, q, metric)
            top = _topk_desc(s if metric == METRIC_COSINE else -s, min(kk, rows.shape[0]))
            keep_rows[qi].append(top + r0)
            keep_s32[qi].append(s[top])
            keep_s64[qi].append(flat_scores_f64(rows[top], q64[qi], metric))
        del rows
    out = []
    for qi in range(len(queries)):
        r = np.concatenate(keep_rows[qi])
        s = np.concatenate(keep_s32[qi])
        s64 = np.concatenate(keep_s64[qi])
        rank = -s.astype(np.float64) if metric == METRIC_COSINE else s.astype(np.float64)
        order = np.lexsort((r, rank))[:k]
        out.append((r[order].astype(np.int64), s[order], min_adjacent_gap(s64, k, metric)))
    return out


class ParallelFlatSearch:
    """``flat_search`` on all host cores: the corpus cut into one contiguous row slab per worker thread, every worker
    scoring ITS slab with the same arithmetic as ``flat_scores`` (a single-threaded sgemv: numpy releases the GIL inside it)
    and ranking it with ``_topk_desc``; the per-slab lists are merged with the same total order, so the result equals
    ``flat_search`` on the whole matrix -- the form of BASELINE.md section 2's "numpy restatement on all host cores"
    (numpy's bundled OpenBLAS stops at 64 threads, and one sgemv call streams from one thread team).
    Each worker pins itself to one CPU and copies its slab there first (first touch = local memory), so the slabs are
    spread over the NUMA nodes instead of sitting where the loader thread put them.  ``search_many`` runs a LIST of
    queries with no barrier between them: every worker walks the whole list over its own slab; throughput = queries / wall."""

    def __init__(self, rows: np.ndarray, workers: Optional[int] = None, metric: int = METRIC_COSINE, pin: bool = True):
        import os
        import threading

        rows = np.asarray(rows, dtype=np.float32)
        try:
            cpus = sorted(os.sched_getaffinity(0))
        except AttributeError:  # not Linux
            cpus = list(range(os.cpu_count() or 1))
        self.workers = max(1, min(int(workers or len(cpus)), max(1, rows.shape[0])))
        self.metric = metric
        n = rows.shape[0]
        per = -(-n // self.workers)
        self.bounds = [(min(i * per, n), min((i + 1) * per, n)) for i in range(self.workers)]
        self.slabs: List[Optional[np.ndarray]] = [None] * self.workers
        self.pinned = 0

        def place(i):
            if pin:
                try:
                    os.sched_setaffinity(0, {cpus[i % len(cpus)]})
                    self.pinned += 1
                except (AttributeError, OSError):
                    pass
            b, e = self.bounds[i]
            self.slabs[i] = np.array(rows[b:e], dtype=np.float32, order="C", copy=True)

        # one persistent thread per slab: it placed the slab, it scans it
        self._jobs: List[Any] = [None] * self.workers
        self._go = [threading.Event() for _ in range(self.workers)]
        self._done = [threading.Event() for _ in range(self.workers)]
        self._out: List[Any] = [None] * self.workers
        self._stop = False

        def loop(i):
            place(i)
            self._done[i].set()
            while True:
                self._go[i].wait()
                self._go[i].clear()
                if self._stop:
                    return
                queries, k = self._jobs[i]
                slab, (b, _) = self.slabs[i], self.bounds[i]
                res = []
                for q in queries:
                    s = flat_scores(slab, q, self.metric)
                    top = _topk_desc(s if self.metric == METRIC_COSINE else -s, min(k, slab.shape[0]))
                    res.append((top + b, s[top]))
                self._out[i] = res
                self._done[i].set()

        self._threads = [threading.Thread(target=loop, args=(i,), daemon=True) for i in range(self.workers)]
        for t in self._threads:
            t.start()
        for d in self._done:
            d.wait()
            d.clear()

    def search_many(self, queries: np.ndarray, k: int) -> List[Tuple[np.ndarray, np.ndarray]]:
        queries = np.asarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries.reshape(1, -1)
        for i in range(self.workers):
            self._jobs[i] = (queries, k)
            self._go[i].set()
        for d in self._done:
            d.wait()
            d.clear()
        out = []
        for qi in range(queries.shape[0]):
            r = np.concatenate([self._out[i][qi][0] for i in range(self.workers)])
            s = np.concatenate([self._out[i][qi][1] for i in range(self.workers)])
            rank = -s.astype(np.float64) if self.metric == METRIC_COSINE else s.astype(np.float64)
            order = np.lexsort((r, rank))[:k]
            out.append((r[order].astype(np.int64), s[order]))
        return out

    def close(self) -> None:
        self._stop = True
        for g in self._go:
            g.set()
        for t in self._threads:
            t.join(timeout=5)
        self.slabs = []


def flat_scores_f64(rows: np.ndarray, query: np.ndarray, metric: int = METRIC_COSINE) -> np.ndarray:
    """fp64 shadow scores used to screen near-ties: exact-ID parity is
    only promised where adjacent true scores differ by more than fp32 error."""
    r = np.asarray(rows, dtype=np.float64)
    q = np.asarray(query, dtype=np.float64)
    if metric == METRIC_COSINE:
        return r @ q
    d = r - q[None, :]
    return np.einsum("ij,ij->i", d, d)


def min_adjacent_gap(scores_f64: np.ndarray, k: int, metric: int = METRIC_COSINE) -> float:
    """Smallest gap between neighbours among the best k+1 fp64 scores."""
    s = scores_f64 if metric == METRIC_COSINE else -scores_f64
    k1 = min(k + 1, s.shape[0])
    if k1 < 2:
        return float("inf")
    top = np.sort(np.partition(s, s.shape[0] - k1)[s.shape[0] - k1:])[::-1]
    return float(np.min(top[:-1] - top[1:]))


# --------------------------------------------------------------------------- #
# per-shard index search  (wdbx/core/indexing.py:983-1030 incl. id mapping)
# --------------------------------------------------------------------------- #
def index_search(
    ids: Sequence[str],
    rows: np.ndarray,
    query: np.ndarray,
    limit: int = 10,
    metric: int = METRIC_COSINE,
    removed: Optional[set] = None,
) -> List[Tuple[str, float]]:
    """One shard's ``VectorIndex.search`` -> ``[(id, float(score))]`` best first.

    ``ids[r]`` is the string id of stored row r (``index_to_id``,
    indexing.py:697-700, used at :1020-1024).  A row whose id was removed is only
    unmapped in the reference (:1062-1074) and comes back as ``str(row)``.
    ``limit`` is clamped to ``next_index`` (= stored rows, :1005).
    """
    n = len(ids)
    idx, sc = flat_search(rows[:n], query, limit, metric)
    out = []
    for r, s in zip(idx, sc):
        r = int(r)
        name = ids[r]
        if removed is not None and r in removed:
            name = str(r)
        val = float(s) if metric == METRIC_COSINE else -float(s)
        out.append((name, val))
    return out


# --------------------------------------------------------------------------- #
# metadata filter  (wdbx/core/vector_store.py:414-463)
# --------------------------------------------------------------------------- #
def matches_filter(metadata: Dict[str, Any], filter_metadata: Dict[str, Any]) -> bool:
    """``VectorStore._matches_filter`` (vector_store.py:414-463) on one row's
    metadata dict.  Equality, or an operator dict of which only the FIRST key is
    honoured (:431-433); a missing key fails every operator except ``$nin`` and
    ``$exists: False``; unknown ``$op`` is ignored (no branch matches)."""
    for key, value in filter_metadata.items():
        if isinstance(value, dict) and list(value.keys())[0].startswith("$"):
            op = list(value.keys())[0]
            arg = value[op]
            present = key in metadata
            if op == "$gt":
                if not present or metadata[key] <= arg:
                    return False
            elif op == "$lt":
                if not present or metadata[key] >= arg:
                    return False
            elif op == "$gte":
                if not present or metadata[key] < arg:
                    return False
            elif op == "$lte":
                if not present or metadata[key] > arg:
                    return False
            elif op == "$in":
                if not present or metadata[key] not in arg:
                    return False
            elif op == "$nin":
                if present and metadata[key] in arg:
                    return False
            elif op == "$exists":
                if arg and not present:
                    return False
                if not arg and present:
                    return False
        else:
            if key not in metadata or metadata[key] != value:
                return False
    return True


# --------------------------------------------------------------------------- #
# shard fan-out merge  (wdbx/core/vector_store.py:323-351)
# --------------------------------------------------------------------------- #
def merge_shard_results(
    shard_results: Sequence[Sequence[Tuple[str, float]]],
    limit: int = 10,
    threshold: float = 0.0,
    filter_metadata: Optional[Dict[str, Any]] = None,
    metadata: Optional[Dict[str, Dict[str, Any]]] = None,
) -> List[Tuple[str, float, Dict[str, Any]]]:
    """``VectorStore.search`` after the per-shard calls (vector_store.py:323-351):
    concatenate in shard order, STABLE sort by score descending (:330), keep
    ``score >= threshold`` only when ``threshold > 0`` (:333-334), post-filter on
    metadata (:337-342), cut to ``limit`` (:345), attach metadata (:348-351)."""
    metadata = metadata or {}
    merged: List[Tuple[str, float]] = []
    for res in shard_results:
        merged.extend(res)
    merged.sort(key=lambda x: x[1], reverse=True)
    if threshold > 0:
        merged = [r for r in merged if r[1] >= threshold]
    if filter_metadata:
        merged = [r for r in merged if matches_filter(metadata.get(r[0], {}), filter_metadata)]
    merged = merged[:limit]
    return [(i, s, metadata.get(i, {})) for i, s in merged]


def vector_search(
    shard_ids: Sequence[Sequence[str]],
    shard_rows: Sequence[np.ndarray],
    query: Sequence[float],
    limit: int = 10,
    threshold: float = 0.0,
    filter_metadata: Optional[Dict[str, Any]] = None,
    metadata: Optional[Dict[str, Dict[str, Any]]] = None,
    vector_dim: Optional[int] = None,
    metric: int = METRIC_COSINE,
) -> List[Tuple[str, float, Dict[str, Any]]]:
    """Whole path ``WDBX.vector_search`` (wdbx.py:303-336): dimension check with
    the reference's message (:323-326), list -> float32 (vector_store.py:321),
    every shard asked for top-``limit`` (:325-327), then the merge above.
    ``shard_rows[s]`` holds shard s's rows as stored (already normalised for
    cosine)."""
    if vector_dim is not None and len(query) != vector_dim:
        raise ValueError(
            f"Vector dimension mismatch: expected {vector_dim}, got {len(query)}"
        )
    q = np.array(query, dtype=np.float32)
    per_shard = [
        index_search(ids, rows, q, limit, metric) for ids, rows in zip(shard_ids, shard_rows)
    ]
    return merge_shard_results(per_shard, limit, threshold, filter_metadata, metadata)


# --------------------------------------------------------------------------- #
# synthetic corpus generator  (SURVEY 8d / BASELINE.md 3; not reference code)
# --------------------------------------------------------------------------- #
SEED_CORPUS = 0xC0FFEE
SEED_QUERY = 0xBEEF


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wraps mod 2^64)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_rows(seed: int, row0: int, n: int, d: int) -> np.ndarray:
    """Counter-based synthetic rows: element (row, col) is
    ``((splitmix64(seed ^ (row*d + col)) >> 40) - 2^23) * 2^-23`` in [-1, 1),
    exactly representable in fp32, so any row range is reproducible on the
    host, in numpy and on the device."""
    if n == 0:
        return np.empty((0, d), np.float32)
    ctr = (np.arange(row0, row0 + n, dtype=np.uint64)[:, None] * np.uint64(d)
           + np.arange(d, dtype=np.uint64)[None, :])
    h = splitmix64(ctr ^ np.uint64(seed))
    v = (h >> np.uint64(40)).astype(np.int64) - (1 << 23)
    return (v.astype(np.float32) * np.float32(2.0 ** -23)).astype(np.float32)
