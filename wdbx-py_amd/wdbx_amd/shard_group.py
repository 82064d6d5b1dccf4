"""Shards across GPUs: one process per GPU, contiguous row ranges, merge of per-shard top-k.

The reference fans a query out over in-process shard objects and merges with
``list.sort`` (wdbx/core/vector_store.py:323-345).  Here every rank owns one shard in
its GPU's HBM and the per-shard ``(row, score)`` records are exchanged INSIDE the library:
``ncclAllGather`` on the shard's stream followed by the merge kernel
(``wdbx_index_search_sharded_device``); nothing touches the host between scan and merged
result (``search_device``).

``search`` is the host-side form of the same fan-out -- blocking local searches, an
all-gather of the records, the numpy merge below -- used as a cross-check of the device
path.  Its exchange is a callable ``exchange(payload: bytes) -> [payload of rank 0, ...]``;
the default goes through the shard's own RCCL communicator
(``wdbx_index_comm_allgather_host``).  There is no other transport in the product: the
world_size-2 CPU tests plug a ``gloo`` all-gather in from ``tests/``.

With contiguous row ranges "stable sort by score keeping shard order"
(vector_store.py:330) is the same as the global order (score desc, row asc), so
the merged result equals the single-shard result exactly.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

METRIC_COSINE = 0
METRIC_L2 = 1


def shard_row_range(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous rows [begin, end) of ``rank``: ``ceil(total/world)`` rows per shard,
    the last shards possibly shorter or empty (row r -> shard r // ceil(N/S))."""
    per = -(-int(total_rows) // int(world_size))
    begin = min(rank * per, total_rows)
    return begin, min(begin + per, total_rows)


def merge_topk(idx_lists: Sequence[np.ndarray], score_lists: Sequence[np.ndarray], k: int,
               metric: int = METRIC_COSINE) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-shard results (global row numbers, -1 = unused slot) into the k best by
    (score desc, row asc) for cosine / (distance asc, row asc) for L2."""
    idx = np.concatenate([np.asarray(a, np.int64).ravel() for a in idx_lists])
    score = np.concatenate([np.asarray(a, np.float32).ravel() for a in score_lists])
    keep = idx >= 0
    idx, score = idx[keep], score[keep]
    rank_val = score.astype(np.float64) if metric == METRIC_L2 else -score.astype(np.float64)
    order = np.lexsort((idx, rank_val))[:k]
    out_idx = np.full(k, -1, np.int64)
    out_score = np.zeros(k, np.float32)
    out_idx[: order.size] = idx[order]
    out_score[: order.size] = score[order]
    return out_idx, out_score


class ShardGroup:
    """The shard fan-out/merge of one rank.  ``search`` returns the same global result on
    every rank."""

    def __init__(self, rank: int, world_size: int, row_base: int, metric: int = METRIC_COSINE,
                 local_index=None, local_search: Optional[Callable] = None,
                 exchange: Optional[Callable[[bytes], List[bytes]]] = None):
        if local_index is None and local_search is None:
            raise ValueError("need a local index or a local search callable")
        self.rank, self.world_size, self.row_base, self.metric = rank, world_size, int(row_base), metric
        self.index = local_index
        self._local_search = local_search or (lambda q, k: local_index.search(q, k))
        self._exchange = exchange
        self._rccl_ready = False

    # ---- RCCL inside the library ----
    def init_rccl(self, unique_id: bytes) -> None:
        self.index.comm_init(self.world_size, self.rank, unique_id, self.row_base)
        self._rccl_ready = True

    def search_device(self, d_queries, nq: int, k: int, d_idx, d_score, query_offset: int = 0) -> None:
        """Asynchronous, device-resident form: scan, all-gather and merge on the shard's stream."""
        if not self._rccl_ready:
            raise RuntimeError("RCCL communicator not initialised (init_rccl)")
        self.index.search_device(d_queries, nq, k, d_idx, d_score, query_offset=query_offset, sharded=True)

    # ---- host-side form (cross-check) ----
    def _allgather(self, payload: bytes) -> List[bytes]:
        if self.world_size == 1:
            return [payload]
        if self._exchange is not None:
            return list(self._exchange(payload))
        if self._rccl_ready:
            # the records travel through the shard's own RCCL communicator, staged by the library
            return self.index.comm_allgather_host(payload, self.world_size)
        raise RuntimeError("no exchange: call init_rccl() or pass exchange=")

    def search(self, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries.reshape(1, -1)
        nq = queries.shape[0]
        l_idx, l_score = self._local_search(queries, k)
        l_idx = np.asarray(l_idx, np.int64).reshape(nq, k).copy()
        l_score = np.ascontiguousarray(np.asarray(l_score, np.float32).reshape(nq, k))
        l_idx[l_idx >= 0] += self.row_base
        parts = self._allgather(l_idx.tobytes() + l_score.tobytes())
        if len(parts) != self.world_size:
            raise RuntimeError(f"exchange returned {len(parts)} payloads for {self.world_size} ranks")
        g_idx = [np.frombuffer(b[: nq * k * 8], np.int64).reshape(nq, k) for b in parts]
        g_score = [np.frombuffer(b[nq * k * 8:], np.float32).reshape(nq, k) for b in parts]
        out_idx = np.empty((nq, k), np.int64)
        out_score = np.empty((nq, k), np.float32)
        for q in range(nq):
            out_idx[q], out_score[q] = merge_topk([a[q] for a in g_idx], [a[q] for a in g_score], k, self.metric)
        return out_idx, out_score
