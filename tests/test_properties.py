"""Property tests (hypothesis) of the product's host logic against the oracle, whose own behaviour is
pinned by the reference-generated goldens: metadata filter grammar, the store merge, the shard
merge.  No GPU."""
import asyncio

import numpy as np
from hypothesis import given, settings, strategies as st

import wdbx_oracle as O
from wdbx_amd.config import WDBXConfig
from wdbx_amd.shard_group import merge_topk, shard_row_range
from wdbx_amd.vector_store import VectorStore, matches_filter

scalar = st.one_of(st.integers(-5, 5), st.sampled_from(["a", "b", ""]), st.booleans(), st.none(),
                   st.floats(-2, 2, allow_nan=False))
keys = st.sampled_from(["n", "s", "flag", "x"])
meta = st.dictionaries(keys, scalar, max_size=4)
op_value = st.one_of(
    st.builds(lambda v: {"$exists": v}, st.booleans()),
    st.builds(lambda v: {"$in": v}, st.lists(scalar, max_size=3)),
    st.builds(lambda v: {"$nin": v}, st.lists(scalar, max_size=3)),
    st.builds(lambda op, v: {op: v}, st.sampled_from(["$gt", "$lt", "$gte", "$lte"]), st.integers(-5, 5)),
    st.builds(lambda v: {"$bogus": v}, st.integers(0, 1)),
)
filt = st.dictionaries(keys, st.one_of(scalar, op_value), max_size=3)


def _same(f, *a):
    """Both raise (e.g. comparing str with int, as the reference would) or both return the same."""
    try:
        r = f(*a)
    except TypeError:
        return "TypeError"
    return r


@settings(max_examples=400, deadline=None)
@given(meta, filt)
def test_filter_grammar_equals_oracle(m, f):
    assert _same(matches_filter, m, f) == _same(O.matches_filter, m, f)


class _Replay:
    def __init__(self, canned):
        self.canned = canned

    def search(self, q, limit=10, row_mask=None):
        return list(self.canned)[:limit]

    async def search_async(self, q, limit=10):
        return self.search(q, limit)

    def search_batch(self, queries, limit=10):
        return [self.search(q, limit) for q in queries]

    thread_pool = None


cand = st.lists(st.tuples(st.sampled_from([f"v{i}" for i in range(12)]),
                          st.sampled_from([0.9, 0.5, 0.5, 0.1, 0.0, -0.3])), max_size=6).map(
    lambda l: sorted(l, key=lambda t: -t[1]))


@settings(max_examples=200, deadline=None)
@given(st.lists(cand, min_size=1, max_size=4), st.integers(1, 8), st.sampled_from([0.0, -1.0, 0.1, 0.5, 0.9]),
       st.one_of(st.none(), st.just({"n": {"$gte": 0}}), st.just({"s": "a"})))
def test_store_merge_equals_oracle_sync_and_async(shards, limit, threshold, flt):
    from concurrent.futures import ThreadPoolExecutor

    metadata = {f"v{i}": ({"n": i - 3, "s": "a" if i % 2 else "b"} if i % 4 else {}) for i in range(12)}
    vs = VectorStore.__new__(VectorStore)
    vs.indices = [_Replay(s) for s in shards]
    vs.metadata, vs.vector_dim, vs.config = metadata, 4, WDBXConfig({})
    vs._mask_cache, vs._meta_version, vs._pending, vs._drain_task = {}, 0, [], None
    vs._group = False  # no devices here: the per-shard calls
    vs.thread_pool = ThreadPoolExecutor(max_workers=2)
    vs._shard_pool = ThreadPoolExecutor(max_workers=max(1, len(shards)))
    exp = O.merge_shard_results([s[:limit] for s in shards], limit, threshold, flt, metadata)  # each shard answers top-`limit`
    assert vs.search([0, 0, 0, 1], limit=limit, threshold=threshold, filter_metadata=flt) == exp
    assert asyncio.run(vs.search_async([0, 0, 0, 1], limit=limit, threshold=threshold, filter_metadata=flt)) == exp
    vs.thread_pool.shutdown()
    vs._shard_pool.shutdown()


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 400), st.integers(1, 8), st.integers(1, 20), st.integers(0, 2 ** 31), st.sampled_from([0, 1]))
def test_shard_merge_equals_single_shard_oracle(n, world, k, seed, metric):
    rng = np.random.default_rng(seed)
    rows = rng.integers(-3, 4, size=(n, 6)).astype(np.float32)  # small integers: exact scores, many ties
    q = rng.integers(-3, 4, size=6).astype(np.float32)
    idxs, scores = [], []
    for r in range(world):
        b, e = shard_row_range(n, world, r)
        i, s = O.flat_search(rows[b:e], q, k, metric, normalize_query=False)
        pi, ps = np.full(k, -1, np.int64), np.zeros(k, np.float32)
        pi[: len(i)], ps[: len(s)] = i + b, s
        idxs.append(pi)
        scores.append(ps)
    gi, gs = merge_topk(idxs, scores, k, metric)
    oi, os_ = O.flat_search(rows, q, k, metric, normalize_query=False)
    assert gi[: len(oi)].tolist() == oi.tolist() and np.all(gi[len(oi):] == -1)
    assert gs[: len(os_)].tolist() == os_.tolist()
