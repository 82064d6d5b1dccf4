"""Host-side logic of the product (no GPU): filter grammar, config, merge, shard ranges --
checked against the reference-generated golden fixtures and against the oracle."""
import json

import numpy as np
import pytest

import wdbx_oracle as O
from wdbx_amd.config import WDBXConfig
from wdbx_amd.indexing import normalize_vector
from wdbx_amd.shard_group import merge_topk, shard_row_range
from wdbx_amd.vector_store import VectorStore, fnv1a_64, matches_filter


def _load(golden_dir, name):
    with open(golden_dir / f"{name}.json") as f:
        return json.load(f)["data"]


def test_filter_grammar_matches_reference_truth_table(golden_dir):
    t = _load(golden_dir, "filter_table")
    for entry in t["table"]:
        for rid, meta in t["rows"].items():
            assert matches_filter(meta, entry["filter"]) == entry["match"][rid], (entry["filter"], rid)


class _Replay:
    def __init__(self, canned):
        self.canned = canned

    def search(self, q, limit=10, row_mask=None):
        return list(self.canned)[:limit]

    async def search_async(self, q, limit=10):
        return self.search(q, limit)

    def search_batch(self, queries, limit=10):
        return [self.search(q, limit) for q in queries]

    @property
    def thread_pool(self):
        return None  # run_in_executor(None, ...) = the loop's default pool


def _store_with(shards, metadata, threads=4):
    vs = VectorStore.__new__(VectorStore)  # the merge does not need a device
    vs.indices = [_Replay([tuple(x) for x in lst]) for lst in shards]
    vs.metadata = metadata
    vs.vector_dim = 4
    vs.config = WDBXConfig({})
    vs._mask_cache, vs._meta_version = {}, 0
    vs._pending, vs._drain_task = [], None
    vs._group = False  # no devices: the per-shard calls (the shard group needs one GPU per shard)
    import threading
    from concurrent.futures import ThreadPoolExecutor

    vs._sync_lock, vs._sync_pending, vs._sync_busy, vs._sync_coalesce, vs._sync_last_batch = threading.Lock(), [], False, True, 0
    vs._group_lock, vs._group_verified, vs._group_path, vs.last_search_path = threading.Lock(), False, "rccl_group", ""

    vs.thread_pool = ThreadPoolExecutor(max_workers=threads)
    vs._shard_pool = ThreadPoolExecutor(max_workers=max(1, len(shards)))
    return vs


def test_store_merge_matches_reference_goldens(golden_dir):
    import asyncio

    for c in _load(golden_dir, "merge"):
        vs = _store_with(c["shards"], c["metadata"])
        got = vs.search([0.1, 0.2, 0.3, 0.4], limit=c["limit"], threshold=c["threshold"], filter_metadata=c["filter"])
        exp = [(i, s, m) for i, s, m in c["expected"]]
        assert got == exp, c["name"]
        agot = asyncio.run(vs.search_async([0.1, 0.2, 0.3, 0.4], limit=c["limit"], threshold=c["threshold"],
                                           filter_metadata=c["filter"]))
        assert agot == exp, c["name"]


def test_search_async_with_a_one_worker_store_pool_does_not_deadlock():
    """a fan-out that runs ON the store's pool must not wait for workers of that pool.  One worker, two
    shards, coalescing off (the path that hands ``_fan_out`` to ``thread_pool``): each call must return."""
    import asyncio

    shards = [[("a", 0.9), ("b", 0.5)], [("c", 0.8), ("d", 0.1)]]
    vs = _store_with(shards, {}, threads=1)
    vs.config = WDBXConfig({"ASYNC_COALESCE": False})

    async def many():
        return await asyncio.wait_for(asyncio.gather(*[vs.search_async([0.1, 0.2, 0.3, 0.4], limit=3) for _ in range(8)]), 20)

    for res in asyncio.run(many()):
        assert [r[0] for r in res] == ["a", "c", "b"]
    vs.config = WDBXConfig({})  # coalesced: a lone caller is still one fan-out on the pool
    assert [r[0] for r in asyncio.run(asyncio.wait_for(vs.search_async([0.1, 0.2, 0.3, 0.4], limit=2), 20))] == ["a", "c"]


def test_normalize_bit_exact_with_reference(golden_dir):
    for item in _load(golden_dir, "normalize"):
        v = np.frombuffer(bytes.fromhex(item["in"]), dtype=np.float32)
        with np.errstate(all="ignore"):
            got = normalize_vector(v.copy())
        assert np.asarray(got, np.float32).tobytes().hex() == item["out"]


def test_config_matches_reference(golden_dir, monkeypatch):
    g = _load(golden_dir, "config")
    monkeypatch.setenv("WDBX_GOLDEN_ENV", "[1, 2]")
    monkeypatch.setenv("WDBX_GOLDEN_BOOL", "yes")
    monkeypatch.setenv("WDBX_GOLDEN_FLOAT", "2.5")
    cfg = WDBXConfig({"WDBX_INT_OPTION": "42", "WDBX_BOOL_OPTION": "true", "WDBX_LIST_OPTION": "[1, 2, 3]", "HNSW_M": 8})
    for key, val in g["defaults"].items():
        assert WDBXConfig.DEFAULT_CONFIG[key] == val
    assert cfg.get("WDBX_GOLDEN_ENV") == g["env_list"] and cfg.get("WDBX_GOLDEN_BOOL") == g["env_bool"]
    assert cfg.get("WDBX_GOLDEN_FLOAT") == g["env_float"]
    assert cfg.get_typed("WDBX_INT_OPTION", int) == g["typed_int"]
    assert cfg.get_typed("WDBX_BOOL_OPTION", bool) is g["typed_bool"]
    assert cfg.get_typed("WDBX_LIST_OPTION", list) == g["typed_list"]
    assert cfg.get_typed("NONEXISTENT", int, 99) == g["typed_default"]
    assert cfg.get("HNSW_M") == g["override"] and cfg.get_source("HNSW_M") == g["source_override"]
    assert cfg.get_source("FAISS_NPROBE") == g["source_default"]
    assert cfg.get_source("WDBX_GOLDEN_ENV") == g["source_env"]
    # reference tests/test_core.py:57-88
    assert WDBXConfig({"WDBX_TEST_OPTION": "test_value"}).get("WDBX_TEST_OPTION") == "test_value"
    assert cfg.get("NONEXISTENT", "default") == "default" and "HNSW_M" in cfg and len(cfg) > 10


def test_shard_row_ranges_are_contiguous_and_complete():
    for total in (0, 1, 7, 8, 9, 1000, 10_000_000, 80_000_000):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_row_range(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            for (a, b), (c, d) in zip(cuts, cuts[1:]):
                assert b == c and a <= b
    assert shard_row_range(10_000_000, 8, 3) == (3_750_000, 5_000_000)


def test_merge_topk_equals_oracle_single_shard_search():
    rng = np.random.default_rng(5)
    rows = O.normalize_rows_fast(rng.standard_normal((4000, 48)).astype(np.float32))
    rows[100] = rows[3000]  # a cross-shard tie
    q = O.normalize_vector(rows[100] + 0.01 * rng.standard_normal(48).astype(np.float32))
    for world in (1, 2, 3, 8):
        for k in (1, 10, 64):
            idxs, scores = [], []
            for r in range(world):
                b, e = shard_row_range(4000, world, r)
                i, s = O.flat_search(rows[b:e], q, k, normalize_query=False)
                pad_i = np.full(k, -1, np.int64)
                pad_s = np.zeros(k, np.float32)
                pad_i[: len(i)] = i + b
                pad_s[: len(s)] = s
                idxs.append(pad_i)
                scores.append(pad_s)
            gi, gs = merge_topk(idxs, scores, k)
            oi, os_ = O.flat_search(rows, q, k, normalize_query=False)
            assert gi.tolist() == oi.tolist() and gs.tolist() == os_.tolist()


def test_merge_topk_l2_and_padding():
    i, s = merge_topk([np.array([5, -1]), np.array([9, 2])], [np.array([0.5, 0.0], np.float32), np.array([0.25, 0.5], np.float32)],
                      3, metric=1)
    assert i.tolist() == [9, 2, 5] and s.tolist() == [0.25, 0.5, 0.5]
    i, s = merge_topk([np.array([-1, -1])], [np.zeros(2, np.float32)], 2)
    assert i.tolist() == [-1, -1]


def test_fnv_placement_is_process_independent():
    assert fnv1a_64("") == 0xCBF29CE484222325
    assert fnv1a_64("a") == 0xAF63DC4C8601EC8C
    assert fnv1a_64("vec_5") % 2 in (0, 1)


def test_async_coalescing_keeps_per_query_semantics(golden_dir):
    """Many concurrent search_async callers with different limits / thresholds / filters: each gets
    exactly what its own call would have returned (reference goldens), batched or not."""
    import asyncio

    cases = _load(golden_dir, "merge")
    same_shards = [c for c in cases if c["shards"] == cases[0]["shards"] and c["metadata"] == cases[0]["metadata"]]
    assert len(same_shards) >= 15
    vs = _store_with(same_shards[0]["shards"], same_shards[0]["metadata"])

    async def run():
        return await asyncio.gather(*[
            vs.search_async([0.1, 0.2, 0.3, 0.4], limit=c["limit"], threshold=c["threshold"], filter_metadata=c["filter"])
            for c in same_shards])

    got = asyncio.run(run())
    for c, g in zip(same_shards, got):
        assert g == [(i, s, m) for i, s, m in c["expected"]], c["name"]


def test_rest_search_handlers_shapes_and_validation():
    """api/server.py:109-113, :141-152: request fields and defaults, response shape; malformed bodies are rejected."""
    import asyncio

    from wdbx_amd.api import search_batch_endpoint, search_endpoint

    calls = []

    class Fake:
        async def vector_search_async(self, q, limit, threshold, flt):
            calls.append((q, limit, threshold, flt))
            return [("a", 0.5, {"k": 1}), ("b", 0.25, {})][:limit]

        def vector_search_batch(self, qs, limit, threshold, flt):
            return [[(f"q{i}", 1.0, {})] for i, _ in enumerate(qs)]

    w = Fake()
    out = asyncio.run(search_endpoint(w, {"query_vector": [1, 2.5, 3]}))
    assert out == {"results": [{"vector_id": "a", "similarity": 0.5, "metadata": {"k": 1}},
                               {"vector_id": "b", "similarity": 0.25, "metadata": {}}]}
    assert calls[-1] == ([1.0, 2.5, 3.0], 10, 0.0, None)
    asyncio.run(search_endpoint(w, {"query_vector": [1], "limit": 1, "threshold": None, "filter_metadata": {"x": 1}}))
    assert calls[-1] == ([1.0], 1, 0.0, {"x": 1})
    for bad in ({}, {"query_vector": "abc"}, {"query_vector": [1], "limit": "5"}, {"query_vector": [1], "filter_metadata": 3},
                {"query_vector": [1, "x"]}, []):
        with pytest.raises(ValueError):
            asyncio.run(search_endpoint(w, bad))
    out = asyncio.run(search_batch_endpoint(w, {"query_vectors": [[1, 2], [3, 4]], "limit": 3}))
    assert out == {"results": [[{"vector_id": "q0", "similarity": 1.0, "metadata": {}}],
                               [{"vector_id": "q1", "similarity": 1.0, "metadata": {}}]]}
    assert asyncio.run(search_batch_endpoint(w, {"query_vectors": []})) == {"results": []}
    with pytest.raises(ValueError):
        asyncio.run(search_batch_endpoint(w, {"query_vector": [1]}))


def test_bench_traffic_records_are_keyed_by_configuration():
    """bench.py reports `roofline.traffic` from the committed PMC passes only for a run on the configuration the record was
    taken on (a lookup must not pass for a measurement of another size)."""
    import importlib.util
    import json
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("wdbx_bench", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    db = json.loads((root / "profiles" / "hbm_traffic.json").read_text())
    for key in ("t_n1_strong_u8", "c4_i8"):
        rec = db[key]
        assert rec["rows"] == 10_000_000 and rec["dim"] == 384
        assert 0.99 < rec["bytes_per_launch"] / rec["algorithmic_bytes"] < 1.05, key
        assert bench.profiled_traffic(db, key, 10_000_000, 384) == rec["bytes_per_launch"]
        assert bench.profiled_traffic(db, key, 1_250_000, 384) is None      # a shard-sized run: no record
        assert bench.profiled_traffic(db, key, 10_000_000, 768) is None
    assert bench.profiled_traffic(db, "no_such_record", 10_000_000, 384) is None


def test_implicit_id_ranges_are_looked_up_by_bisection_after_a_compaction_splits_them():
    """indexing.py ``optimize`` turns one bulk range into the runs of its surviving rows; ids <-> rows must keep
    resolving (by bisection, not by a linear walk over tens of thousands of runs)."""
    from wdbx_amd.indexing import HipFlatIndex

    ix = HipFlatIndex.__new__(HipFlatIndex)  # (the id tables need no device)
    ix.id_to_index, ix.index_to_id = {"x": 3}, {3: "x"}
    ix._implicit_removed, ix._implicit_key = {12}, None
    # runs (first_row, count, prefix, first_label): rows 0-2 = row_0..2, row 3 explicit, rows 4-9 = row_10..15,
    # rows 10-14 = doc_0..4 (row 12 removed), rows 15-16 = row_100..101
    ix._implicit = [(0, 3, "row_", 0), (4, 6, "row_", 10), (10, 5, "doc_", 0), (15, 2, "row_", 100)]
    assert [ix._id_of(r) for r in (0, 2, 3, 4, 9, 10, 12, 14, 15, 16, 17)] == \
        ["row_0", "row_2", "x", "row_10", "row_15", "doc_0", "12", "doc_4", "row_100", "row_101", "17"]
    assert [ix._row_of(v) for v in ("row_0", "row_2", "row_3", "row_10", "row_15", "row_16", "doc_2", "doc_3", "row_101", "row_099",
                                    "x", "nope", "row_")] == [0, 2, None, 4, 9, None, None, 13, 16, None, 3, None, None]
    ix._implicit.append((17, 1, "row_", 7))  # appended later (another bulk call): picked up
    assert ix._row_of("row_7") == 17 and ix._id_of(17) == "row_7"
    assert sorted(ix.mapped_rows())[:5] == [(0, "row_0"), (1, "row_1"), (2, "row_2"), (3, "x"), (4, "row_10")]


class _RamNative:
    """A stand-in for ``_native.NativeIndex`` that keeps the rows in host memory: just enough of its surface for the
    PERSISTENCE logic of ``HipFlatIndex`` (files, generations, id maps) to run without a device.  No search."""

    def __init__(self, dim, metric=0, device_id=0, capacity_rows=1024):
        self.dim, self.rows = int(dim), np.zeros((0, int(dim)), np.float32)

    def add(self, rows, normalize=False):
        first = self.rows.shape[0]
        self.rows = np.concatenate([self.rows, np.asarray(rows, np.float32).reshape(-1, self.dim)])
        return first

    def set_rows(self, first, rows, normalize=False):
        rows = np.asarray(rows, np.float32).reshape(-1, self.dim)
        self.rows[first:first + rows.shape[0]] = rows

    def get_rows(self, first, n):
        return self.rows[first:first + n].copy()

    def compact(self, src):
        self.rows = self.rows[np.asarray(src, np.int64)].copy()

    def clear(self):
        self.rows = self.rows[:0]

    def size(self):
        return self.rows.shape[0]

    def capacity(self):
        return max(1024, self.rows.shape[0])

    def set_option(self, name, value):
        pass

    def close(self):
        pass


def _ram_index(monkeypatch, tmp_path, dim=8, **config):
    from wdbx_amd import _native, indexing

    monkeypatch.setattr(_native, "NativeIndex", _RamNative)
    cfg = {"HIP_METRIC": "l2", "HIP_AUTOSAVE_ROWS": 0}
    cfg.update(config)
    return indexing.HipFlatIndex(dim, tmp_path / "shard_0" / "index", use_gpu=True, config=cfg)


def _stored(ix):
    """id -> stored vector, through the id maps (what a search would resolve)."""
    return {vid: ix._native.get_rows(row, 1)[0].tolist() for row, vid in ix.mapped_rows()}


def test_save_after_optimize_commits_rows_and_mapping_as_one_generation(monkeypatch, tmp_path):
    """after a compaction the old mapping names rows that have moved.  The save must not write into
    the committed row file: a crash between the row rewrite and the mapping replace has to leave the OLD pair readable
    (every id still resolving to ITS vector), and a completed save the new pair -- never old mapping over shifted rows,
    never a file shorter than its mapping (which makes the loader drop the whole index)."""
    import os

    rng = np.random.default_rng(5)
    vecs = {f"v{i}": rng.standard_normal(8).astype(np.float32) for i in range(40)}
    ix = _ram_index(monkeypatch, tmp_path)
    assert ix.batch_add(vecs) and ix.add_rows(None, rng.standard_normal((30, 8)).astype(np.float32), id_prefix="bulk_")[1] == 30
    assert ix.save() and ix._rows_gen == 0 and not ix.unsaved()
    before = _stored(ix)
    for vid in ("v3", "v4", "v17", "bulk_45", "v39", "bulk_69"):   # live rows at the front, in the middle and at the very end
        assert ix.remove(vid)
        before.pop(vid)
    assert ix.save()                                              # tombstones written in place: still generation 0
    assert ix._rows_gen == 0
    assert ix.optimize() and ix.unsaved() and ix.next_index == 64
    after = _stored(ix)
    assert after == before                                        # compaction keeps every id's vector

    # --- a crash between the row rewrite and the mapping replace: the process dies inside os.replace(mapping) ---
    real_replace = os.replace

    def dying_replace(src, dst):
        if str(dst).endswith(".mapping.json"):
            raise KeyboardInterrupt("killed before the mapping was committed")  # (not an Exception: nothing swallows it)
        return real_replace(src, dst)

    monkeypatch.setattr(os, "replace", dying_replace)
    with pytest.raises(KeyboardInterrupt):
        ix.save()
    monkeypatch.setattr(os, "replace", real_replace)
    survivor = _ram_index(monkeypatch, tmp_path)                  # the next process: loads whatever is committed
    assert survivor.next_index == 70 and survivor._rows_gen == 0  # the OLD pair, whole
    assert _stored(survivor) == before                            # ... and every id resolves to its own vector
    assert not list((tmp_path / "shard_0").glob("index.rows.g*.npy*"))   # the interrupted generation was swept

    # --- the save completes: the new pair, the old generation's file gone ---
    assert ix.save() and ix._rows_gen == 1 and not ix.unsaved()
    files = sorted(p.name for p in (tmp_path / "shard_0").iterdir())
    assert files == ["index.mapping.json", "index.rows.g1.npy"]
    again = _ram_index(monkeypatch, tmp_path)
    assert again.next_index == 64 and again._rows_gen == 1 and _stored(again) == before
    # appends after that go in place into the committed generation
    assert again.add("late", np.ones(8, np.float32)) and again.save() and again._rows_gen == 1
    third = _ram_index(monkeypatch, tmp_path)
    assert third.next_index == 65 and _stored(third)["late"] == [1.0] * 8

    # --- dropping only TAIL rows moves nothing, but the committed file is longer than the new mapping allows: same rule ---
    for vid in ("late", "bulk_68"):
        assert third.remove(vid)
    assert third.optimize() and third.unsaved() and third.next_index == 63
    assert third.save() and third._rows_gen == 2
    assert _stored(_ram_index(monkeypatch, tmp_path)) == {k: v for k, v in before.items() if k != "bulk_68"}

    # --- clear(): an empty generation is committed, nothing of the old corpus can be loaded again ---
    assert third.clear() and third._rows_gen == 3
    empty = _ram_index(monkeypatch, tmp_path)
    assert empty.next_index == 0 and _stored(empty) == {}


def test_threaded_synchronous_callers_are_coalesced_and_keep_their_own_semantics(golden_dir):
    """N threads calling ``VectorStore.search`` must not be N serial scans.  A caller that finds a
    search in flight queues up and the next leader answers the whole queue with one batched pass per shard -- every caller
    still gets ITS limit / threshold / filter applied (the reference-generated merge goldens, replayed from 8 threads at
    once against an index that is slow enough for a queue to form)."""
    import threading
    import time

    cases = [c for c in _load(golden_dir, "merge") if len(c["shards"]) == len(_load(golden_dir, "merge")[0]["shards"])][:12]
    shards = cases[0]["shards"]
    cases = [c for c in cases if c["shards"] == shards and c["metadata"] == cases[0]["metadata"]]
    assert len(cases) >= 3
    vs = _store_with(shards, cases[0]["metadata"])
    calls = {"single": 0, "batches": []}
    lock = threading.Lock()

    class Slow(_Replay):
        def search(self, q, limit=10, row_mask=None):
            with lock:
                calls["single"] += 1
            time.sleep(0.02)
            return super().search(q, limit)

        def search_batch(self, queries, limit=10):
            with lock:
                calls["batches"].append(len(queries))
            time.sleep(0.02)
            return [list(self.canned)[:limit] for _ in queries]

    vs.indices = [Slow([tuple(x) for x in lst]) for lst in shards]
    results, errors = {}, []

    def worker(t):
        try:
            for rep in range(6):
                c = cases[(t + rep) % len(cases)]
                got = vs.search([0.1, 0.2, 0.3, 0.4], limit=c["limit"], threshold=c["threshold"], filter_metadata=c["filter"])
                assert got == [(i, s, m) for i, s, m in c["expected"]], (t, rep, c["name"])
            results[t] = True
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(30)
    assert not errors, errors[:1]
    assert len(results) == 8 and not vs._sync_busy and not vs._sync_pending
    assert calls["batches"] and max(calls["batches"]) >= 4            # queues formed and were served in one pass per shard
    # 48 searches on 2+ shards one at a time would be 48 x shards single calls; coalesced it is a fraction of that
    assert calls["single"] + len(calls["batches"]) < 48 * len(shards) // 2
    # a lone caller is served at once through the one-query path, and SYNC_COALESCE=False restores per-call fan-outs
    before = calls["single"]
    vs.search([0.1, 0.2, 0.3, 0.4], limit=3)
    assert calls["single"] == before + len(shards)
    vs._sync_coalesce = False
    vs.search([0.1, 0.2, 0.3, 0.4], limit=3)
    assert calls["single"] == before + 2 * len(shards)


def test_a_failing_batch_reaches_every_waiter_and_the_queue_keeps_moving():
    import threading
    import time

    vs = _store_with([[("a", 0.9)], [("b", 0.8)]], {})

    class Flaky(_Replay):
        fail = True

        def search(self, q, limit=10, row_mask=None):
            time.sleep(0.03)
            return super().search(q, limit)

        def search_batch(self, queries, limit=10):
            if Flaky.fail:
                raise RuntimeError("device fell over")
            return [list(self.canned)[:limit] for _ in queries]

    vs.indices = [Flaky([("a", 0.9)]), Flaky([("b", 0.8)])]
    outcomes = []

    def worker():
        try:
            outcomes.append([r[0] for r in vs.search([0.1, 0.2, 0.3, 0.4], limit=2)])
        except RuntimeError as e:
            outcomes.append(str(e))

    threads = [threading.Thread(target=worker) for _ in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(20)
    assert len(outcomes) == 6 and ["a", "b"] in outcomes and "device fell over" in outcomes   # first caller fine, a batch failed
    assert not vs._sync_busy and not vs._sync_pending
    Flaky.fail = False
    assert [r[0] for r in vs.search([0.1, 0.2, 0.3, 0.4], limit=2)] == ["a", "b"]              # ... and the store still works
