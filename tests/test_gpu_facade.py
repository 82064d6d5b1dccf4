"""The reference's own core tests (tests/test_core.py in the reference), re-expressed against the
drop-in facade on a real GPU, plus multi-shard parity with the oracle.  Same fixtures and
assertions; ``asyncio.run`` replaces pytest-asyncio (not installed here)."""
import asyncio
import shutil
import tempfile
from pathlib import Path

import numpy as np
import pytest

import wdbx_oracle as O

pytestmark = pytest.mark.gpu

CONFIG = {"WDBX_TEST_OPTION": "test_value", "WDBX_VECTOR_STORE_SAVE_IMMEDIATELY": False}


@pytest.fixture
def temp_dir():
    d = tempfile.mkdtemp()
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture
def db(temp_dir):
    from wdbx_amd import WDBX

    w = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir, config=CONFIG, enable_plugins=False)
    asyncio.run(w.initialize())
    yield w
    asyncio.run(w.shutdown())


def test_wdbx_creation(temp_dir):
    from wdbx_amd import WDBX

    w = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir, config=CONFIG)
    assert w.vector_dim == 4 and w.num_shards == 2 and w.data_dir == Path(temp_dir)
    assert w.config.get("WDBX_TEST_OPTION") == "test_value"
    asyncio.run(w.shutdown())


def test_vector_store_and_search(db):
    # reference tests/test_core.py:113-142
    vector = [0.1, 0.2, 0.3, 0.4]
    vid = db.vector_store(vector, {"source": "test", "content": "test vector"})
    assert db.count_vectors() == 1
    stored, meta = db.get_vector(vid)
    assert len(stored) == 4 and np.allclose(stored, vector) and meta["source"] == "test"
    res = db.vector_search(vector, limit=5)
    assert len(res) == 1
    found, sim, fmeta = res[0]
    assert found == vid and sim > 0.99 and fmeta["source"] == "test"


def test_vector_async_operations(db):
    # reference tests/test_core.py:146-192
    async def run():
        vector = [0.1, 0.2, 0.3, 0.4]
        vid = await db.vector_store_async(vector, {"source": "async_test"})
        assert db.count_vectors() == 1
        stored, meta = await db.get_vector_async(vid)
        assert np.allclose(stored, vector) and meta["source"] == "async_test"
        res = await db.vector_search_async(vector, limit=5)
        assert len(res) == 1 and res[0][0] == vid and res[0][1] > 0.99
        assert await db.update_metadata_async(vid, {"source": "updated"}) is True
        assert (await db.get_vector_async(vid))[1]["source"] == "updated"
        assert await db.delete_vector_async(vid) is True
        assert db.count_vectors() == 0

    asyncio.run(run())


def test_vector_batch_operations(db):
    # reference tests/test_core.py:196-235
    vectors = {f"vec_{i}": [i / 10, (i + 1) / 10, (i + 2) / 10, (i + 3) / 10] for i in range(10)}
    metadata = {vid: {"index": i, "source": "batch_test"} for i, vid in enumerate(vectors)}
    assert db.vector_store.batch_store(vectors, metadata) == 10
    assert db.count_vectors() == 10
    q = [0.5, 0.6, 0.7, 0.8]
    res = db.vector_search(q, limit=1)
    assert len(res) == 1 and res[0][0] == "vec_5"
    flt = db.vector_search(q, limit=10, filter_metadata={"index": {"$lt": 3}})
    assert len(flt) == 3 and all(m["index"] < 3 for _, _, m in flt)
    assert db.clear() == 10 and db.count_vectors() == 0


def test_error_handling(db, golden_dir):
    # reference tests/test_core.py:242-259 + the reference's exact message (golden)
    import json

    with pytest.raises(ValueError, match="dimension mismatch"):
        db.vector_store([0.1, 0.2, 0.3], {})
    with open(golden_dir / "facade.json") as f:
        fac = json.load(f)["data"]
    with pytest.raises(ValueError) as e:
        db.vector_search([0.1, 0.2, 0.3])
    assert str(e.value) == fac["vector_search"]["message"]
    assert db.get_vector("nonexistent_id") is None
    assert db.delete_vector("nonexistent_id") is False
    assert db.update_metadata("nonexistent_id", {"test": "value"}) is False


def test_persistence(temp_dir):
    # reference tests/test_core.py:266-312
    from wdbx_amd import WDBX

    w1 = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir, config=CONFIG)
    asyncio.run(w1.initialize())
    vector = [0.1, 0.2, 0.3, 0.4]
    vid = w1.vector_store(vector, {"source": "persistence_test"})
    w1.vector_store._save_metadata()
    w1.vector_store._save_vectors()
    asyncio.run(w1.shutdown())
    w2 = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir, config=CONFIG)
    asyncio.run(w2.initialize())
    assert w2.count_vectors() == 1
    stored, meta = w2.get_vector(vid)
    assert np.allclose(stored, vector) and meta["source"] == "persistence_test"
    res = w2.vector_search(vector, limit=3)  # the shard's rows came back into HBM too
    assert [r[0] for r in res] == [vid]
    asyncio.run(w2.shutdown())


def test_statistics(db, golden_dir):
    # reference tests/test_core.py:319-341 + key set from the reference (golden)
    import json

    for i in range(5):
        db.vector_store([i / 10, (i + 1) / 10, (i + 2) / 10, (i + 3) / 10], {"index": i})
    stats = db.get_stats()
    assert stats["vector_dimension"] == 4 and stats["num_shards"] == 2 and stats["total_vectors"] == 5
    assert "version" in stats and stats["vector_count"] == 5 and "index_type" in stats
    assert len(stats["indices"]) == 2
    with open(golden_dir / "facade.json") as f:
        fac = json.load(f)["data"]
    assert sorted(stats.keys()) == fac["stats_keys"]
    assert sorted(stats["indices"][0].keys()) == fac["stats_index_entry_keys"]


@pytest.mark.parametrize("shards", [1, 2, 5])
def test_multi_shard_store_matches_oracle(temp_dir, shards):
    """End to end: ids, order, scores and metadata of the facade == oracle.vector_search given the
    same shard contents (threshold, filter, limit all exercised)."""
    from wdbx_amd import WDBX

    d, n = 384, 3000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    vectors = {f"id_{i}": raw[i].tolist() for i in range(n)}
    metadata = {f"id_{i}": {"index": i, "parity": "even" if i % 2 == 0 else "odd"} for i in range(n)}
    w = WDBX(vector_dimension=d, num_shards=shards, data_dir=temp_dir, enable_plugins=False)
    assert w.vector_store.batch_store(vectors, metadata) == n
    shard_ids = [[None] * ix.next_index for ix in w.vector_store.indices]
    for s, ix in enumerate(w.vector_store.indices):
        for vid, row in ix.id_to_index.items():
            shard_ids[s][row] = vid
    shard_rows = [O.normalize_rows(np.array([vectors[v] for v in ids], np.float32)) if ids
                  else np.empty((0, d), np.float32) for ids in shard_ids]
    queries = O.synth_rows(O.SEED_QUERY, 0, 4, d)
    cases = [dict(limit=10), dict(limit=1), dict(limit=50, threshold=0.05),
             dict(limit=20, filter_metadata={"parity": "odd"}),
             dict(limit=10, filter_metadata={"index": {"$lt": 500}}, threshold=0.01)]
    for q in queries:
        for kw in cases:
            got = w.vector_search(q.tolist(), **kw)
            exp = O.vector_search(shard_ids, shard_rows, q.tolist(), metadata=metadata, vector_dim=d, **kw)
            assert [g[0] for g in got] == [e[0] for e in exp], kw
            np.testing.assert_allclose([g[1] for g in got], [e[1] for e in exp], atol=1e-5, rtol=0)
            assert [g[2] for g in got] == [e[2] for e in exp]
            agot = asyncio.run(w.vector_search_async(q.tolist(), **kw))
            assert agot == got
    batch = w.vector_search_batch([q.tolist() for q in queries], limit=10)
    assert batch == [w.vector_search(q.tolist(), limit=10) for q in queries]
    asyncio.run(w.shutdown())


def test_remove_and_replace_semantics(temp_dir):
    from wdbx_amd import WDBX

    w = WDBX(vector_dimension=4, num_shards=1, data_dir=temp_dir, enable_plugins=False)
    w.vector_store([1, 0, 0, 0], {"n": "a"}, id="a")
    w.vector_store([0, 1, 0, 0], {"n": "b"}, id="b")
    w.vector_store([0.9, 0.1, 0, 0], {"n": "c"}, id="c")
    assert [r[0] for r in w.vector_search([1, 0, 0, 0], limit=3)] == ["a", "c", "b"]
    w.vector_store([0, 0, 1, 0], {"n": "a2"}, id="a")  # same id: replaced in place
    res = w.vector_search([1, 0, 0, 0], limit=3)
    assert [r[0] for r in res] == ["c", "a", "b"] and res[1][2] == {"n": "a2"}
    assert w.delete_vector("c") is True and w.count_vectors() == 2
    res = w.vector_search([1, 0, 0, 0], limit=3)
    # the removed row is zeroed and unmapped; like the reference it can still surface as str(row)
    assert [r[0] for r in res][:1] != ["c"] and all(r[1] == 0.0 for r in res)
    asyncio.run(w.shutdown())


def test_filter_pushdown_returns_full_limit_and_matches_oracle(temp_dir):
    """with ``prefilter=True`` the filter is applied before the scan, so a selective
    filter still returns ``limit`` hits; the default keeps the reference's post-filter behaviour."""
    from wdbx_amd import WDBX

    d, n = 64, 4000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    vectors = {f"id_{i}": raw[i].tolist() for i in range(n)}
    metadata = {f"id_{i}": {"index": i, "bucket": i % 50} for i in range(n)}
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, enable_plugins=False)
    w.vector_store.batch_store(vectors, metadata)
    q = O.synth_rows(O.SEED_QUERY, 0, 1, d)[0].tolist()
    flt = {"bucket": 7}  # 2 % of the rows
    post = w.vector_search(q, limit=10, filter_metadata=flt)
    pre = w.vector_search(q, limit=10, filter_metadata=flt, prefilter=True)
    assert len(post) < 10 and len(pre) == 10 and all(m["bucket"] == 7 for _, _, m in pre)
    assert [p[0] for p in post] == [p[0] for p in pre[: len(post)]]  # post-filter hits are the head of the full answer
    # oracle: exact top-10 among allowed rows of each shard, merged
    shard_ids = [[None] * ix.next_index for ix in w.vector_store.indices]
    for s, ix in enumerate(w.vector_store.indices):
        for vid, row in ix.id_to_index.items():
            shard_ids[s][row] = vid
    per_shard = []
    for ids in shard_ids:
        rows = O.normalize_rows(np.array([vectors[v] for v in ids], np.float32))
        allowed = np.array([metadata[v]["bucket"] == 7 for v in ids])
        idx, sc = O.flat_search(rows, np.array(q, np.float32), 10, allowed=allowed)
        per_shard.append([(ids[i], float(s)) for i, s in zip(idx, sc)])
    exp = O.merge_shard_results(per_shard, 10, 0.0, flt, metadata)
    assert [p[0] for p in pre] == [e[0] for e in exp]
    np.testing.assert_allclose([p[1] for p in pre], [e[1] for e in exp], atol=1e-5, rtol=0)
    # cache invalidation: metadata change -> mask recomputed
    w.update_metadata(pre[0][0], {"index": -1, "bucket": 8})
    again = w.vector_search(q, limit=10, filter_metadata=flt, prefilter=True)
    assert pre[0][0] not in [a[0] for a in again] and len(again) == 10
    # nothing matches -> empty
    assert w.vector_search(q, limit=5, filter_metadata={"bucket": 99}, prefilter=True) == []
    asyncio.run(w.shutdown())


def test_bulk_ingest_implicit_ids_matches_oracle_and_persists(temp_dir):
    """contiguous [N, d] ingest (one copy per shard, device normalisation,
    implicit ids), contiguous-range sharding == single-shard answer, flat on-disk format round trip."""
    from wdbx_amd import WDBX

    d, n, shards = 384, 100_003, 3
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    w = WDBX(vector_dimension=d, num_shards=shards, data_dir=temp_dir, enable_plugins=False)
    assert w.vector_store.bulk_store(raw, metadata={"row_5": {"tag": "five"}}) == n
    assert w.count_vectors() == n and w.get_stats()["total_vectors"] == n
    assert sum(ix.size() for ix in w.vector_store.indices) == n
    rows = O.normalize_rows_fast(raw)
    queries = O.synth_rows(O.SEED_QUERY, 0, 3, d)
    expected = []
    for q in queries:
        got = w.vector_search(q.tolist(), limit=10)
        o_idx, o_score = O.flat_search(rows, q, 10)  # single shard, global rows
        assert [g[0] for g in got] == [f"row_{i}" for i in o_idx]
        np.testing.assert_allclose([g[1] for g in got], o_score, atol=1e-5, rtol=0)
        expected.append(got)
    stored, meta = w.get_vector("row_5")
    np.testing.assert_allclose(stored, rows[5], rtol=1e-6, atol=1e-30)
    assert meta == {"tag": "five"} and w.get_vector("row_%d" % n) is None
    # a self query finds its row; deleting it removes it from the answer
    top = w.vector_search(raw[77_777].tolist(), limit=1)[0]
    assert top[0] == "row_77777" and top[1] > 0.9999
    assert w.delete_vector("row_77777") is True and w.count_vectors() == n - 1
    assert w.vector_search(raw[77_777].tolist(), limit=1)[0][0] != "row_77777"
    assert w.delete_vector("row_77777") is False
    # explicit ids + more bulk rows appended later keep working
    w.vector_store.bulk_store(raw[:10] * np.float32(-1.0), ids=[f"neg_{i}" for i in range(10)])
    assert w.vector_search((-raw[3]).tolist(), limit=1)[0][0] == "neg_3"
    # on-disk round trip: rows come back into HBM from the flat files, ids resolve to the same shards
    expected = [w.vector_search(q.tolist(), limit=10) for q in queries]
    asyncio.run(w.shutdown())
    w2 = WDBX(vector_dimension=d, num_shards=shards, data_dir=temp_dir, enable_plugins=False)
    assert w2.count_vectors() == n - 1 + 10
    for q, exp in zip(queries, expected):
        assert w2.vector_search(q.tolist(), limit=10) == exp
    assert w2.get_vector("neg_3") is not None and w2.get_vector("row_77777") is None
    asyncio.run(w2.shutdown())


def test_concurrent_async_callers_are_coalesced_onto_the_batched_kernel(temp_dir):
    """64 concurrent ``vector_search_async`` callers (what the reference's REST server does,
    api/server.py:143): answered by batched passes, each result equal to its own sync search."""
    from wdbx_amd import WDBX

    d, n = 128, 200_000  # 100k rows per shard: above the batched path's 65 536-row floor
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, enable_plugins=False)
    w.vector_store.bulk_store(O.synth_rows(O.SEED_CORPUS, 0, n, d))
    queries = [q.tolist() for q in O.synth_rows(O.SEED_QUERY, 0, 64, d)]
    limits = [1 + (i % 12) for i in range(64)]
    sync = [w.vector_search(q, limit=l) for q, l in zip(queries, limits)]
    natives = [ix._native for ix in w.vector_store.indices]
    for nat in natives:
        nat.profile(True)
        nat.profile_read()
        nat.profile_read_gemm()

    async def run():
        return await asyncio.gather(*[w.vector_search_async(q, limit=l) for q, l in zip(queries, limits)])

    got = asyncio.run(run())
    gemm = sum(nat.profile_read_gemm()["gemm_launches"] for nat in natives)
    scans = sum(nat.profile_read()["scan_launches"] for nat in natives)
    assert gemm >= 2 and scans <= 2 * 2, (gemm, scans)  # the first caller may go alone, the rest ride in batches
    for g, s in zip(got, sync):
        assert [x[0] for x in g] == [x[0] for x in s]
        np.testing.assert_allclose([x[1] for x in g], [x[1] for x in s], atol=2e-6, rtol=0)
    with pytest.raises(ValueError, match="dimension mismatch"):
        asyncio.run(w.vector_search_async([0.0] * 3))
    asyncio.run(w.shutdown())


def test_concurrent_async_stores_keep_the_id_maps_consistent(temp_dir):
    """Many ``vector_store_async`` calls at once (the reference mutates its id maps unguarded from a
    4-worker pool): every vector must land in exactly one row and be found again."""
    from wdbx_amd import WDBX

    d, n = 32, 400
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, enable_plugins=False)

    async def run():
        return await asyncio.gather(*[w.vector_store_async(raw[i].tolist(), {"i": i}, id=f"v{i}") for i in range(n)])

    ids = asyncio.run(run())
    assert ids == [f"v{i}" for i in range(n)] and w.count_vectors() == n
    assert sum(ix.next_index for ix in w.vector_store.indices) == n
    for ix in w.vector_store.indices:
        assert sorted(ix.index_to_id) == list(range(ix.next_index))
    for i in (0, 57, 399):
        top = w.vector_search(raw[i].tolist(), limit=1)[0]
        assert top[0] == f"v{i}" and top[1] > 0.9999 and top[2] == {"i": i}
    asyncio.run(w.shutdown())


def test_reference_examples_shapes(temp_dir):
    """The call shapes of the reference's two examples: examples/basic_usage.py:12-40 (384-d, one
    shard, store one vector, ``vector_search_async(limit=5)``, stats) and
    examples/rag_implementation.py:38-44 (``vector_search_async(limit=5, threshold=0.6)`` feeding a
    context built from ``metadata['content']``), with a plugin object registered the way the
    reference's plugins register themselves."""
    from wdbx_amd import WDBX

    class FakeEmbedder:  # the plugin contract the examples rely on: name/version/create_embedding
        name, version, description = "ollama", "0.0", "test embedder"

        async def create_embedding(self, text):
            rng = np.random.default_rng(abs(hash(text)) % (2 ** 32))
            return rng.standard_normal(384).astype(np.float32).tolist()

    async def run():
        db = WDBX(vector_dimension=384, num_shards=1, data_dir=temp_dir, enable_plugins=True)
        await db.initialize()
        assert db.register_plugin(FakeEmbedder()) is True and "ollama" in db.plugins
        # basic_usage.py
        vector = [0.1] * 384
        vid = await db.vector_store_async(vector, {"source": "example", "content": "Sample text for demonstration"})
        results = await db.vector_search_async(vector, limit=5)
        assert len(results) == 1 and results[0][0] == vid and abs(results[0][1] - 1.0) < 1e-5
        assert results[0][2].get("content") == "Sample text for demonstration"
        stats = db.get_stats()
        assert stats["total_vectors"] == 1 and stats["vector_dimension"] == 384
        # rag_implementation.py: documents embedded by the plugin, retrieval with threshold 0.6
        emb = db.plugins["ollama"]
        docs = [f"document number {i}" for i in range(50)]
        for i, text in enumerate(docs):
            await db.vector_store_async(await emb.create_embedding(text), {"content": text, "i": i})
        q = await emb.create_embedding(docs[7])        # the same text -> similarity 1
        hits = await db.vector_search_async(q, limit=5, threshold=0.6)
        assert [h[2]["content"] for h in hits] == [docs[7]] and hits[0][1] > 0.999
        context = "\\n".join(f"Document {i+1} (Similarity: {s:.2f}):\\n{m['content']}\\n" for i, (_, s, m) in enumerate(hits)
                            if "content" in m)
        assert "document number 7" in context
        unrelated = await db.vector_search_async(await emb.create_embedding("something else"), limit=5, threshold=0.6)
        assert unrelated == []  # nothing reaches 0.6: the reference's "No relevant information found." branch
        await db.shutdown()

    asyncio.run(run())


def test_facade_on_the_selection_paths_at_scale(temp_dir):
    """The WDBX facade over one 300 k-row shard: lone searches run on the u8 selection scan (large enough corpus),
    filter push-down rides on its row mask, deletes and replacements reach the shadow copies, a limit of 500 takes
    the radix-select epilogue, and a batch goes through the bf16 tiles -- every answer against the oracle."""
    from wdbx_amd import WDBX

    d, n = 128, 300_000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    w = WDBX(vector_dimension=d, num_shards=1, data_dir=temp_dir, enable_plugins=False)
    meta = {f"row_{i}": {"lang": "en" if i % 4 else "de", "n": i} for i in range(0, n, 1000)}
    assert w.vector_store.bulk_store(raw, metadata=meta) == n
    rows = O.normalize_rows_fast(raw)
    native_ix = w.vector_store.indices[0]._native
    queries = O.synth_rows(O.SEED_QUERY, 0, 5, d)
    for q in queries[:3]:
        got = w.vector_search(q.tolist(), limit=10)
        o_idx, o_score = O.flat_search(rows, q, 10)
        assert [g[0] for g in got] == [f"row_{i}" for i in o_idx]
        np.testing.assert_allclose([g[1] for g in got], o_score, atol=1e-5, rtol=0)
    assert native_ix.get_option("last_single_path") == 2 and native_ix.get_option("shadow8_rows") == n
    # a limit in the radix-select range
    got = w.vector_search(queries[3].tolist(), limit=250)
    o_idx, o_score = O.flat_search(rows, queries[3], 250)
    assert [g[0] for g in got] == [f"row_{i}" for i in o_idx] and native_ix.get_option("last_single_path") == 2
    # filter push-down: only rows carrying metadata can match {"lang": "de"}; the mask rides on the u8 scan
    got = w.vector_search(queries[0].tolist(), limit=5, filter_metadata={"lang": "de"}, prefilter=True)
    allowed = np.zeros(n, bool)
    allowed[[i for i in range(0, n, 1000) if i % 4 == 0]] = True
    o_idx, o_score = O.flat_search(rows, queries[0], 5, allowed=allowed)
    assert [g[0] for g in got] == [f"row_{i}" for i in o_idx] and all(g[2]["lang"] == "de" for g in got)
    assert native_ix.get_option("last_single_path") == 2
    # delete the current winner, replace another row with the query itself
    top = w.vector_search(queries[1].tolist(), limit=2)
    assert w.delete_vector(top[0][0]) is True
    assert w.vector_search(queries[1].tolist(), limit=1)[0][0] == top[1][0]
    w.vector_store.store("row_4242", queries[1].tolist(), {"replaced": True})
    hit = w.vector_search(queries[1].tolist(), limit=1)[0]
    assert hit[0] == "row_4242" and hit[1] > 0.9999 and hit[2] == {"replaced": True}
    assert native_ix.get_option("shadow8_rows") == n           # refreshed in place, not rebuilt
    # a batch: one pass on the bf16 tiles, same answers as the lone searches
    batch = w.vector_search_batch([q.tolist() for q in queries], limit=3)
    for q, res in zip(queries, batch):
        assert [r[0] for r in res] == [r[0] for r in w.vector_search(q.tolist(), limit=3)]
    assert native_ix.get_option("last_gemm_family") == 2
    asyncio.run(w.shutdown())


def test_shadow_copies_can_be_switched_off_by_config(temp_dir):
    """HIP_U8_SHADOW / HIP_BF16_SHADOW = False: no extra device memory is taken, the answers stay the same."""
    from wdbx_amd import WDBX

    d, n = 128, 250_000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    rows = O.normalize_rows_fast(raw)
    queries = O.synth_rows(O.SEED_QUERY, 0, 6, d)
    answers = []
    for cfg, single_path, family in (({}, 2, 2), ({"HIP_U8_SHADOW": False}, 1, 2), ({"HIP_U8_SHADOW": False, "HIP_BF16_SHADOW": False}, 0, 1)):
        w = WDBX(vector_dimension=d, num_shards=1, data_dir=f"{temp_dir}/{single_path}{family}", config=cfg, enable_plugins=False)
        w.vector_store.bulk_store(raw)
        ix = w.vector_store.indices[0]._native
        lone = [w.vector_search(q.tolist(), limit=5) for q in queries[:2]]
        assert ix.get_option("last_single_path") == single_path
        batch = w.vector_search_batch([q.tolist() for q in queries], limit=5)
        assert ix.get_option("last_gemm_family") == family
        assert (ix.get_option("shadow8_bytes") > 0) == (single_path == 2)
        assert (ix.get_option("shadow_bytes") > 0) == (family == 2)
        answers.append(([[r[0] for r in res] for res in lone], [[r[0] for r in res] for res in batch]))
        asyncio.run(w.shutdown())
    for q, ids in zip(queries[:2], answers[0][0]):
        assert ids == [f"row_{i}" for i in O.flat_search(rows, q, 5)[0]]
    assert answers[0] == answers[1] == answers[2]
