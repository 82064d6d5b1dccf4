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
    # the values the reference reports for the same calls and default flags (index_type aside: "hip" here)
    for key, val in fac["stats_values"].items():
        if key != "index_type":
            assert stats[key] == val, key


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
    # the removed row is gone for good: no zero-score ghost under a str(row) id (the reference's FAISS backend has
    # that flaw, indexing.py:1021); the two remaining rows score 0 and come back in row order
    assert [r[0] for r in res] == ["a", "b"] and all(r[1] == 0.0 for r in res)
    asyncio.run(w.shutdown())


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_removed_rows_never_come_back(temp_dir, metric):
    """Deleted rows are NaN tombstones: never a result, on no path, whatever ``limit`` asks for -- small corpus with
    ``limit >= n`` (fp32 scan), and a corpus large enough for the u8 selection scan and the batched tiles, with
    enough removed rows that they would fill every candidate buffer if they were kept as candidates."""
    from wdbx_amd import WDBX

    cfg = {"HIP_METRIC": metric}
    w = WDBX(vector_dimension=4, num_shards=1, data_dir=temp_dir + "/small", config=cfg, enable_plugins=False)
    for i in range(5):
        w.vector_store([1.0, 0.1 * i, 0, 0], {"i": i}, id=f"v{i}")
    assert w.delete_vector("v2") and w.count_vectors() == 4
    res = w.vector_search([1, 0, 0, 0], limit=5)
    assert sorted(r[0] for r in res) == ["v0", "v1", "v3", "v4"]
    assert asyncio.run(w.vector_search_async([1, 0, 0, 0], limit=50)) == w.vector_search([1, 0, 0, 0], limit=50)
    asyncio.run(w.shutdown())

    d, n = 64, 220_000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    w = WDBX(vector_dimension=d, num_shards=1, data_dir=temp_dir + "/big", config=cfg, enable_plugins=False)
    w.vector_store.bulk_store(raw)
    q = raw[1234] + np.float32(0.01)
    before = w.vector_search(q.tolist(), limit=20)
    assert before[0][0] == "row_1234"
    doomed = [int(r[0][4:]) for r in before[:10:2]] + list(range(100_000, 108_000))  # 5 of the top 10 + 8000 more
    for r in doomed:
        assert w.delete_vector(f"row_{r}")
    assert w.count_vectors() == n - len(doomed)
    rows = O.normalize_rows_fast(raw) if metric == "cosine" else raw.copy()
    rows[doomed] = np.nan
    ix = w.vector_store.indices[0]._native
    for limit in (20, 200):  # list kernels / radix select
        got = w.vector_search(q.tolist(), limit=limit)
        assert ix.get_option("last_single_path") == 2
        if limit == 20:
            assert ix.batch_status(1)["counts"][0] < len(doomed)  # the tombstones took no candidate slots
        o_idx, o_score = O.flat_search(rows, q, limit, O.METRIC_COSINE if metric == "cosine" else O.METRIC_L2)
        assert [g[0] for g in got] == [f"row_{i}" for i in o_idx]
        np.testing.assert_allclose([abs(g[1]) for g in got], np.abs(o_score), atol=1e-5, rtol=1e-5)
    batch = w.vector_search_batch([q.tolist()] * 8, limit=20)  # the batched tiles
    assert all(b == w.vector_search(q.tolist(), limit=20) for b in batch)
    asyncio.run(w.shutdown())


def test_async_row_management_matches_sync(temp_dir):
    """The async twins handle bulk-ingested rows, masks and counters exactly like the sync forms."""
    from wdbx_amd import WDBX

    d, n = 16, 4000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, enable_plugins=False)
    w.vector_store.bulk_store(raw, metadata={f"row_{i}": {"bucket": i % 4} for i in range(n)})

    async def run():
        assert await w.update_metadata_async("row_7", {"bucket": 99}) is True
        assert w.get_vector("row_7")[1] == {"bucket": 99}
        assert await w.delete_vector_async("row_8") is True
        assert await w.delete_vector_async("row_8") is False
        assert await w.update_metadata_async("row_8", {"x": 1}) is False
        assert w.count_vectors() == n - 1
        q = raw[8].tolist()
        flt = {"bucket": 0}
        sync_pre = w.vector_search(q, limit=10, filter_metadata=flt, prefilter=True)
        assert len(sync_pre) == 10 and all(r[2]["bucket"] == 0 for r in sync_pre) and "row_8" not in [r[0] for r in sync_pre]
        assert await w.vector_store.search_async(q, limit=10, filter_metadata=flt, prefilter=True) == sync_pre
        assert await w.vector_search_async(q, limit=10, filter_metadata=flt) == w.vector_search(q, limit=10, filter_metadata=flt)
        # overwriting an implicit bulk id keeps the count; deleting it afterwards removes exactly one row
        assert await w.vector_store_async(raw[3].tolist(), {"bucket": 5}, id="row_9") == "row_9"
        assert w.count_vectors() == n - 1
        assert await w.delete_vector_async("row_9") is True and w.count_vectors() == n - 2

    asyncio.run(run())
    asyncio.run(w.shutdown())


def test_index_files_follow_ingest_and_survive_an_unclean_exit(temp_dir):
    """The shards' row files are appended to as vectors arrive (every HIP_AUTOSAVE_ROWS adds, the reference's
    every-1000-adds cadence, indexing.py:898) and by VECTOR_STORE_SAVE_IMMEDIATELY; after an unclean exit with index
    files OLDER than vectors.pickle every stored vector is searchable again, and bulk ranges that never reached the
    disk are dropped instead of claimed."""
    from wdbx_amd import WDBX

    d, n = 8, 2500
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    cfg = {"HIP_AUTOSAVE_ROWS": 1000}
    w = WDBX(vector_dimension=d, num_shards=1, data_dir=temp_dir, config=cfg, enable_plugins=False)
    for i in range(n):
        w.vector_store(raw[i].tolist(), {"i": i}, id=f"v{i}")
    ix = w.vector_store.indices[0]
    assert ix._saved_rows == 2000 and ix.unsaved()
    rows_file = ix._files()[0]
    assert np.load(rows_file, mmap_mode="r").shape == (2000, d)
    w.vector_store.bulk_store(raw[:300], id_prefix="bulk_")            # HBM only so far
    w.vector_store._save_vectors()                                      # the tables are newer than the index files ...
    w.vector_store._save_metadata()
    expected = w.vector_search(raw[2400].tolist(), limit=5)
    del w, ix                                                           # ... and the process "dies" without shutdown()

    w2 = WDBX(vector_dimension=d, num_shards=1, data_dir=temp_dir, config=cfg, enable_plugins=False)
    assert w2.count_vectors() == n                                      # 500 vectors re-added from the id -> vector table
    assert w2.vector_store._bulk_ranges == [] and w2.get_vector("bulk_5") is None
    got = w2.vector_search(raw[2400].tolist(), limit=5)
    assert got[0][0] == "v2400" and got[0][1] > 0.9999 and expected[0][0] == "v2400"
    # incremental saves: replace one row, remove one, append some, and everything comes back
    w2.vector_store(raw[7].tolist(), {"i": -1}, id="v100")
    assert w2.delete_vector("v200")
    w2.vector_store(raw[9].tolist(), {"i": -2}, id="fresh")
    asyncio.run(w2.shutdown())
    w3 = WDBX(vector_dimension=d, num_shards=1, data_dir=temp_dir, config=cfg, enable_plugins=False)
    assert w3.count_vectors() == n and w3.get_vector("v200") is None
    assert w3.vector_search(raw[7].tolist(), limit=2)[0][1] > 0.9999
    assert {r[0] for r in w3.vector_search(raw[7].tolist(), limit=2)} == {"v7", "v100"}
    assert {r[0] for r in w3.vector_search(raw[9].tolist(), limit=2)} == {"v9", "fresh"}
    assert "v200" not in [r[0] for r in w3.vector_search(raw[200].tolist(), limit=10)]
    asyncio.run(w3.shutdown())


def test_save_immediately_covers_the_index_files(temp_dir):
    from wdbx_amd import WDBX

    cfg = {"VECTOR_STORE_SAVE_IMMEDIATELY": True}
    w = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir, config=cfg, enable_plugins=False)
    ids = [w.vector_store([1, i, 0, 0], {"i": i}) for i in range(6)]
    assert all(not ix.unsaved() for ix in w.vector_store.indices)
    del w  # no shutdown
    w2 = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir, config=cfg, enable_plugins=False)
    assert w2.count_vectors() == 6 and sum(ix.next_index for ix in w2.vector_store.indices) == 6
    assert w2.vector_search([1, 3, 0, 0], limit=1)[0][0] == ids[3]
    asyncio.run(w2.shutdown())


def test_enable_gpu_flag_is_plumbed_like_the_reference(temp_dir, caplog):
    """wdbx.py:44,124 -> vector_store.py:128 -> indexing.py:741-748: default False; the flag is recorded and reported
    as given; this backend serves from the GPU either way and says so once when the flag is False."""
    import logging

    from wdbx_amd import WDBX
    from wdbx_amd.indexing import HipFlatIndex

    HipFlatIndex._warned_no_cpu_path = False
    with caplog.at_level(logging.INFO):
        w = WDBX(vector_dimension=4, num_shards=2, data_dir=temp_dir + "/a", enable_plugins=False)
    assert w.enable_gpu is False and w.vector_store.use_gpu is False
    assert sum("no CPU path" in r.getMessage() for r in caplog.records) == 1
    stats = w.get_stats()
    assert stats["gpu_enabled"] is False and stats["use_gpu"] is False
    assert stats["indices"][0]["stats"]["gpu_enabled"] is False and "device" in stats["indices"][0]["stats"]
    vid = w.vector_store([0.1, 0.2, 0.3, 0.4], {})
    assert w.vector_search([0.1, 0.2, 0.3, 0.4], limit=1)[0][0] == vid  # served by the GPU all the same
    asyncio.run(w.shutdown())
    caplog.clear()
    with caplog.at_level(logging.INFO):
        w = WDBX(vector_dimension=4, num_shards=1, data_dir=temp_dir + "/b", enable_plugins=False, enable_gpu=True)
    assert any("GPU acceleration" in r.getMessage() for r in caplog.records)
    assert w.get_stats()["gpu_enabled"] is True and w.get_stats()["indices"][0]["stats"]["gpu_enabled"] is True
    asyncio.run(w.shutdown())


def _group_case(temp_dir, shards, devices, mode):
    from wdbx_amd import WDBX

    d, n = 96, 70_000 * shards
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    meta = {f"row_{i}": {"bucket": i % 3} for i in range(0, n, 7)}
    outs = []
    for group in (mode, False):
        cfg = {"HIP_GROUP_SEARCH": group, "HIP_DEVICES": devices}
        w = WDBX(vector_dimension=d, num_shards=shards, data_dir=f"{temp_dir}/{group}", config=cfg, enable_plugins=False)
        w.vector_store.bulk_store(raw, metadata=meta)
        w.vector_store.batch_store({"dup_a": raw[5].tolist(), "dup_b": raw[5].tolist()}, {"dup_a": {"bucket": 0}})
        res = []
        for q in list(O.synth_rows(O.SEED_QUERY, 0, 3, d)) + [raw[5]]:
            for kw in (dict(limit=10), dict(limit=3, threshold=0.2), dict(limit=25, filter_metadata={"bucket": 0}),
                       dict(limit=300)):
                res.append(w.vector_search(q.tolist(), **kw))
                want = "threads" if not group else "rccl_group" if len(set(devices)) == len(devices) else "copy_group"
                assert w.vector_store.last_search_path == want
                assert asyncio.run(w.vector_search_async(q.tolist(), **kw)) == res[-1]
            res.append(w.vector_search_batch([q.tolist(), raw[9].tolist()], limit=7))
        if group:
            info = w.vector_store._group.info()
            rccl = len(set(devices)) == len(devices)  # one shard per device: RCCL ranks; shared devices: device copies
            assert info["shards"] == shards and info["rccl_nranks"] == (shards if rccl else 0)
        outs.append(res)
        asyncio.run(w.shutdown())
    return outs


def test_shard_group_path_equals_the_per_shard_path_on_one_device(temp_dir):
    """SURVEY 8b "who calls it": VectorStore.search through ONE library call (wdbx_group_attach +
    wdbx_group_search_merged: local scans, ncclAllGather of the key lists, merge kernel) returns exactly the
    (id, score, metadata) lists of the per-shard calls + Python merge.  One device: a 1-rank communicator
    (``HIP_GROUP_SEARCH="always"``; by default a single shard has nothing to gather)."""
    with_group, without = _group_case(temp_dir, 1, [0], "always")
    assert with_group == without


def test_shard_group_path_for_shards_sharing_one_device(temp_dir):
    """Three shards on ONE GPU: the group (device-copy exchange, a host thread per shard) is the default fan-out
    (``HIP_GROUP_SEARCH="auto"``) and returns exactly what the per-shard calls + Python merge return."""
    from wdbx_amd import WDBX

    with_group, without = _group_case(temp_dir, 3, [0, 0, 0], True)
    assert with_group == without
    w = WDBX(vector_dimension=8, num_shards=2, data_dir=temp_dir + "/auto", enable_plugins=False)
    w.vector_store.batch_store({f"v{i}": [float(i == j) for j in range(8)] for i in range(8)})
    assert w.vector_search([1.0] + [0.0] * 7, limit=1)[0][0] == "v0"
    assert w.vector_store.last_search_path == "copy_group"
    asyncio.run(w.shutdown())


def test_shard_group_over_all_visible_gpus(temp_dir):
    """The same on every GPU of the box (one shard per device: ncclCommInitAll with S > 1 ranks, all-gather over
    xGMI).  Skipped on a one-GPU box; an 8-GPU driver run exercises it."""
    from wdbx_amd import _native

    ndev = _native.device_count()
    if ndev < 2:
        pytest.skip("needs at least 2 GPUs")
    with_group, without = _group_case(temp_dir, ndev, list(range(ndev)), True)
    assert with_group == without


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_optimize_compacts_removed_rows_and_keeps_every_answer(temp_dir, metric):
    """VectorStore.optimize() / HipFlatIndex.optimize() (the reference's rebuild hook, indexing.py:1124-1149): 30 % of
    200 k bulk rows and some explicitly named rows are deleted, then compacted away on the device.  Same answers before
    and after (ids, scores, order), fewer stored rows, ids <-> rows still resolve, further ingest and deletes work, and
    the compacted shard survives a save / reopen."""
    from wdbx_amd import WDBX

    d, n = 96, 200_000
    rng = np.random.default_rng(41)
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    cfg = {"HIP_METRIC": metric, "HIP_CAPACITY_ROWS": n // 2}
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, config=cfg, enable_plugins=False)
    vs = w.vector_store
    vs.bulk_store(raw, metadata={f"row_{i}": {"bucket": i % 5} for i in range(0, n, 11)})
    named = {f"named_{i}": raw[i * 7].tolist() for i in range(300)}
    vs.batch_store(named, {v: {"bucket": 9} for v in named})
    vs._save_now()  # (the row files exist before the compaction: the save after it must NOT write into them -- a new generation)
    dead = set(rng.choice(n, int(0.3 * n), replace=False).tolist())
    for i in dead:
        assert w.delete_vector(f"row_{i}")
    for i in range(0, 300, 3):
        assert w.delete_vector(f"named_{i}")
    queries = [raw[i] for i in (5, 77_777, 150_001)] + list(O.synth_rows(O.SEED_QUERY, 0, 5, d))
    kws = (dict(limit=10), dict(limit=40, filter_metadata={"bucket": 9}), dict(limit=300), dict(limit=5, filter_metadata={"bucket": 2}, prefilter=True))
    before = [w.vector_search(q.tolist(), **kw) for q in queries for kw in kws]
    batch_before = w.vector_search_batch([q.tolist() for q in queries], limit=10)
    stored_before = [ix.next_index for ix in vs.indices]
    count_before = vs.count()
    assert vs.optimize()
    stored_after = [ix.next_index for ix in vs.indices]
    assert sum(stored_after) == count_before == vs.count() and all(a < b for a, b in zip(stored_after, stored_before))
    assert [ix._native.size() for ix in vs.indices] == stored_after
    assert [w.vector_search(q.tolist(), **kw) for q in queries for kw in kws] == before
    # (a batch may reach the same ranking through another kernel path once the dead rows are gone: ids, order and
    # metadata identical, scores equal to fp32 summation-order noise)
    batch_after = w.vector_search_batch([q.tolist() for q in queries], limit=10)
    for got, want in zip(batch_after, batch_before):
        assert [(v, m) for v, _, m in got] == [(v, m) for v, _, m in want]
        np.testing.assert_allclose([s for _, s, _ in got], [s for _, s, _ in want], atol=1e-6, rtol=1e-6)
    # no dead id came back, live ids resolve to their (normalised) vectors
    alive = next(i for i in range(n) if i not in dead)
    assert w.get_vector(f"row_{next(iter(dead))}") is None and w.get_vector("named_0") is None
    got = np.asarray(w.get_vector(f"row_{alive}")[0], np.float32)
    want = raw[alive] / np.linalg.norm(raw[alive]) if metric == "cosine" else raw[alive]
    np.testing.assert_allclose(got, want, atol=1e-6)
    assert w.get_vector("named_1")[1] == {"bucket": 9}
    # the compacted store keeps working: ingest, delete, a second optimize with nothing to do, reopen
    vid = w.vector_store(raw[3].tolist(), {"bucket": 7}, id="after")
    assert w.vector_search(raw[3].tolist(), limit=1, filter_metadata={"bucket": 7})[0][0] == vid
    assert w.delete_vector(f"row_{alive}") and vs.optimize()
    again = [w.vector_search(q.tolist(), limit=10) for q in queries]
    for s in range(2):  # the compaction has not touched the committed row files: the mapping on disk still describes them
        assert sorted(p.name for p in (Path(temp_dir) / f"shard_{s}").glob("index.rows*")) == ["index.rows.npy"]
    asyncio.run(w.shutdown())
    for s in range(2):  # the save after it committed a new generation with its mapping and swept the old one
        assert sorted(p.name for p in (Path(temp_dir) / f"shard_{s}").glob("index.rows*")) == ["index.rows.g1.npy"]
    w2 = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, config=cfg, enable_plugins=False)
    assert w2.vector_store.count() == count_before  # (+ "after", - one more deleted row)
    assert [w2.vector_search(q.tolist(), limit=10) for q in queries] == again
    asyncio.run(w2.shutdown())


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


def test_threaded_synchronous_callers_are_coalesced_onto_the_batched_kernel(temp_dir):
    """8 threads calling the synchronous ``vector_search`` (the reference serves ``search`` from 4-worker pools,
    indexing.py:692, :1045-1048): callers that find a search in flight queue up and are answered together by the next
    leader's batched pass.  Every answer equals the single-thread answer of the same query with the same limit / threshold /
    filter; the shards saw far fewer scan launches than queries; ``SYNC_COALESCE=False`` gives one call per query again."""
    import threading

    from wdbx_amd import WDBX

    d, n = 128, 200_000
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, enable_plugins=False)
    w.vector_store.bulk_store(O.synth_rows(O.SEED_CORPUS, 0, n, d), metadata={f"row_{i}": {"bucket": i % 3} for i in range(0, n, 5)})
    queries = [q.tolist() for q in O.synth_rows(O.SEED_QUERY, 0, 48, d)]
    kws = [dict(limit=1 + (i % 12)) if i % 4 else dict(limit=8, threshold=0.05, filter_metadata={"bucket": 1}) for i in range(48)]
    want = [w.vector_search(q, **kw) for q, kw in zip(queries, kws)]
    natives = [ix._native for ix in w.vector_store.indices]
    for nat in natives:
        nat.profile(True)
        nat.profile_read(), nat.profile_read_gemm()
    got, errors = {}, []

    def worker(t):
        try:
            for rep in range(3):
                for i in range(t, 48, 8):
                    got[(rep, i)] = w.vector_search(queries[i], **kws[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errors and len(got) == 144
    for (rep, i), res in got.items():
        assert [(v, m) for v, _, m in res] == [(v, m) for v, _, m in want[i]], (rep, i)
        np.testing.assert_allclose([s for _, s, _ in res], [s for _, s, _ in want[i]], atol=2e-6, rtol=0)
    gemm = sum(nat.profile_read_gemm()["gemm_launches"] for nat in natives)
    assert gemm >= 2                                                   # batched passes ran
    assert not w.vector_store._sync_busy and not w.vector_store._sync_pending
    with pytest.raises(ValueError, match="dimension mismatch"):
        w.vector_search([0.0] * 3)
    w.vector_store._sync_coalesce = False
    assert w.vector_search(queries[5], **kws[5]) == want[5]
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
    # a batch: one pass on the i8 tiles, same answers as the lone searches
    batch = w.vector_search_batch([q.tolist() for q in queries], limit=3)
    for q, res in zip(queries, batch):
        assert [r[0] for r in res] == [r[0] for r in w.vector_search(q.tolist(), limit=3)]
    assert native_ix.get_option("last_gemm_family") == 3
    asyncio.run(w.shutdown())


def test_shadow_copies_can_be_switched_off_by_config(temp_dir):
    """HIP_U8_SHADOW / HIP_BF16_SHADOW = False: no extra device memory is taken, the answers stay the same."""
    from wdbx_amd import WDBX

    d, n = 128, 250_000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    rows = O.normalize_rows_fast(raw)
    queries = O.synth_rows(O.SEED_QUERY, 0, 6, d)
    answers = []
    for cfg, single_path, family in (({}, 2, 3), ({"HIP_U8_SHADOW": False}, 1, 3), ({"HIP_U8_SHADOW": False, "HIP_BF16_SHADOW": False}, 0, 1)):
        w = WDBX(vector_dimension=d, num_shards=1, data_dir=f"{temp_dir}/{single_path}{family}", config=cfg, enable_plugins=False)
        w.vector_store.bulk_store(raw)
        ix = w.vector_store.indices[0]._native
        lone = [w.vector_search(q.tolist(), limit=5) for q in queries[:2]]
        assert ix.get_option("last_single_path") == single_path
        batch = w.vector_search_batch([q.tolist() for q in queries], limit=5)
        assert ix.get_option("last_gemm_family") == family
        assert (ix.get_option("shadow8_bytes") > 0) == (single_path == 2)
        assert (ix.get_option("shadow_bytes") > 0) == (single_path == 1)     # the bf16 copy: only the bf16 single-query path needs it
        assert (ix.get_option("shadowg_bytes") > 0) == (family == 3)
        answers.append(([[r[0] for r in res] for res in lone], [[r[0] for r in res] for res in batch]))
        asyncio.run(w.shutdown())
    for q, ids in zip(queries[:2], answers[0][0]):
        assert ids == [f"row_{i}" for i in O.flat_search(rows, q, 5)[0]]
    assert answers[0] == answers[1] == answers[2]


def test_rest_search_handlers_against_the_oracle(temp_dir):
    """the reference's ``POST /vectors/search`` body and response shapes (api/server.py:109-113,
    :141-152) over the HIP backend, plus the batch form; hits equal the oracle's on the same shard contents, and 32
    concurrent requests (what a server produces) are answered from coalesced batched passes."""
    from wdbx_amd import WDBX
    from wdbx_amd.api import search_batch_endpoint, search_endpoint

    d, n = 128, 150_000
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    rows = O.normalize_rows_fast(raw)
    meta = {f"row_{i}": {"bucket": i % 5} for i in range(0, n, 3)}
    w = WDBX(vector_dimension=d, num_shards=2, data_dir=temp_dir, enable_plugins=False)
    w.vector_store.bulk_store(raw, metadata=meta)
    queries = O.synth_rows(O.SEED_QUERY, 0, 32, d)

    def expect(q, limit, flt=None):
        o_idx, o_score = O.flat_search(rows, q, limit)
        hits = [(f"row_{i}", float(s), meta.get(f"row_{i}", {})) for i, s in zip(o_idx, o_score)]
        return [h for h in hits if flt is None or O.matches_filter(h[2], flt)]

    async def run():
        out = await search_endpoint(w, {"query_vector": queries[0].tolist(), "limit": 7})
        exp = expect(queries[0], 7)
        assert [r["vector_id"] for r in out["results"]] == [e[0] for e in exp]
        np.testing.assert_allclose([r["similarity"] for r in out["results"]], [e[1] for e in exp], atol=1e-5, rtol=0)
        assert [r["metadata"] for r in out["results"]] == [e[2] for e in exp]
        flt = {"bucket": {"$in": [0, 1]}}
        out = await search_endpoint(w, {"query_vector": queries[1].tolist(), "limit": 40, "filter_metadata": flt})
        # post-filter over the union of the two shards' top-40 lists (reference semantics): a superset of the global top-40's hits
        exp_ids = [e[0] for e in expect(queries[1], 40, flt)]
        got_ids = [r["vector_id"] for r in out["results"]]
        assert got_ids[: len(exp_ids)] == exp_ids and all(r["metadata"]["bucket"] in (0, 1) for r in out["results"])
        outs = await asyncio.gather(*[search_endpoint(w, {"query_vector": q.tolist(), "limit": 5}) for q in queries])
        for q, o in zip(queries, outs):
            assert [r["vector_id"] for r in o["results"]] == [e[0] for e in expect(q, 5)]
        batch = await search_batch_endpoint(w, {"query_vectors": [q.tolist() for q in queries[:9]], "limit": 5})
        assert batch["results"] == [o["results"] for o in outs[:9]]
        with pytest.raises(ValueError):
            await search_endpoint(w, {"query_vector": [0.0] * (d - 1)})

    asyncio.run(run())
    asyncio.run(w.shutdown())
