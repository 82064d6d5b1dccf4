#!/usr/bin/env python3
"""Generate golden fixtures by DRIVING THE REFERENCE (test infrastructure only).

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONPATH=/root/reference python /root/repo/oracle/gen_golden.py

Writes ``tests/golden/*.json``.  Nothing from the reference's source is copied:
the fixtures are inputs plus the outputs the reference's own code produced.

What can be driven here: ``VectorStore.search/search_async/
_matches_filter/store/batch_store``, ``FaissIndex._normalize_vector``, the
``WDBX`` facade validation and ``get_stats``.  The per-shard index classes need
``hnswlib``/``faiss`` (absent; ordinary ModuleNotFoundError), so at the
``VectorIndex`` seam the harness plugs in a stub that replays CANNED per-shard
candidate lists -- the stub computes nothing, every candidate list is an input
recorded in the fixture.  The distance arithmetic is therefore not pinned here
(see oracle/wdbx_oracle.py header: "parity unpinned" at the arithmetic).
"""

import asyncio
import json
import os
import struct
import sys
import tempfile
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def f32_hex(a):
    return np.asarray(a, dtype=np.float32).tobytes().hex()


class ReplayIndex:
    """Harness stub for the VectorIndex seam: replays canned candidates."""

    canned = {}  # shard number -> list[(id, score)]
    counter = 0

    def __init__(self, vector_dim, index_path, config=None, **kw):
        self.vector_dim = vector_dim
        self.shard = ReplayIndex.counter
        ReplayIndex.counter += 1
        self.added = []

    async def initialize(self):
        pass

    async def shutdown(self):
        pass

    def add(self, vector_id, vector):
        self.added.append(vector_id)
        return True

    async def add_async(self, vector_id, vector):
        return self.add(vector_id, vector)

    def batch_add(self, vectors):
        self.added.extend(vectors.keys())
        return True

    async def batch_add_async(self, vectors):
        return self.batch_add(vectors)

    def search(self, query_vector, limit=10):
        return list(ReplayIndex.canned.get(self.shard, []))[:limit]

    async def search_async(self, query_vector, limit=10):
        return self.search(query_vector, limit)

    def remove(self, vector_id):
        return True

    async def remove_async(self, vector_id):
        return True

    def clear(self):
        self.added = []
        return True

    async def clear_async(self):
        return True

    def optimize(self):
        return True

    async def optimize_async(self):
        return True

    def size(self):
        return len(self.added)

    def get_stats(self):
        return {"type": "replay", "size": self.size(), "dimension": self.vector_dim}


def make_store(num_shards, dim=4):
    import wdbx.core.vector_store as vs

    vs.HNSWIndex = ReplayIndex
    ReplayIndex.counter = 0
    ReplayIndex.canned = {}
    tmp = tempfile.mkdtemp(prefix="wdbx_golden_")
    return vs.VectorStore(vector_dim=dim, data_dir=Path(tmp), num_shards=num_shards)


def merge_cases():
    """(name, shard candidate lists, metadata table, limit, threshold, filter)."""
    meta10 = {f"vec_{i}": {"index": i, "source": "batch_test", "tag": "even" if i % 2 == 0 else "odd"}
              for i in range(10)}
    meta10["vec_7"].pop("tag")  # a row with a missing key
    two = [
        [("vec_5", 0.99), ("vec_3", 0.80), ("vec_1", 0.5), ("vec_9", -0.25)],
        [("vec_4", 0.95), ("vec_2", 0.80), ("vec_0", 0.5), ("vec_8", -0.5), ("vec_7", 0.1), ("vec_6", 0.0)],
    ]
    cases = []

    def add(name, shards, meta, limit=10, threshold=0.0, flt=None):
        cases.append({"name": name, "shards": shards, "metadata": meta, "limit": limit,
                      "threshold": threshold, "filter": flt})

    add("basic_two_shards", two, meta10)
    add("limit_cuts", two, meta10, limit=3)
    add("limit_1", two, meta10, limit=1)
    add("limit_gt_candidates", two, meta10, limit=50)
    add("ties_keep_shard_order", [[("a", 0.5), ("b", 0.5)], [("c", 0.5), ("d", 0.7)], [("e", 0.5)]],
        {k: {"k": k} for k in "abcde"}, limit=4)
    add("threshold_zero_keeps_negatives", two, meta10, threshold=0.0)
    add("threshold_negative_ignored", two, meta10, threshold=-1.0)
    add("threshold_inclusive", two, meta10, threshold=0.80)
    add("threshold_between", two, meta10, threshold=0.81)
    add("threshold_above_all", two, meta10, threshold=1.5)
    add("empty_shards", [[], []], {})
    add("one_empty_shard", [[], [("x", 0.3)]], {"x": {"a": 1}})
    add("missing_metadata_row", [[("ghost", 0.9), ("vec_1", 0.2)]], meta10)
    add("filter_eq", two, meta10, flt={"tag": "even"})
    add("filter_eq_two_keys", two, meta10, flt={"tag": "odd", "source": "batch_test"})
    add("filter_eq_no_match", two, meta10, flt={"source": "nope"})
    add("filter_lt", two, meta10, flt={"index": {"$lt": 3}})
    add("filter_gt", two, meta10, flt={"index": {"$gt": 6}})
    add("filter_gte", two, meta10, flt={"index": {"$gte": 6}})
    add("filter_lte", two, meta10, flt={"index": {"$lte": 3}})
    add("filter_in", two, meta10, flt={"index": {"$in": [1, 4, 9, 42]}})
    add("filter_nin", two, meta10, flt={"index": {"$nin": [1, 4, 9, 42]}})
    add("filter_nin_missing_key_passes", two, meta10, flt={"tag": {"$nin": ["even"]}})
    add("filter_in_missing_key_fails", two, meta10, flt={"tag": {"$in": ["even", "odd"]}})
    add("filter_exists_true", two, meta10, flt={"tag": {"$exists": True}})
    add("filter_exists_false", two, meta10, flt={"tag": {"$exists": False}})
    add("filter_gt_missing_key_fails", two, meta10, flt={"absent": {"$gt": 0}})
    add("filter_multi_operator_first_key_only", two, meta10, flt={"index": {"$gte": 2, "$lt": 5}})
    add("filter_unknown_operator_ignored", two, meta10, flt={"index": {"$regex": "x"}})
    add("filter_dict_value_equality", [[("p", 0.4), ("q", 0.3)]],
        {"p": {"cfg": {"a": 1}}, "q": {"cfg": {"a": 2}}}, flt={"cfg": {"a": 1}})
    add("filter_then_limit_shrinks", two, meta10, limit=2, flt={"tag": "odd"})
    add("filter_and_threshold", two, meta10, threshold=0.5, flt={"index": {"$lt": 4}})
    # post-filter under-return: per-shard lists are already cut to `limit` by the
    # index; the store filters afterwards and can return fewer than `limit`.
    add("post_filter_under_return", [[("vec_5", 0.9), ("vec_3", 0.8)], [("vec_4", 0.7), ("vec_2", 0.6)]],
        meta10, limit=2, flt={"index": {"$lt": 3}})
    eight = [[(f"s{s}_r{r}", round(0.9 - 0.01 * (r * 8 + ((s * 3) % 8)), 4)) for r in range(10)]
             for s in range(8)]
    add("eight_shards_k10", eight, {}, limit=10)
    return cases


def run_merge(cases):
    out = []
    for c in cases:
        store = make_store(len(c["shards"]))
        ReplayIndex.canned = {i: [tuple(x) for x in lst] for i, lst in enumerate(c["shards"])}
        store.metadata = json.loads(json.dumps(c["metadata"]))
        q = [0.1, 0.2, 0.3, 0.4]
        res = store.search(q, limit=c["limit"], threshold=c["threshold"], filter_metadata=c["filter"])
        res_async = asyncio.run(
            store.search_async(q, limit=c["limit"], threshold=c["threshold"], filter_metadata=c["filter"]))
        assert res == res_async, c["name"]
        d = dict(c)
        d["expected"] = [[i, s, m] for i, s, m in res]
        d["async_equal"] = True
        out.append(d)
    return out


def run_filter_rows():
    """Direct _matches_filter truth table."""
    store = make_store(1)
    rows = {
        "r0": {"n": 0, "s": "a", "f": 1.5, "lst": [1, 2], "none": None},
        "r1": {"n": 5, "s": "b", "flag": True},
        "r2": {},
        "r3": {"n": 5.0, "s": "", "flag": False},
    }
    store.metadata = rows
    filters = [
        {"n": 5}, {"n": 0}, {"s": "a"}, {"s": ""}, {"flag": True}, {"flag": False}, {"none": None},
        {"n": {"$gt": 0}}, {"n": {"$gt": 5}}, {"n": {"$gte": 5}}, {"n": {"$lt": 5}}, {"n": {"$lte": 0}},
        {"s": {"$in": ["a", "b"]}}, {"s": {"$nin": ["a"]}}, {"zzz": {"$nin": ["a"]}},
        {"flag": {"$exists": True}}, {"flag": {"$exists": False}}, {"n": {"$exists": 1}},
        {"n": {"$lt": 10, "$gt": 100}}, {"n": {"$gt": 100, "$lt": 10}},
        {"n": {"$bogus": 1}}, {"n": 5, "s": "b"}, {"n": 5, "s": {"$in": ["", "q"]}},
        {"lst": [1, 2]}, {"f": {"$gte": 1.5}}, {},
    ]
    table = []
    for f in filters:
        table.append({"filter": f, "match": {rid: bool(store._matches_filter(rid, f)) for rid in rows}})
    return {"rows": rows, "table": table}


def run_normalize():
    from wdbx.core.indexing import FaissIndex
    from wdbx.utils.data_utils import normalize_vector as du_norm

    rng = np.random.default_rng(1234)
    vecs = [
        np.array([3, 4, 0, 0], np.float32),
        np.zeros(8, np.float32),
        np.array([1e-30, -1e-30, 2e-30], np.float32),
        np.array([1e-23, 1e-23], np.float32),  # squares underflow to subnormal/zero
        np.array([1e20, 1e20, -1e20], np.float32),  # squares overflow in fp32
        np.full(384, 0.1, np.float32),
        np.array([0.1, 0.2, 0.3, 0.4], np.float32),
        np.array([-0.0, 0.0, 5.0], np.float32),
    ]
    for d in (4, 7, 384, 768, 1000):
        vecs.append(rng.standard_normal(d).astype(np.float32))
        vecs.append((rng.random(d, dtype=np.float32) * 2 - 1).astype(np.float32))
    out = []
    for v in vecs:
        with np.errstate(all="ignore"):
            r = FaissIndex._normalize_vector(None, v.astype(np.float32))
        r = np.asarray(r)
        item = {"dim": int(v.shape[0]), "in": f32_hex(v), "out": f32_hex(r), "out_dtype": str(r.dtype)}
        try:
            with np.errstate(all="ignore"):
                r2 = du_norm(v.tolist())
            item["data_utils_equal_within_1e-6"] = bool(
                np.allclose(np.asarray(r2, np.float64), r.astype(np.float64), rtol=1e-6, atol=1e-30,
                            equal_nan=True))
        except Exception as e:  # pragma: no cover
            item["data_utils_error"] = type(e).__name__
        out.append(item)
    return out


def run_facade():
    """Facade validation + stats keys + the F5 attribute/method clash."""
    import wdbx.core.vector_store as vs
    from wdbx import WDBX

    vs.HNSWIndex = ReplayIndex
    ReplayIndex.counter = 0
    ReplayIndex.canned = {}
    tmp = tempfile.mkdtemp(prefix="wdbx_golden_")
    db = WDBX(vector_dimension=4, num_shards=2, data_dir=tmp, enable_plugins=False, log_level="ERROR")
    out = {}
    for name, call in (("vector_search", lambda: db.vector_search([0.1, 0.2, 0.3])),
                       ("vector_search_long", lambda: db.vector_search([0.0] * 9)),
                       ("vector_search_async", lambda: asyncio.run(db.vector_search_async([1.0])))):
        try:
            call()
            out[name] = None
        except Exception as e:
            out[name] = {"type": type(e).__name__, "message": str(e)}
    try:
        db.vector_store([0.1, 0.2, 0.3, 0.4], {"a": 1})
        out["vector_store_call"] = None
    except Exception as e:
        out["vector_store_call"] = {"type": type(e).__name__, "message": str(e)}
    # the reference tests use the attribute form too
    n = db.vector_store.batch_store({f"vec_{i}": [i / 10, (i + 1) / 10, (i + 2) / 10, (i + 3) / 10]
                                     for i in range(5)}, {f"vec_{i}": {"index": i} for i in range(5)})
    stats = db.get_stats()
    out["batch_store_return"] = n
    out["stats_keys"] = sorted(stats.keys())
    out["stats_index_entry_keys"] = sorted(stats["indices"][0].keys())
    out["stats_values"] = {k: stats[k] for k in ("vector_dimension", "num_shards", "total_vectors",
                                                 "vector_count", "metadata_count", "index_type",
                                                 "vector_dim", "use_gpu", "gpu_enabled",
                                                 "plugins_enabled", "plugins_loaded",
                                                 "distributed_enabled")}
    out["n_indices"] = len(stats["indices"])
    out["get_vector"] = db.get_vector("vec_3")
    out["get_vector_missing"] = db.get_vector("nope")
    out["delete_missing"] = db.delete_vector("nope")
    out["update_missing"] = db.update_metadata("nope", {"x": 1})
    out["delete_existing"] = db.delete_vector("vec_3")
    out["count_after_delete"] = db.count_vectors()
    out["clear_return"] = db.clear()
    out["count_after_clear"] = db.count_vectors()
    return out


def run_config():
    from wdbx.core.config import WDBXConfig

    os.environ["WDBX_GOLDEN_ENV"] = "[1, 2]"
    os.environ["WDBX_GOLDEN_BOOL"] = "yes"
    os.environ["WDBX_GOLDEN_FLOAT"] = "2.5"
    cfg = WDBXConfig({"WDBX_INT_OPTION": "42", "WDBX_BOOL_OPTION": "true", "WDBX_LIST_OPTION": "[1, 2, 3]",
                      "HNSW_M": 8})
    out = {
        "defaults": {k: v for k, v in WDBXConfig.DEFAULT_CONFIG.items() if k != "VECTOR_STORE_THREADS"},
        "env_list": cfg.get("WDBX_GOLDEN_ENV"), "env_bool": cfg.get("WDBX_GOLDEN_BOOL"),
        "env_float": cfg.get("WDBX_GOLDEN_FLOAT"),
        "typed_int": cfg.get_typed("WDBX_INT_OPTION", int),
        "typed_bool": cfg.get_typed("WDBX_BOOL_OPTION", bool),
        "typed_list": cfg.get_typed("WDBX_LIST_OPTION", list),
        "typed_default": cfg.get_typed("NONEXISTENT", int, 99),
        "override": cfg.get("HNSW_M"), "source_override": cfg.get_source("HNSW_M"),
        "source_default": cfg.get_source("FAISS_NPROBE"), "source_env": cfg.get_source("WDBX_GOLDEN_ENV"),
    }
    for k in ("WDBX_GOLDEN_ENV", "WDBX_GOLDEN_BOOL", "WDBX_GOLDEN_FLOAT"):
        os.environ.pop(k)
    return out


def main():
    if not Path("/root/reference/wdbx").exists():
        sys.exit("reference not present: fixtures are generated in the build container only")
    sys.path.insert(0, "/root/reference")
    OUT.mkdir(parents=True, exist_ok=True)
    import wdbx

    header = {"generated_by": "oracle/gen_golden.py", "reference_version": wdbx.__version__,
              "numpy": np.__version__}
    for name, payload in (("merge", run_merge(merge_cases())), ("filter_table", run_filter_rows()),
                          ("normalize", run_normalize()), ("facade", run_facade()),
                          ("config", run_config())):
        with open(OUT / f"{name}.json", "w") as f:
            json.dump({"header": header, "data": payload}, f, indent=1)  # key order is data: filters honour the FIRST operator
        print("wrote", OUT / f"{name}.json")


if __name__ == "__main__":
    main()
