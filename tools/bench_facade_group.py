#!/usr/bin/env python3
"""`WDBX.vector_search` wall clock through the shard group (one library call: per-shard threads, exchange, merge) against the
direct per-shard path, on ONE GPU: a single shard (`HIP_GROUP_SEARCH="always"`: RCCL 1-rank communicator) and 2 / 4 / 8 shards
sharing the GPU (device-copy exchange), rows per shard as given.  usage: bench_facade_group.py [rows_per_shard]"""
import asyncio, json, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wdbx-py_amd"), str(ROOT / "oracle")]
import wdbx_oracle as O
from wdbx_amd import WDBX

per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
d = 384
queries = [q.tolist() for q in O.synth_rows(O.SEED_QUERY, 0, 220, d)]
out = {}
for shards in (1, 2, 4, 8):
    res = {}
    for mode in ("always", False):
        cfg = {"HIP_GROUP_SEARCH": mode, "HIP_DEVICES": [0], "HIP_CAPACITY_ROWS": per, "HIP_PERSIST_INDEX": False}
        w = WDBX(vector_dimension=d, num_shards=shards, data_dir=tempfile.mkdtemp(), config=cfg, enable_plugins=False, log_level="ERROR")
        w.vector_store.bulk_store_synthetic(per * shards, O.SEED_CORPUS)
        for q in queries[:20]:
            r = w.vector_search(q, limit=10)
        lat = []
        for q in queries[20:]:
            t0 = time.perf_counter(); r = w.vector_search(q, limit=10); lat.append(time.perf_counter() - t0)
        res["group" if mode else "per_shard_calls"] = {"p50_us": round(float(np.percentile(lat, 50)) * 1e6, 1), "p99_us": round(float(np.percentile(lat, 99)) * 1e6, 1),
                                                      "path": w.vector_store.last_search_path, "top": [x[0] for x in r[:3]]}
        asyncio.run(w.shutdown())
    res["same_answer"] = res["group"].pop("top") == res["per_shard_calls"].pop("top")
    out[f"{shards} x {per}"] = res
    print(shards, res, flush=True)
print(json.dumps(out))
