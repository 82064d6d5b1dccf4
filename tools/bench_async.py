#!/usr/bin/env python3
"""Serving-shaped measurement through the drop-in facade: C concurrent `vector_search_async` clients
(one event loop, like the reference's REST server) on a 10M x 384 shard.  Concurrent callers are
coalesced into batched passes, so throughput rises with concurrency while a lone client keeps the
single-scan latency."""
import asyncio, json, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wdbx-py_amd"), str(ROOT / "oracle")]
import wdbx_oracle as O
from wdbx_amd import WDBX

n, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000), 384
w = WDBX(vector_dimension=d, num_shards=1, data_dir=tempfile.mkdtemp(), enable_plugins=False, enable_gpu=True, log_level="ERROR",
         config={"HIP_CAPACITY_ROWS": n, "HIP_PERSIST_INDEX": False})   # a scratch corpus: nothing written at exit
w.vector_store.bulk_store_synthetic(n, O.SEED_CORPUS)         # corpus generated in HBM, implicit ids row_<n>
queries = [q.tolist() for q in O.synth_rows(O.SEED_QUERY, 0, 512, d)]

async def client(cid, stop_at, lat):
    i = cid
    while time.perf_counter() < stop_at:
        t0 = time.perf_counter()
        r = await w.vector_search_async(queries[i % len(queries)], limit=10)
        lat.append(time.perf_counter() - t0)
        assert len(r) == 10
        i += 1

async def run(C, seconds):
    lat = []
    stop_at = time.perf_counter() + seconds
    t0 = time.perf_counter()
    await asyncio.gather(*[client(c, stop_at, lat) for c in range(C)])
    el = time.perf_counter() - t0
    return {"clients": C, "queries_per_s": len(lat) / el, "p50_ms": float(np.percentile(lat, 50) * 1e3),
            "p99_ms": float(np.percentile(lat, 99) * 1e3), "queries": len(lat)}

out = []
for C in (1, 2, 3, 4, 16, 64, 256):
    asyncio.run(run(C, 0.5))
    out.append(asyncio.run(run(C, 3.0)))
    print(out[-1], flush=True)
print(json.dumps({"workload": f"{n} x {d} fp32 cosine top-10, 1 shard, WDBX.vector_search_async", "runs": out}))
