#!/usr/bin/env python3
"""Serving-shaped measurements through the drop-in facade on one shard:

    python tools/bench_async.py [rows] [--threads | --async] [--seconds S]

  --async    (default) C concurrent `vector_search_async` clients on one event loop, like the reference's REST server
             (api/server.py:143): coalesced into batched passes, so throughput rises with concurrency while a lone client
             keeps the single-scan latency;
  --threads  T threads calling the SYNCHRONOUS `WDBX.vector_search` (the reference's per-index pools call search from 4
             workers, indexing.py:692, :1045-1048): callers that arrive while a search is in flight are answered together by
             the next leader (VectorStore._search_coalesced); `SYNC_COALESCE=False` beside it = one corpus scan per call
             behind the handle's mutex.  Every thread's answers are compared with the single-thread answers of the same queries.
"""
import argparse
import asyncio
import json
import sys
import tempfile
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wdbx-py_amd"), str(ROOT / "oracle")]
import wdbx_oracle as O  # noqa: E402  (query generator only)
from wdbx_amd import WDBX  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("rows", nargs="?", type=int, default=10_000_000)
ap.add_argument("--threads", action="store_true")
ap.add_argument("--async", dest="use_async", action="store_true")
ap.add_argument("--seconds", type=float, default=3.0)
args = ap.parse_args()
n, d = args.rows, 384


def make(coalesce=True):
    w = WDBX(vector_dimension=d, num_shards=1, data_dir=tempfile.mkdtemp(), enable_plugins=False, enable_gpu=True, log_level="ERROR",
             config={"HIP_CAPACITY_ROWS": n, "HIP_PERSIST_INDEX": False, "SYNC_COALESCE": coalesce})   # a scratch corpus
    w.vector_store.bulk_store_synthetic(n, O.SEED_CORPUS)         # corpus generated in HBM, implicit ids row_<n>
    return w


queries = [q.tolist() for q in O.synth_rows(O.SEED_QUERY, 0, 512, d)]


# ------------------------------------------------------------------------------------------------ asyncio clients
async def client(w, cid, stop_at, lat):
    i = cid
    while time.perf_counter() < stop_at:
        t0 = time.perf_counter()
        r = await w.vector_search_async(queries[i % len(queries)], limit=10)
        lat.append(time.perf_counter() - t0)
        assert len(r) == 10
        i += 1


async def run_async(w, C, seconds):
    lat = []
    stop_at = time.perf_counter() + seconds
    t0 = time.perf_counter()
    await asyncio.gather(*[client(w, c, stop_at, lat) for c in range(C)])
    el = time.perf_counter() - t0
    return {"clients": C, "queries_per_s": len(lat) / el, "p50_ms": float(np.percentile(lat, 50) * 1e3),
            "p99_ms": float(np.percentile(lat, 99) * 1e3), "queries": len(lat)}


# ------------------------------------------------------------------------------------------------ synchronous threads
def run_threads(w, T, seconds, expect):
    lat, wrong = [[] for _ in range(T)], [0] * T
    stop_at = time.perf_counter() + seconds
    start = threading.Barrier(T)

    def worker(t):
        i = t * 7
        start.wait()
        while time.perf_counter() < stop_at:
            qi = i % len(queries)
            t0 = time.perf_counter()
            r = w.vector_search(queries[qi], limit=10)
            lat[t].append(time.perf_counter() - t0)
            if [x[0] for x in r] != expect[qi]:
                wrong[t] += 1
            i += 1

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    el = time.perf_counter() - t0
    flat = [x for per in lat for x in per]
    return {"threads": T, "queries_per_s": len(flat) / el, "p50_ms": float(np.percentile(flat, 50) * 1e3),
            "p99_ms": float(np.percentile(flat, 99) * 1e3), "queries": len(flat), "answers_differing_from_single_thread": int(sum(wrong))}


if args.threads:
    out = {"workload": f"{n} x {d} fp32 cosine top-10, 1 shard, WDBX.vector_search from T threads", "coalesced": [], "one_scan_per_call": []}
    for coalesce, key in ((True, "coalesced"), (False, "one_scan_per_call")):
        w = make(coalesce)
        expect = [[x[0] for x in w.vector_search(q, limit=10)] for q in queries]   # single-thread answers
        for T in (1, 2, 4, 8, 16):
            run_threads(w, T, 0.3, expect)
            out[key].append(run_threads(w, T, args.seconds, expect))
            print(key, out[key][-1], flush=True)
        asyncio.run(w.shutdown())
    base = out["coalesced"][0]["queries_per_s"]
    out["speedup_8_threads_over_1"] = next(r["queries_per_s"] for r in out["coalesced"] if r["threads"] == 8) / base
    print(json.dumps(out))
else:
    w = make()
    runs = []
    for C in (1, 2, 3, 4, 16, 64, 256):
        asyncio.run(run_async(w, C, 0.5))
        runs.append(asyncio.run(run_async(w, C, args.seconds)))
        print(runs[-1], flush=True)
    print(json.dumps({"workload": f"{n} x {d} fp32 cosine top-10, 1 shard, WDBX.vector_search_async", "runs": runs}))
