#!/usr/bin/env python3
"""End-to-end latency of the drop-in facade (what a WDBX user sees): WDBX.vector_search on
BASELINE config 1 (10k x 384, cosine, top-10, 1 shard) and on 1M rows, beside the numpy oracle on
the same host."""
import json, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wdbx-py_amd"), str(ROOT / "oracle")]
import wdbx_oracle as O
from wdbx_amd import WDBX

out = {}
for n in (10_000, 1_000_000):
    d = 384
    raw = O.synth_rows(O.SEED_CORPUS, 0, n, d)
    w = WDBX(vector_dimension=d, num_shards=1, data_dir=tempfile.mkdtemp(), enable_plugins=False, log_level="ERROR")
    w.vector_store.bulk_store(raw)
    queries = [q.tolist() for q in O.synth_rows(O.SEED_QUERY, 0, 300, d)]
    for q in queries[:20]:
        w.vector_search(q, limit=10)
    lat = []
    for q in queries:
        t0 = time.perf_counter(); r = w.vector_search(q, limit=10); lat.append(time.perf_counter() - t0)
    rows = O.normalize_rows_fast(raw)
    cl = []
    for q in queries[:100]:
        t0 = time.perf_counter(); O.vector_search([[str(i) for i in range(0)]] and [[]] or [[]], [rows[:0]], q, limit=10) if False else O.flat_search(rows, np.array(q, np.float32), 10); cl.append(time.perf_counter() - t0)
    ix = w.vector_store.indices[0]._native
    qn = np.array(queries[0], np.float32)
    nl = []
    for _ in range(200):
        t0 = time.perf_counter(); ix.search(qn, 10); nl.append(time.perf_counter() - t0)
    out[n] = {"facade_p50_us": np.percentile(lat, 50) * 1e6, "facade_p99_us": np.percentile(lat, 99) * 1e6,
              "c_abi_blocking_p50_us": np.percentile(nl, 50) * 1e6,
              "numpy_oracle_p50_us": np.percentile(cl, 50) * 1e6}
print(json.dumps(out))
