#!/usr/bin/env python3
"""Where does the u8 selection scan start to beat the fp32 scan?  Pipelined single queries and single-client
latency for small corpora (d = 384, k = 10)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
d, k, nq = 384, 10, 64
for n in (65_536, 100_000, 150_000, 262_144, 524_288, 1_048_576):
    ix = _native.NativeIndex(d, capacity_rows=n)
    ix.set_option("single_min_rows", 0)  # measure the selection scan for lone queries at every size (the default routes them to the fp32 scan below 262 144 rows)
    ix.fill_synthetic(0xC0FFEE, 0, n, True)
    dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
    d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    row = {}
    for name, opt in (("u8", 2), ("fp32", 0)):
        ix.set_option("scan_shadow", opt)
        ix.search_device(dq, nq, k, d_idx, d_score); ix.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.search_device(dq, nq, k, d_idx, d_score)
        ix.synchronize()
        pipe = (time.perf_counter() - t0) / (5 * nq) * 1e6
        lat = []
        for i in range(40):
            t1 = time.perf_counter()
            ix.search_device(dq, 1, k, d_idx, d_score, query_offset=i); ix.synchronize()
            lat.append(time.perf_counter() - t1)
        row[name] = (round(pipe, 1), round(float(np.median(lat)) * 1e6, 1))
    print(n, "us/query pipelined, single-client p50:", row, flush=True)
    ix.close()
