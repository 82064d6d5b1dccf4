#!/usr/bin/env python3
"""Rounds of resident queries on the fp32 scan (corpora below the selection paths' sizes), A/B over option scan_one_grid in one
process (alternating): queries/s of 512 queries enqueued as one call, and ms per blocking call of 8 / 32 host queries.
    python tools/probes/one_grid_ab.py [rows ...]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

d, k, nq = 384, 10, 512
sizes = [int(v) for v in sys.argv[1:]] or [10_000, 50_000, 100_000, 190_000]
rng = np.random.default_rng(4)
hq = rng.standard_normal((32, d)).astype(np.float32)
hq /= np.linalg.norm(hq, axis=1, keepdims=True)
for n in sizes:
    ix = _native.NativeIndex(d, capacity_rows=n)
    ix.fill_synthetic(0xC0FFEE, 0, n, True)
    ix.set_option("scan_shadow", 0)   # (the plain fp32 scan at every size)
    dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
    d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    res, ref = {1: [], 0: []}, None
    blk = {1: {8: [], 32: []}, 0: {8: [], 32: []}}
    for rep in range(4):
        for og in (1, 0):
            ix.set_option("scan_one_grid", og)
            ix.search_device(dq, nq, k, d_idx, d_score)
            ix.synchronize()
            t0 = time.perf_counter()
            ix.search_device(dq, nq, k, d_idx, d_score)
            ix.synchronize()
            res[og].append(nq / (time.perf_counter() - t0))
            got = d_idx.download(np.int64, (nq, k))
            ref = got if ref is None else ref
            assert np.array_equal(got, ref)
            for b in (8, 32):
                for _ in range(12):
                    t0 = time.perf_counter()
                    ix.search(hq[:b], k)
                    blk[og][b].append(time.perf_counter() - t0)
    print(f"{n:8d} x {d}: one grid per round {np.median(res[1]):9.0f} q/s, a launch per query {np.median(res[0]):9.0f} q/s;  "
          f"blocking 8 queries {np.median(blk[1][8]) * 1e6:6.1f} / {np.median(blk[0][8]) * 1e6:6.1f} us, "
          f"32 queries {np.median(blk[1][32]) * 1e6:6.1f} / {np.median(blk[0][32]) * 1e6:6.1f} us", flush=True)
    ix.close()
