#!/usr/bin/env python3
"""Small blocking batches (wdbx_index_search with 2 .. 32 queries) on a small corpus, A/B over the option poll_done in one
process (alternating): p50 wall clock per call.    python tools/probes/small_batch_latency.py [rows=10000] [dim=384]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
k = 10
rng = np.random.default_rng(3)
qs = rng.standard_normal((64, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ix = _native.NativeIndex(d, capacity_rows=n)
ix.fill_synthetic(0xC0FFEE, 0, n, True)
for nq in (1, 2, 8, 16, 32, 48):
    lat = {1: [], 0: []}
    ref = None
    for rep in range(5):
        for poll in (1, 0):
            ix.set_option("poll_done", poll)
            for i in range(3):
                ix.search(qs[:nq], k)
            for i in range(40):
                q = qs[(i % 4) * 4:(i % 4) * 4 + nq] if nq <= 48 else qs[:nq]
                t0 = time.perf_counter()
                r = ix.search(q, k)
                lat[poll].append(time.perf_counter() - t0)
                if i == 0:
                    ref = r[0] if ref is None else ref
                    assert np.array_equal(r[0], ref)
    print(f"{n} x {d}, {nq:2d} queries per call: polled {np.median(lat[1]) * 1e6:6.1f} us   event {np.median(lat[0]) * 1e6:6.1f} us", flush=True)
ix.close()
