#!/usr/bin/env python3
"""Blocking single-query latency through the C ABI (wdbx_index_search, nq = 1) by corpus size, with the query and
result staged through mapped host memory (zero_copy=1, default) or through explicit copies (zero_copy=0)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
d, k = 384, 10
rng = np.random.default_rng(0)
qs = rng.standard_normal((64, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
for n in (10_000, 100_000, 300_000, 1_000_000, 10_000_000):
    ix = _native.NativeIndex(d, capacity_rows=n)
    ix.fill_synthetic(0xC0FFEE, 0, n, True)
    out = {}
    for zc in (1, 0):
        ix.set_option("zero_copy", zc)
        for q in qs[:8]:
            ix.search(q, k)
        lat = []
        for q in qs:
            t0 = time.perf_counter(); ix.search(q, k); lat.append(time.perf_counter() - t0)
        out[f"zero_copy={zc}"] = round(float(np.median(lat)) * 1e6, 1)
    print(n, "p50 us:", out, "path", ix.get_option("last_single_path"), flush=True)
    ix.close()
