#!/usr/bin/env python3
"""N blocking lone searches (wdbx_index_search, nq = 1) on a synthetic corpus, wall clock per call printed: the program
`tools/gpu/kernel_timeline.sh` traces to see the blocking path's kernels and gaps.
    lone_blocking.py <rows> [dim] [calls] [option=value ...]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rng = np.random.default_rng(0)
qs = rng.standard_normal((calls + 8, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ix = _native.NativeIndex(d, capacity_rows=n)
ix.fill_synthetic(0xC0FFEE, 0, n, True)
for o in sys.argv[4:]:
    name, v = o.split("=")
    ix.set_option(name, int(v))
for q in qs[:8]:
    ix.search(q, 10)
lat = []
for q in qs[8:]:
    t0 = time.perf_counter()
    ix.search(q, 10)
    lat.append(time.perf_counter() - t0)
print(f"{n} x {d}: blocking lone search p50 {np.median(lat) * 1e6:.1f} us, min {min(lat) * 1e6:.1f} us, path {ix.get_option('last_single_path')}")
ix.close()
