#!/usr/bin/env python3
"""Scan + merge time as a function of k on the 10M x 384 corpus (large-k callers: the reference's
visualisation asks for up to 1000)."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
rows, dim = (int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000), 384
ix = _native.NativeIndex(dim, capacity_rows=rows)
ix.set_option("scan_shadow", 0)  # this tool measures the fp32 scan kernel
ix.fill_synthetic(0xC0FFEE, 0, rows, True)
nq = 8
dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
out = {}
ix.set_option("exchange_batch", 1)
for k in (1, 10, 64, 65, 100, 256, 1000, 2048):
    d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    ix.profile(True)
    ix.search_device(dq, 2, k, d_idx, d_score); ix.synchronize(); ix.profile_read()
    t0 = time.perf_counter()
    ix.search_device(dq, nq, k, d_idx, d_score); ix.synchronize()
    wall = (time.perf_counter() - t0) / nq * 1e3
    p = ix.profile_read()
    out[k] = {"scan_ms": round(p["scan_ms"] / p["scan_launches"], 4), "merge_ms": round(p["merge_ms"] / p["merge_launches"], 4),
              "wall_ms_per_query": round(wall, 4)}
    print(k, out[k], flush=True)
print(json.dumps(out))
