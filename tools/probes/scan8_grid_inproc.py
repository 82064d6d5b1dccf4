#!/usr/bin/env python3
"""One grid per round (blockIdx.y = query) against one launch per query for the u8 selection scan's full passes, alternating
in ONE process on one resident corpus (wall clock of 256 pipelined queries, host-synchronised).  usage: scan8_grid_inproc.py [rows ...]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1_250_000, 2_500_000, 5_000_000, 10_000_000]
d, k, nq = 384, 10, 256
out = {}
for rows in sizes:
    ix = _native.NativeIndex(d, capacity_rows=rows)
    ix.fill_synthetic(0xC0FFEE, 0, rows, True)
    dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
    d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    ix.search_device(dq, nq, k, d_idx, d_score)
    ix.synchronize()
    res = {"one_grid": [], "per_query": []}
    for rnd in range(5):
        for name, v in (("one_grid", 0), ("per_query", 1)):
            ix.set_option("scan8_per_query", v)
            ix.synchronize()
            t0 = time.perf_counter()
            ix.search_device(dq, nq, k, d_idx, d_score)
            ix.synchronize()
            res[name].append(round((time.perf_counter() - t0) / nq * 1e3, 4))
    out[rows] = {n: {"ms_per_query": v, "median": sorted(v)[len(v) // 2]} for n, v in res.items()}
    ix.close()
print(json.dumps(out, indent=1))
