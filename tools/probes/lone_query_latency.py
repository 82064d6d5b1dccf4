#!/usr/bin/env python3
"""Lone-query latency by corpus size, A/B over a library option in ONE process (alternating, so box and clock state are
shared): the blocking call (wdbx_index_search, nq = 1: query and result through mapped host memory, the host ranks the
re-scored keys) and the device-resident call (wdbx_index_search_device with one query + synchronise: everything on the device).

    python tools/probes/lone_query_latency.py [option=merge_fast] [values=1,0] [sizes=10000,300000,1000000,2500000,10000000] [dim=384] [k=10] [metric=cosine|l2]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

opt = sys.argv[1] if len(sys.argv) > 1 else "merge_fast"
values = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,0").split(",")]
sizes = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "10000,300000,1000000,2500000,10000000").split(",")]
d = int(sys.argv[4]) if len(sys.argv) > 4 else 384
k = int(sys.argv[5]) if len(sys.argv) > 5 else 10
l2 = len(sys.argv) > 6 and sys.argv[6] == "l2"
rng = np.random.default_rng(0)
qs = rng.standard_normal((96, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
report = {"option": opt, "values": values, "dim": d, "k": k, "metric": "l2" if l2 else "cosine", "sizes": {}}
for n in sizes:
    ix = _native.NativeIndex(d, metric=_native.METRIC_L2 if l2 else _native.METRIC_COSINE, capacity_rows=n)
    ix.fill_synthetic(0xC0FFEE, 0, n, True)
    dq = ix.device_queries(qs)
    d_idx, d_score = ix.alloc(k * 8), ix.alloc(k * 4)
    lat = {v: {"blocking": [], "resident": []} for v in values}
    ref = None
    for rep in range(6):                      # alternate the option value: every value sees every phase of the box
        for v in values:
            ix.set_option(opt, v)
            for q in qs[:4]:
                ix.search(q, k)
            for i, q in enumerate(qs[:64]):
                t0 = time.perf_counter()
                r = ix.search(q, k)
                lat[v]["blocking"].append(time.perf_counter() - t0)
                if rep == 0:
                    if ref is None:
                        ref = {}
                    ref.setdefault(i, r[0].tolist())
                    assert r[0].tolist() == ref[i], (n, v, i)
            for i in range(64):
                t0 = time.perf_counter()
                ix.search_device(dq, 1, k, d_idx, d_score, query_offset=i)
                ix.synchronize()
                lat[v]["resident"].append(time.perf_counter() - t0)
    rec = {f"{opt}={v}": {kind: round(float(np.median(x)) * 1e6, 1) for kind, x in lat[v].items()} for v in values}
    rec["path"] = ix.get_option("last_single_path")
    report["sizes"][str(n)] = rec
    print(n, rec, flush=True)
    ix.close()
print(json.dumps(report))
