#!/usr/bin/env python3
"""Kernel-variant sweep on one resident corpus (experiments; interleaved rounds in ONE process,
cdna_hip_programming.md rule 24).  usage: sweep.py [rows] [dim] [k] [metric]"""
import itertools
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
metric = int(sys.argv[4]) if len(sys.argv) > 4 else 0
grid = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {
    "scan_lanes": [8, 16, 32], "scan_nt": [1], "scan_blocked": [0, 1], "scan_blocks": [0, 512, 1024, 2048, 4096]}
rounds, per = 3, 30

ix = _native.NativeIndex(dim, metric=metric, capacity_rows=rows)
ix.set_option("scan_shadow", 0)  # this tool measures the fp32 scan kernel
ix.fill_synthetic(0xC0FFEE, 0, rows, True)
dq = ix.device_queries_synthetic(0xBEEF, 0, per, True)
d_idx, d_score = ix.alloc(per * k * 8), ix.alloc(per * k * 4)
names = list(grid)
combos = list(itertools.product(*[grid[n] for n in names]))
res = {c: [] for c in combos}
ix.profile(True)
for r in range(rounds):
    for c in combos:
        for n, v in zip(names, c):
            ix.set_option(n, v)
        ix.search_device(dq, 3, k, d_idx, d_score)
        ix.synchronize()
        ix.profile_read()
        ix.search_device(dq, per, k, d_idx, d_score)
        ix.synchronize()
        p = ix.profile_read()
        res[c].append(p["scan_ms"] / p["scan_launches"])
bytes_ = rows * dim * 4
out = []
for c in combos:
    med, mn = float(np.median(res[c])), float(np.min(res[c]))
    out.append((med, c, mn))
out.sort()
for med, c, mn in out:
    print(dict(zip(names, c)), f"median {med:.4f} ms  min {mn:.4f} ms  {bytes_ / med / 1e6:.0f} GB/s", flush=True)
