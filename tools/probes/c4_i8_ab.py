#!/usr/bin/env python3
"""A/B of gemm_i8_kernel variants in ONE process (interleaved rounds): ms per 256-query batch of the tile launches (HIP events)
for each value of the library option `gemm8_variant`.  usage: c4_i8_ab.py [rows] [dim] [variants, e.g. 0,1] [rounds]"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
nq, k = 256, 10
ix = _native.NativeIndex(dim, capacity_rows=rows)
ix.fill_synthetic(0xC0FFEE, 0, rows, True)
dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
res = {v: [] for v in variants}
ref = None
ix.profile(True)
for r in range(rounds + 1):
    for v in variants:
        ix.set_option("gemm8_variant", v)
        ix.profile_read_gemm()
        for _ in range(5):
            ix.search_batch_device(dq, nq, k, d_idx, d_score)
        g = ix.profile_read_gemm()
        got = d_idx.download(np.int64, (nq, k))
        if v not in (8, 10, 11, 30, 31, 32):  # (timing-only forms with wrong answers)
            ref = got if ref is None else ref
            assert np.array_equal(got, ref), f"variant {v} changed the answer"
        if r:  # round 0 = warm-up
            res[v].append(g["gemm_ms"] / 5)
out = {f"variant_{v}": {"median_ms": float(np.median(t)), "min_ms": float(np.min(t)), "all": [round(x, 4) for x in t]} for v, t in res.items()}
print(json.dumps({"rows": rows, "dim": dim, "family": ix.get_option("last_gemm_family"), **out}, indent=1))
