#!/usr/bin/env python3
"""Timing-only ablation of scan8_kernel's phase 1 (option scan8_ablate=1: a quarter of the convert + fma work per 16 bytes,
wrong answers): how much of the gap to the fp32 scan kernel's bandwidth is vector-ALU work?  usage: scan8_ablate.py [rows]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d, k, nq = 384, 10, 96
ix = _native.NativeIndex(d, capacity_rows=rows)
ix.fill_synthetic(0xC0FFEE, 0, rows, True)
dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
ix.search_device(dq, nq, k, d_idx, d_score)
ix.synchronize()
ix.profile(True)
out = {}
for rnd in range(3):
    for v in (0, 2, 1):
        ix.set_option("scan8_ablate", v)
        ix.profile_read_gemm()
        ix.search_device(dq, nq, k, d_idx, d_score)
        g = ix.profile_read_gemm()
        out.setdefault(f"ablate_{v}", []).append(round(g["gemm_ms"] / max(g["gemm_launches"], 1), 4))
bytes_per_pass = rows * 388
print(json.dumps({"rows": rows, "phase1_ms_per_query": out,
                  "TBps": {k2: [round(bytes_per_pass / (ms * 1e-3) / 1e12, 3) for ms in v2] for k2, v2 in out.items()}}, indent=1))
