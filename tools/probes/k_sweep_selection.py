#!/usr/bin/env python3
"""Per-query time of the u8 selection scan across k (10 M x 384 cosine), against the fp32 scan."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
n, d, nq = 10_000_000, 384, 32
ix = _native.NativeIndex(d, capacity_rows=n)
ix.fill_synthetic(0xC0FFEE, 0, n, True)
dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
for k in (1, 10, 100, 199, 200, 500, 1000, 2048):
    d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    out = {}
    for name, opt in (("u8", 2), ("fp32", 0)):
        ix.set_option("scan_shadow", opt)
        ix.search_device(dq, 8, k, d_idx, d_score); ix.synchronize()
        ix.profile(True); ix.profile_read(); ix.profile_read_gemm(); ix.profile_read_sample()
        t0 = time.perf_counter()
        ix.search_device(dq, nq, k, d_idx, d_score); ix.synchronize()
        el = (time.perf_counter() - t0) / nq * 1e3
        p, g, s = ix.profile_read(), ix.profile_read_gemm(), ix.profile_read_sample()
        st = ix.batch_status(nq if k < 200 else 1) if ix.get_option("last_single_path") == 2 else None
        out[name] = {"ms": round(el, 4), "path": ix.get_option("last_single_path"), "scan_ms": round((g["gemm_ms"] + s["sample_ms"] + p["scan_ms"]) / nq, 4),
                     "merge_ms": round(p["merge_ms"] / nq, 4), "cand": None if st is None else int(st["counts"].mean())}
    print(k, out, flush=True)
