#!/usr/bin/env python3
"""Single-query time across dimensions: the bf16-shadow selection path (default) against the fp32 scan on
the same handle (about 4 GB of fp32 rows each), with the ids compared."""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
out = {}
nq, k = 24, 10
for d in [int(x) for x in sys.argv[1:]] or (32, 64, 100, 128, 200, 256, 300, 384, 512, 768, 1000, 1024, 1536, 3072, 4096):
    rows = max(100_000, int(4e9 // (d * 4)))
    ix = _native.NativeIndex(d, capacity_rows=rows)
    ix.fill_synthetic(0xC0FFEE, 0, rows, True)
    dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
    d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    res = {}
    for name, opt in (("shadow", 1), ("fp32_scan", 0)):
        ix.set_option("scan_shadow", opt)
        ix.search_device(dq, 4, k, d_idx, d_score); ix.synchronize()
        ix.profile(True); ix.profile_read(); ix.profile_read_gemm()
        import time
        t0 = time.perf_counter()
        ix.search_device(dq, nq, k, d_idx, d_score); ix.synchronize()
        el = (time.perf_counter() - t0) / nq * 1e3
        p, g = ix.profile_read(), ix.profile_read_gemm()
        res[name] = {"ms_per_query": round(el, 4), "kernel_ms": round((g["gemm_ms"] / max(g["gemm_launches"] / 2, 1)) if opt else p["scan_ms"] / max(p["scan_launches"], 1), 4),
                     "idx": d_idx.download(np.int64, (nq, k))}
    same = bool(np.array_equal(res["shadow"].pop("idx"), res["fp32_scan"].pop("idx")))
    pitch16 = (d + 127) // 128 * 128
    sh_gb = rows * pitch16 * 2 * (1 + 1 / 32) / 1e9
    out[d] = {"rows": rows, "shadow": res["shadow"], "fp32_scan": res["fp32_scan"], "speedup": round(res["fp32_scan"]["ms_per_query"] / res["shadow"]["ms_per_query"], 2),
              "shadow_frac_of_8TBps": round(sh_gb / res["shadow"]["kernel_ms"] * 1e3 / 8000, 3), "ids_equal": same}
    print(d, out[d], flush=True)
    ix.close()
print(json.dumps(out))
