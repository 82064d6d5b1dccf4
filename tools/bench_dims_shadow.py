#!/usr/bin/env python3
"""Single-query time across dimensions: the default selection path (u8 shadow scan; WDBX_SCAN_SHADOW=1 for the
bf16 tile path) against the fp32 scan on the same handle (about 4 GB of fp32 rows each), with the ids compared."""
import json, os, sys, time
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
    for name, opt in (("selection", int(os.environ.get("WDBX_SCAN_SHADOW", "2"))), ("fp32_scan", 0)):
        ix.set_option("scan_shadow", opt)
        ix.search_device(dq, 4, k, d_idx, d_score); ix.synchronize()
        ix.profile(True); ix.profile_read(); ix.profile_read_gemm(); ix.profile_read_sample()
        t0 = time.perf_counter()
        ix.search_device(dq, nq, k, d_idx, d_score); ix.synchronize()
        el = (time.perf_counter() - t0) / nq * 1e3
        p, g, sm = ix.profile_read(), ix.profile_read_gemm(), ix.profile_read_sample()
        res[name] = {"ms_per_query": round(el, 4), "path": ix.get_option("last_single_path"),
                     "kernels_ms_per_query": round((g["gemm_ms"] + sm["sample_ms"] + p["scan_ms"]) / nq, 4),
                     "idx": d_idx.download(np.int64, (nq, k))}
    same = bool(np.array_equal(res["selection"].pop("idx"), res["fp32_scan"].pop("idx")))
    out[d] = {"rows": rows, "selection": res["selection"], "fp32_scan": res["fp32_scan"],
              "speedup": round(res["fp32_scan"]["ms_per_query"] / res["selection"]["ms_per_query"], 2), "ids_equal": same}
    print(d, out[d], flush=True)
    ix.close()
print(json.dumps(out))
