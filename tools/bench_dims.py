#!/usr/bin/env python3
"""Scan-kernel efficiency across dimensions (about 4 GB corpora): which kernel serves each d and what
fraction of the HBM roofline it reaches."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
out = {}
for d in [int(x) for x in sys.argv[1:]] or (16, 32, 64, 96, 100, 128, 200, 256, 300, 384, 512, 768, 1000, 1024, 1536, 3072, 4096):
    rows = max(100_000, int(4e9 // (d * 4)))
    ix = _native.NativeIndex(d, capacity_rows=rows)
    ix.set_option("scan_shadow", 0)  # this tool measures the fp32 scan kernel
    ix.fill_synthetic(0xC0FFEE, 0, rows, True)
    dq = ix.device_queries_synthetic(0xBEEF, 0, 24, True)
    d_idx, d_score = ix.alloc(24 * 80), ix.alloc(24 * 40)
    ix.profile(True)
    ix.search_device(dq, 4, 10, d_idx, d_score); ix.synchronize(); ix.profile_read()
    ix.search_device(dq, 24, 10, d_idx, d_score); ix.synchronize(); p = ix.profile_read()
    ms = p["scan_ms"] / p["scan_launches"]
    gb = rows * ix.pitch * 4 / 1e9
    out[d] = {"rows": rows, "scan_ms": round(ms, 4), "GBps": round(gb / ms * 1e3), "frac_of_8TBps": round(gb / ms * 1e3 / 8000, 3)}
    print(d, out[d], flush=True)
    ix.close()
print(json.dumps(out))
