#!/usr/bin/env python3
"""Read-bandwidth ceiling on this device vs the scan kernel (same buffer, same process)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native
rows, dim = 10_000_000, 384
ix = _native.NativeIndex(dim, capacity_rows=rows)
ix.set_option("scan_shadow", 0)  # this tool measures the fp32 scan kernel
ix.fill_synthetic(0xC0FFEE, 0, rows, True)
gb = rows * dim * 4 / 1e9
out = {}
for nt in (1, 0):
    for blocks in (512, 1024, 2048, 4096, 8192):
        ms = ix.probe_read_ms(bool(nt), blocks, 20)
        out[f"probe_nt{nt}_blocks{blocks}"] = round(gb / ms * 1e3, 1)
dq = ix.device_queries_synthetic(0xBEEF, 0, 30, True)
d_idx, d_score = ix.alloc(30 * 80), ix.alloc(30 * 40)
ix.profile(True); ix.search_device(dq, 5, 10, d_idx, d_score); ix.synchronize(); ix.profile_read()
ix.search_device(dq, 30, 10, d_idx, d_score); ix.synchronize(); p = ix.profile_read()
out["scan_kernel_GBps"] = round(gb / (p["scan_ms"] / p["scan_launches"]) * 1e3, 1)
print(json.dumps(out))
