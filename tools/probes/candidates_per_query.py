import sys, numpy as np
sys.path.insert(0, "wdbx-py_amd")
from wdbx_amd import _native
ix = _native.NativeIndex(384, capacity_rows=10_000_000)
ix.fill_synthetic(0xC0FFEE, 0, 10_000_000, True)
dq = ix.device_queries_synthetic(0xBEEF, 0, 32, True)
d_idx, d_score = ix.alloc(32 * 10 * 8), ix.alloc(32 * 10 * 4)
for path in (2, 1):
    ix.set_option("scan_shadow", path)
    ix.search_device(dq, 32, 10, d_idx, d_score); ix.synchronize()
    st = ix.batch_status(32)
    print("path", path, "candidates mean/min/max", st["counts"].mean(), st["counts"].min(), st["counts"].max(), "cap", st["capacity"])
print("shadow8_bytes", ix.get_option("shadow8_bytes"), "shadow_bytes", ix.get_option("shadow_bytes"))
