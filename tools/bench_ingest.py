#!/usr/bin/env python3
"""Ingest rates: host rows -> HBM through wdbx_index_add, with and without the
device normalise kernel; and the Python-level paths (batch_add dict vs add_rows array)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 384
rng = np.random.default_rng(0)
rows = rng.standard_normal((n, d), dtype=np.float32)
out = {}
for norm in (False, True):
    ix = _native.NativeIndex(d, capacity_rows=n)
    ix.add(rows[:1000], normalize=norm)
    ix.clear()
    t0 = time.perf_counter()
    ix.add(rows, normalize=norm)
    dt = time.perf_counter() - t0
    out[f"add_normalize_{int(norm)}"] = {"rows_per_s": n / dt, "GBps": rows.nbytes / dt / 1e9, "seconds": dt}
    ix.close()
t0 = time.perf_counter()
nn = np.sqrt(np.einsum("ij,ij->i", rows, rows))
out["numpy_vectorised_normalise_s"] = time.perf_counter() - t0
print(json.dumps(out))
