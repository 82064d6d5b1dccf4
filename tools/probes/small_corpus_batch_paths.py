#!/usr/bin/env python3
"""Blocking batches on corpora below gemm_min_rows: the int8 tiles (gemm_min_rows = 1) against the default choice (tiles from
queries x rows >= gemm_min_work, else the fp32 scan with the round as one grid).  Answers asserted identical.
    python tools/probes/small_corpus_batch_paths.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, "wdbx-py_amd")
from wdbx_amd import _native
d, k = 384, 10
rng = np.random.default_rng(5)
qs = rng.standard_normal((256, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
for n in (21_000, 40_000, 60_000):
    ix = _native.NativeIndex(d, capacity_rows=n)
    ix.fill_synthetic(0xC0FFEE, 0, n, True)
    for nq in (4, 8, 32, 128, 256):
        out = {}
        ref = None
        for rep in range(3):
            for mr in (1, 65536):
                ix.set_option("gemm_min_rows", mr)
                for _ in range(2):
                    r = ix.search(qs[:nq], k)
                lat = []
                for _ in range(10):
                    t0 = time.perf_counter(); r = ix.search(qs[:nq], k); lat.append(time.perf_counter() - t0)
                out[mr] = np.median(lat) * 1e6
                ref = r[0] if ref is None else ref
                assert np.array_equal(r[0], ref), (n, nq, mr)
        print(f"{n} rows, {nq:3d} queries: tiles {out[1]:8.1f} us   default choice {out[65536]:8.1f} us   family {ix.get_option('last_gemm_family')}", flush=True)
    ix.close()
