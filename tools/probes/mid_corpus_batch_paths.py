#!/usr/bin/env python3
"""Blocking batches of 4 .. 12 queries on corpora of 70 k .. 200 k rows: the int8 tiles, the default choice, and the rounds on the
fp32 scan (scan_shadow = 0) -- p50 us per call, answers asserted identical.   python tools/probes/mid_corpus_batch_paths.py"""
import sys, time
import numpy as np
sys.path.insert(0, "wdbx-py_amd")
from wdbx_amd import _native
d, k = 384, 10
rng = np.random.default_rng(5)
qs = rng.standard_normal((64, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
for n in (70_000, 100_000, 150_000, 200_000):
    ix = _native.NativeIndex(d, capacity_rows=n)
    ix.fill_synthetic(0xC0FFEE, 0, n, True)
    for nq in (4, 6, 8, 12):
        out = {}
        ref = None
        for rep in range(3):
            for name, mr, mw, sh in (("tiles", 1, 1, 2), ("default", 65536, 800000, 2), ("fp32 rounds", 1 << 40, 0, 0)):
                ix.set_option("gemm_min_rows", mr); ix.set_option("gemm_min_work", mw); ix.set_option("scan_shadow", sh)
                for _ in range(2):
                    r = ix.search(qs[:nq], k)
                lat = []
                for _ in range(10):
                    t0 = time.perf_counter(); r = ix.search(qs[:nq], k); lat.append(time.perf_counter() - t0)
                out[name] = (np.median(lat) * 1e6, ix.get_option("last_single_path"))
                ref = r[0] if ref is None else ref
                assert np.array_equal(r[0], ref), (n, nq, name)
        print(f"{n} rows, {nq:2d} queries: tiles {out['tiles'][0]:7.1f} us   default {out['default'][0]:7.1f} us   fp32 rounds {out['fp32 rounds'][0]:7.1f} us   work {nq*n}", flush=True)
    ix.close()
