#!/usr/bin/env python3
"""Lone blocking queries through an in-process group of 2 shards on one GPU (device-copy exchange): p50 wall clock with the
polled completion word (option poll_done on the root shard) and with the stream wait, alternating in one process.
    python tools/probes/group_lone_latency.py <rows>"""
import sys, time
import numpy as np
sys.path.insert(0, "wdbx-py_amd")
from wdbx_amd import _native
n, d, k = int(sys.argv[1]), 384, 10
S = 2
shards = []
for s in range(S):
    ix = _native.NativeIndex(d, capacity_rows=n // S)
    ix.fill_synthetic(0xC0FFEE, s * (n // S), n // S, True)
    shards.append(ix)
grp = _native.NativeGroup.attach(shards, exchange=_native.NativeGroup.EXCHANGE_COPY)
grp.set_row_bases([s * (n // S) for s in range(S)])
rng = np.random.default_rng(1)
qs = rng.standard_normal((64, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
ref = None
for rep in range(4):
    for poll in (1, 0):
        shards[0].set_option("poll_done", poll)
        for q in qs[:4]:
            grp.search_merged(q[None, :], k, k)
        lat, res = [], []
        for q in qs:
            t0 = time.perf_counter()
            r = grp.search_merged(q[None, :], k, k)
            lat.append(time.perf_counter() - t0)
            res.append(r[0][0].tolist())
        if ref is None:
            ref = res
        assert res == ref
        if rep == 3:
            print(n, "poll_done", poll, "group lone p50 us", round(float(np.median(lat)) * 1e6, 1))
