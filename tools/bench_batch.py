#!/usr/bin/env python3
"""BASELINE config 4 measurement: 10M x 384 fp32, cosine, batch of 256 queries sharing one matrix-core
pass (default: bf16 selection tiles over the bf16 shadow + exact fp32 re-scoring; WDBX_OPTS=gemm_bf16=0 for the
exact fp32 tiles), top-10.  Reports queries/s, achieved fp32 TFLOP/s of the GEMM launches (HIP events) and the
effective corpus read rate.  usage: bench_batch.py [rows] [dim] [nq] [k] [reps]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))
from wdbx_amd import _native  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
metric = int(sys.argv[6]) if len(sys.argv) > 6 else 0  # 0 cosine, 1 L2

ix = _native.NativeIndex(dim, metric=metric, capacity_rows=rows)
ix.fill_synthetic(0xC0FFEE, 0, rows, True)
import os  # noqa: E402
opts = dict(kv.split("=") for kv in os.environ.get("WDBX_OPTS", "").split(",") if kv)   # e.g. WDBX_OPTS=gemm_bf16=1
ref = None
want = int(opts.get("gemm_bf16", 3))  # 3 = i8 tiles (default), 2 = bf16 tiles on the bf16 shadow, 1 = on the fp32 rows, 0 = fp32 tiles
if want:  # reference answer from the fp32 tiles first
    ix.set_option("gemm_bf16", 0)
    dq0 = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
    r_idx, r_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
    ix.search_batch_device(dq0, nq, k, r_idx, r_score)
    ref = (r_idx.download(np.int64, (nq, k)), r_score.download(np.float32, (nq, k)))
    ix.set_option("gemm_bf16", want)
for name, v in opts.items():
    ix.set_option(name, int(v))
dq = ix.device_queries_synthetic(0xBEEF, 0, nq, True)
d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
ix.search_batch_device(dq, nq, k, d_idx, d_score)
st = ix.batch_status(nq)
agree = None
if ref is not None:
    got = (d_idx.download(np.int64, (nq, k)), d_score.download(np.float32, (nq, k)))
    agree = {"ids_equal": bool(np.array_equal(got[0], ref[0])), "max_score_diff": float(np.max(np.abs(got[1] - ref[1]))),
             "rows_differing": int(np.sum(np.any(got[0] != ref[0], axis=1)))}
ix.profile(True)
ix.profile_read_gemm()
t0 = time.perf_counter()
for _ in range(reps):
    ix.search_batch_device(dq, nq, k, d_idx, d_score)
ix.synchronize()
el = (time.perf_counter() - t0) / reps
g = ix.profile_read_gemm()
m = ix.profile_read()
blocks = (nq + 255) // 256
flops_full = 2.0 * blocks * 256 * dim * rows          # phase 1 (all tiles), padded block width
gemm_ms = g["gemm_ms"] / reps
# single-query reference on the same corpus
ix.profile_read()
ix.search_device(dq, 8, k, d_idx, d_score)
ix.synchronize()
scan = ix.profile_read()
out = {
    "workload": f"{rows} x {dim} fp32, {'L2' if metric else 'cosine'}, batch_queries={nq} as one MFMA pass, top-{k}",
    "ms_per_batch": el * 1e3, "queries_per_s": nq / el,
    "gemm_ms_per_batch": gemm_ms, "gemm_launches_per_batch": g["gemm_launches"] / reps,
    "tflops_useful": 2.0 * nq * dim * rows / el / 1e12,
    "tflops_gemm_kernels": flops_full * (1 + 1.0 / 32) / (gemm_ms * 1e-3) / 1e12,
    "mfma_peak_tflops": 157.3,
    "corpus_GBps_effective": rows * dim * 4 / el / 1e9,
    "merge_ms_per_batch": m["merge_ms"] / reps,
    "candidates_per_query_mean": float(st["counts"].mean()), "candidates_max": int(st["counts"].max()),
    "capacity": st["capacity"], "overflowed": st["overflowed"],
    "single_query_scan_ms": scan["scan_ms"] / max(scan["scan_launches"], 1),
    "gemm_family": ix.get_option("last_gemm_family"), "shadow_bytes": ix.get_option("shadow_bytes"),
    "options": opts, "agreement_with_fp32_tiles": agree,
    "speedup_vs_single_query_scans": (scan["scan_ms"] / max(scan["scan_launches"], 1)) * nq / (el * 1e3),
}
print(json.dumps(out))
