#!/usr/bin/env python3
"""bench.py -- WDBX vector_search hot path on MI355X: queries/sec, latency, HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload t|c2|c3|c1] [--scaling strong|weak]

One "step" = one single-vector query = one full scan of the (rank's) corpus + top-k
(+ the cross-rank exchange for N > 1).  Inputs (corpus, queries) are resident in HBM
before the timed region.  The timed region enqueues exactly K steps on the shard's
stream and is bracketed by barrier + device synchronisation; the maximum over ranks
is reported by rank 0 as ONE JSON line.

Workloads (BASELINE.json configs / BASELINE.md section 3):
  t   10M x 384 fp32, cosine, top-10   north-star target row (default)
  c2  1M  x 384 fp32, cosine, top-10   BASELINE configs[1]
  c3  10M x 768 fp32, L2,     top-100  BASELINE configs[2]
  c1  10k x 384 fp32, cosine, top-10   BASELINE configs[0] (the reference's CPU-runnable case)
  c4  10M x 384 fp32, cosine, top-10, batch_queries=256 on the matrix cores   BASELINE configs[3]
      (one step = one batch of 256 queries; value stays queries/s; default tiles: int8 selection (v_mfma_i32_16x16x64_i8)
      over the group-scaled i8 shadow copy + exact fp32 re-scoring, roofline bound = hbm; --opt gemm_bf16=2: bf16 selection
      tiles over the bf16 shadow; --opt gemm_bf16=0: exact fp32 tiles, bound = mfma)
N > 1, contiguous row ranges; "strong" (default) splits the workload's rows over the GPUs, "weak" gives every GPU the
full row count (C5 = t at N=8, weak).  The exchange is an RCCL all-gather of the per-shard (row, score) records inside
the library, followed by a merge kernel.  Two ways to drive N GPUs, same library path underneath:
  * `python bench.py --gpus N` as typed (no launcher environment): ONE process, the in-process shard group
    (wdbx_group_attach + wdbx_group_search_resident: a host thread per shard, communicators from ncclCommInitAll) -- the
    reference's VectorStore(num_shards=N) shape (wdbx/core/vector_store.py:323-345);
  * under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (RANK / WORLD_SIZE in the environment): one
    process per GPU, ncclCommInitRank.  No torch in either: the unique id travels through a file of the launcher's
    process group, barrier and max-reduction of the time go through RCCL itself (wdbx_index_comm_allgather_host).
N > 1 lines say how the exchange was amortised: `value` is the STREAM form (the K timed queries are resident and enqueued as
one call; the library cuts a call into chunks and issues ONE all-gather + merge per chunk: `exchanges_in_timed_region`), and
`per_query_exchange` is a second timed leg of the same K queries as K calls of one query each, back to back without a host
synchronisation -- one all-gather + merge PER QUERY, the reference's call shape (vector_store.py:323-345 merges per query).
`--mode group` runs the in-process group at N = 1 too (1-rank communicator); `--devices 0,0,0,0` with `--gpus 4` rehearses
four shards on one GPU (exchange by device copies, reported as such).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))

from wdbx_amd import _native  # noqa: E402
from wdbx_amd.shard_group import ShardGroup, merge_topk, shard_row_range  # noqa: E402

SEED_CORPUS, SEED_QUERY = 0xC0FFEE, 0xBEEF
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)

WORKLOADS = {
    "t": dict(rows=10_000_000, dim=384, metric="cosine", k=10, name="10M x 384 fp32, cosine, top-10"),
    "c2": dict(rows=1_000_000, dim=384, metric="cosine", k=10, name="1M x 384 fp32, cosine, top-10"),
    "c3": dict(rows=10_000_000, dim=768, metric="l2", k=100, name="10M x 768 fp32, L2, top-100"),
    "c1": dict(rows=10_000, dim=384, metric="cosine", k=10, name="10k x 384 fp32, cosine, top-10"),
    # BASELINE configs[3]: one step = one batch of 256 queries sharing one corpus pass on the matrix cores
    "c4": dict(rows=10_000_000, dim=384, metric="cosine", k=10, batch=256,
               name="10M x 384 fp32, cosine, batch_queries=256 as one MFMA pass, top-10"),
}
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_I8_SUSTAINED_TOPS = 4050.0  # bare v_mfma_i32_16x16x64_i8 loop on random bytes, ~2.0 GHz held (profiles/r02/mfma_shape_clock.txt)
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8: twice the bf16 rate per clock (MI355X_MICROARCH.md, Matrix cores)


def batch_roofline(ix, wl, rows, gemm_ms_per_batch, k, traffic_db=None):
    """Roofline record of the batched path's tile kernel pair (sample pass + full pass) for one batch.
    fp32 tiles are bound by the fp32 MFMA peak; the bf16 selection tiles by HBM: they read the bf16 shadow
    copy (2 B/element, rows padded to 128 elements) or the fp32 rows once per pass."""
    family = ix.get_option("last_gemm_family")
    div = ix.get_option("gemm_sample_div") or (32 if family == 0 else min(32, max(4, 1024 // k)))
    passes = 1.0 + 1.0 / div
    flops = 2.0 * wl.get("batch", 1) * wl["dim"] * rows * passes
    tf = flops / (gemm_ms_per_batch * 1e-3) / 1e12 if gemm_ms_per_batch > 0 else 0.0
    if family == 0:
        return {"bound": "mfma", "kernel": "gemm_topk_kernel", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_step": flops,
                "gemm_ms_per_step": gemm_ms_per_batch}
    if family == 3:  # i8 tiles: one signed byte per element (rows padded to 128 bytes) + 16 bytes of group table per 64 rows
        pitch8 = (wl["dim"] + 127) // 128 * 128
        alg = rows * (pitch8 + 0.25) * passes
        gbps = alg / (gemm_ms_per_batch * 1e-3) / 1e9 if gemm_ms_per_batch > 0 else 0.0
        tops = 2.0 * wl.get("batch", 1) * pitch8 * rows * passes / (gemm_ms_per_batch * 1e-3) / 1e12 if gemm_ms_per_batch > 0 else 0.0
        t = profiled_traffic(traffic_db or {}, "c4_i8", rows, wl["dim"])
        return {"bound": "hbm", "kernel": "gemm_i8_kernel<phase 0 + phase 1> (i8 selection tiles over the group-scaled i8 shadow)",
                "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": t,
                "traffic_source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this configuration)" if t else None,
                "i8_mfma_sustained_TOPs_bare_loop": MFMA_I8_SUSTAINED_TOPS,
                "algorithmic_bytes_per_step": alg, "gemm_ms_per_step": gemm_ms_per_batch,
                "i8_mfma_TOPs": tops, "i8_mfma_frac": tops / MFMA_I8_PEAK_TOPS,
                "note": "selection pass pair only (sampled tiles + all tiles); the kept rows are re-scored in fp32 from the "
                        "fp32 rows (scattered reads, not counted here)"}
    el_bytes = 2 if family == 2 else 4
    pitch = (wl["dim"] + 127) // 128 * 128 if family == 2 else (wl["dim"] + 3) // 4 * 4
    alg = rows * pitch * el_bytes * passes
    gbps = alg / (gemm_ms_per_batch * 1e-3) / 1e9 if gemm_ms_per_batch > 0 else 0.0
    return {"bound": "hbm", "kernel": "gemm_bf16w8_kernel" + ("<shadow>" if family == 2 else "<fp32 rows>"), "achieved": gbps,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": None,
            "algorithmic_bytes_per_step": alg, "gemm_ms_per_step": gemm_ms_per_batch,
            "bf16_mfma_TFLOPs": tf, "bf16_mfma_frac": tf / MFMA_BF16_PEAK_TFLOPS,
            "note": "selection pass only; candidates are re-scored in fp32 from the fp32 rows (scattered reads, "
                    "not counted here)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="t", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--rows", type=int, default=0, help="override the workload's row count")
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--latency-queries", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="auto", choices=["auto", "index", "group"],
                    help="auto: plain index at N=1, in-process shard group for N>1 without a launcher, one rank per GPU under "
                         "a launcher; group: the in-process shard group also at N=1")
    ap.add_argument("--devices", default="", help="comma-separated device ids of the in-process group's shards "
                                                  "(default 0..N-1; repeating a device rehearses several shards on one GPU)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (experiments)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the brief runs of the other BASELINE configs")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events in the timed region (overhead check)")
    ap.add_argument("--verify", type=int, default=4,
                    help="re-derive the top-k of this many TIMED queries with the oracle from the rows read back from HBM "
                         "(outside the timed region) and report parity_check")
    ap.add_argument("--no-facade", action="store_true", help="skip the WDBX.vector_search wall-clock leg")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child passes (falls back to profiles/hbm_traffic.json)")
    ap.add_argument("--traffic-steps", type=int, default=24, help="queries (batches) per rocprofv3 child pass")
    return ap.parse_args()


def host_rows(ix, want_rows):
    """The stored rows read back from HBM for the CPU legs: the whole corpus when host memory allows (MemAvailable >=
    2.5 x its bytes), else the first 1 M rows.  Returns (rows, whole)."""
    n = ix.size()
    need = n * ix.dim * 4
    avail = 0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    take = n if (want_rows >= n and avail >= 2.5 * need) else min(n, 1_000_000)
    rows = np.empty((take, ix.dim), np.float32)
    step = 1 << 20
    for r0 in range(0, take, step):
        rows[r0:r0 + step] = ix.get_rows(r0, min(step, take - r0))
    return rows, take == n


def cpu_baseline(rows, whole, wl, k, metric_id, budget_s):
    """The oracle (numpy restatement of the reference's exact path, wdbx/core/indexing.py:983-1030) timed on this box's
    host cores: the same corpus bytes (read back from HBM) and the same query generator -- on the WHOLE corpus when it fits
    in host memory, else on its first 1 M rows scaled linearly in rows to the workload.  Two forms:
      * `single_call`: one OpenBLAS sgemv over the whole matrix per query + top-k (numpy's bundled OpenBLAS stops at 64
        threads, whatever the box has);
      * the reported value: the SAME arithmetic slab-parallel on every usable CPU (oracle.ParallelFlatSearch: one pinned
        worker thread per CPU, each scoring and ranking its own row slab with a single-threaded sgemv, lists merged with the
        oracle's total order; BLAS limited to one thread per call through threadpoolctl) -- "on all host cores"."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import wdbx_oracle as O

    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits = None
        blas_threads = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None
    try:  # a cgroup CPU quota below the affinity mask bounds what the threads can get
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    sample_rows = rows.shape[0]
    queries = O.normalize_rows_fast(O.synth_rows(SEED_QUERY, 0, 64, wl["dim"]))
    scale = wl["rows"] / sample_rows
    how = ("the whole corpus" if whole and scale == 1 else
           f"the first {sample_rows} rows of the same corpus, scaled x{scale:g} in rows to the workload")

    # ---- one sgemv call per query (<= 64 BLAS threads) ----
    first = O.flat_search(rows, queries[0], k, metric_id, normalize_query=False)  # warm-up
    times, t_end, i = [], time.perf_counter() + budget_s / 3.0, 0
    while (time.perf_counter() < t_end or i < 3) and i < 2000:
        t0 = time.perf_counter()
        O.flat_search(rows, queries[i % len(queries)], k, metric_id, normalize_query=False)
        times.append(time.perf_counter() - t0)
        i += 1
    per_query = float(np.median(times))
    single = {"value": 1.0 / (per_query * scale), "cores": int(min(blas_threads, usable)), "queries": len(times),
              "median_ms_per_query": per_query * 1e3, "sample_gbps": sample_rows * wl["dim"] * 4 / per_query / 1e9}

    # ---- the same arithmetic on every usable CPU ----
    par = None
    try:
        limit = threadpool_limits(limits=1, user_api="blas") if threadpool_limits else None
        try:
            t_setup = time.perf_counter()
            pfs = O.ParallelFlatSearch(rows, workers=usable, metric=metric_id)
            t_setup = time.perf_counter() - t_setup
            try:
                warm = pfs.search_many(queries[:1], k)[0]
                same = bool(np.array_equal(warm[0], first[0]))
                t0 = time.perf_counter()
                pfs.search_many(queries[:4], k)
                per4 = time.perf_counter() - t0
                nq = int(max(8, min(2000, (budget_s * 2.0 / 3.0) / max(per4 / 4, 1e-6))))
                qs = np.concatenate([queries] * (nq // len(queries) + 1))[:nq]
                t0 = time.perf_counter()
                pfs.search_many(qs, k)
                el = time.perf_counter() - t0
                par = {"value": nq / el / scale, "cores": pfs.workers, "pinned_workers": pfs.pinned, "queries": nq, "seconds": el,
                       "setup_seconds": t_setup, "sample_gbps": sample_rows * wl["dim"] * 4 * nq / el / 1e9,
                       "ids_equal_single_call": same}
            finally:
                pfs.close()
        finally:
            if limit is not None:
                limit.restore_original_limits() if hasattr(limit, "restore_original_limits") else limit.unregister()
    except Exception as e:  # the single-call figure stands by itself
        par = {"error": f"{type(e).__name__}: {e}"}

    best = par if (par and "value" in par and par["value"] >= single["value"]) else single
    host = f"host cpu_count={os.cpu_count()}, usable (affinity)={usable}" + (f", cgroup quota={quota:g} CPUs" if quota else "")
    if best is par:
        sample = (f"numpy oracle slab-parallel (one pinned worker per usable CPU, single-threaded sgemv + top-k per slab, merged): "
                  f"{par['queries']} queries on {how} in {par['seconds']:.1f} s; {host}")
    else:
        sample = (f"numpy oracle (one OpenBLAS sgemv + top-k per query), {single['queries']} queries on {how}, median "
                  f"{per_query * 1e3:.2f} ms/query; {host}")
    return {
        "value": best["value"],
        "unit": "queries/s",
        "cores": int(best["cores"]),
        "kind": "port",
        "sample": sample,
        "extrapolated": not (whole and scale == 1),
        "sample_qps": best["value"] * scale,
        "sample_gbps": best["sample_gbps"],
        "host_cpu_count": os.cpu_count(),
        "usable_cpus": usable,
        "cgroup_cpu_quota": quota,
        "parallel": par,
        "single_call": single,
    }


def verify_timed_queries(rows, whole, ix, queries, res_idx, res_score, k, metric_id, row_base=0):
    """Parity of the TIMED output (outside the timed region): the top-k of the first len(queries) timed queries
    re-derived by the oracle from the rows read back from HBM; ids identical, scores within 1e-5 (cosine) / 1e-5
    relative (L2).  Needs the whole corpus on the host."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import wdbx_oracle as O

    if not whole or not len(queries):
        return {"parity_check": "skipped", "reason": "corpus not held on the host" if len(queries) else "no queries"}
    exp = O.slab_search(lambda r0, c: rows[r0:r0 + c], rows.shape[0], queries, k, metric_id, slab=1_000_000)
    worst, bad = 0.0, []
    for i, (e_idx, e_score) in enumerate(exp):
        got_i, got_s = res_idx[i] - row_base, res_score[i]
        tol = 1e-5 * (np.maximum(1.0, np.abs(e_score)) if metric_id == O.METRIC_L2 else 1.0)
        if got_i[: len(e_idx)].tolist() != e_idx.tolist() or not np.all(np.abs(got_s[: len(e_idx)] - e_score) <= tol):
            bad.append(i)
        worst = max(worst, float(np.max(np.abs(got_s[: len(e_idx)] - e_score))))
    return {"parity_check": "ok" if not bad else f"MISMATCH on timed queries {bad}", "queries_checked": len(exp),
            "max_abs_score_diff": worst, "oracle": "oracle/wdbx_oracle.py slab_search over the rows read back from HBM"}


def facade_latency(wl, k, n_queries=200, num_shards=1, devices=None):
    """Wall clock of the reference-facing call on the headline corpus (SURVEY 8d "timing method"): ``WDBX.vector_search(list,
    limit)`` -- list -> ndarray, normalisation, ctypes, query to the device, kernels, results back, id mapping, merge,
    metadata -- one caller, one query at a time, next to the blocking C-ABI call on the same handle."""
    import tempfile

    from wdbx_amd import WDBX

    tmp = tempfile.mkdtemp(prefix="wdbx_bench_")
    cfg = {"HIP_CAPACITY_ROWS": -(-wl["rows"] // num_shards), "HIP_METRIC": wl["metric"], "HIP_PERSIST_INDEX": False}  # a scratch corpus
    if num_shards > 1:  # the one-call fan-out (shard group) is what is measured; the per-shard calls + Python merge beside it
        cfg.update(HIP_GROUP_SEARCH=True, HIP_DEVICES=devices)
    w = WDBX(vector_dimension=wl["dim"], num_shards=num_shards, data_dir=tmp, config=cfg, enable_plugins=False, enable_gpu=True,
             log_level="ERROR")
    try:
        w.vector_store.bulk_store_synthetic(wl["rows"], SEED_CORPUS)
        sys.path.insert(0, str(ROOT / "oracle"))
        import wdbx_oracle as O  # (query generator only)

        queries = [q.tolist() for q in O.synth_rows(SEED_QUERY, 0, n_queries + 20, wl["dim"])]
        for q in queries[:20]:
            w.vector_search(q, limit=k)
        lat = []
        for q in queries[20:]:
            t0 = time.perf_counter()
            r = w.vector_search(q, limit=k)
            lat.append(time.perf_counter() - t0)
        assert len(r) == k
        if num_shards > 1:
            path = w.vector_store.last_search_path
            want = [w.vector_search(q, limit=k) for q in queries[:8]]
            w.vector_store._group.close()
            w.vector_store._group = False  # the reference's shape: one call per shard (thread pool) + Python merge
            same = [w.vector_search(q, limit=k) for q in queries[:8]] == want
            tl = []
            for q in queries[20:]:
                t0 = time.perf_counter()
                w.vector_search(q, limit=k)
                tl.append(time.perf_counter() - t0)
            return {"what": "wall clock per call, single client, %s over %d shards" % (wl["name"], num_shards),
                    "WDBX.vector_search (%s)" % path: {"p50": float(np.percentile(lat, 50) * 1e3), "p99": float(np.percentile(lat, 99) * 1e3),
                                                       "qps": float(1.0 / np.median(lat))},
                    "WDBX.vector_search (per-shard calls on a thread pool + Python merge)": {
                        "p50": float(np.percentile(tl, 50) * 1e3), "p99": float(np.percentile(tl, 99) * 1e3)},
                    "group_equals_per_shard_path": bool(same)}
        nix = w.vector_store.indices[0]._native
        qn = np.asarray(queries[0], np.float32)
        qn /= np.linalg.norm(qn)
        nl = []
        for _ in range(n_queries):
            t0 = time.perf_counter()
            nix.search(qn, k)
            nl.append(time.perf_counter() - t0)
        return {"what": "wall clock per call, single client, %s" % wl["name"],
                "WDBX.vector_search": {"p50": float(np.percentile(lat, 50) * 1e3), "p99": float(np.percentile(lat, 99) * 1e3),
                                       "qps": float(1.0 / np.median(lat))},
                "wdbx_index_search (blocking C ABI)": {"p50": float(np.percentile(nl, 50) * 1e3),
                                                       "p99": float(np.percentile(nl, 99) * 1e3)},
                "single_path": nix.get_option("last_single_path")}
    except Exception as e:  # an extra must never cost the main result
        return {"error": str(e)}
    finally:
        try:
            import asyncio
            import shutil

            asyncio.run(w.shutdown())
            shutil.rmtree(tmp, ignore_errors=True)
        except Exception:
            pass


def live_traffic(args, batch):
    """roofline.traffic measured for THIS run's command line: two child passes of bench.py under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (counters in passes of their own, kernel trace only -- the recipe of
    MI355X_MICROARCH.md), each a short run of the same workload / options with every extra leg switched off.  Started after
    the last timed leg (children, never an exec).  Per unit (query, or batch of the batched workload):
        (2 * sum FETCH_SIZE + sum WRITE_SIZE) KiB * 1024 / units      over every launch of the dominant kernel's family
    (the guide's gfx950 correction: FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes).
    Returns (record, error): record = {"by_family": {family: {...}}, "units": n} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "already running under a profiler"
    steps, warm = max(4, args.traffic_steps), 4
    child = [sys.executable, str(ROOT / "bench.py"), "--workload", args.workload, "--steps", str(steps), "--warmup", str(warm),
             "--latency-queries", "0", "--verify", "0", "--no-cpu-baseline", "--no-facade", "--no-other-configs",
             "--no-live-traffic", "--no-profile"]
    if args.rows:
        child += ["--rows", str(args.rows)]
    if args.k:
        child += ["--k", str(args.k)]
    for o in args.opt:
        child += ["--opt", o]
    tmp = tempfile.mkdtemp(prefix="wdbx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass: rc {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {counter} pass wrote no counter_collection.csv"
            for f in files:
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != counter:
                            continue
                        fam = row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").strip()
                        rec = sums.setdefault(fam, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
                        rec[counter] += float(row["Counter_Value"])
                        if counter == "FETCH_SIZE":
                            rec["launches"] += 1
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 child pass timed out"
    except Exception as e:  # the bench line must not depend on the profiler
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    units = warm + steps
    fams = {f: {"bytes_per_unit": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / units, "FETCH_SIZE_KiB_sum": v["FETCH_SIZE"],
                "WRITE_SIZE_KiB_sum": v["WRITE_SIZE"], "launches": v["launches"]} for f, v in sums.items()}
    return {"by_family": fams, "units": units, "unit": "batch" if batch > 1 else "query"}, None


def apply_live_traffic(roofline, live, live_err, family):
    """Put the measured bytes of the dominant kernel's family into the roofline record (or say why not)."""
    # (the u8 selection scan's family: its full passes and both forms of its sample pass)
    members = [f for f in (live or {}).get("by_family", {}) if f == family or (family == "scan8_kernel" and f.startswith("scan8_"))]
    if members:
        parts = [live["by_family"][f] for f in members]
        rec = {"bytes_per_unit": sum(p["bytes_per_unit"] for p in parts), "FETCH_SIZE_KiB_sum": sum(p["FETCH_SIZE_KiB_sum"] for p in parts),
               "WRITE_SIZE_KiB_sum": sum(p["WRITE_SIZE_KiB_sum"] for p in parts), "launches": sum(p["launches"] for p in parts),
               "kernels": members}
        roofline["traffic"] = rec["bytes_per_unit"]
        roofline["traffic_source"] = (f"live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this command line "
                                      f"({live['units']} {live['unit']}s each, {rec['launches']} {family} launches); "
                                      "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / units (gfx950 FETCH correction)")
        roofline["traffic_detail"] = {"family": family, **rec, "units": live["units"]}
        alg = roofline.get("algorithmic_bytes_per_launch") or roofline.get("algorithmic_bytes_per_step")
        if alg:
            roofline["traffic_over_algorithmic"] = rec["bytes_per_unit"] / alg
    elif live_err or live:
        roofline["traffic_live_error"] = live_err or f"no {family} launches in the profiled pass"
        if roofline.get("traffic") is not None and not roofline.get("traffic_source"):
            roofline["traffic_source"] = "profiles/hbm_traffic.json (committed rocprofv3 --pmc passes of this configuration)"
    return roofline


def profiled_traffic(traffic_db, key, rows, dim):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json) -- only when that record
    was taken on THIS configuration (rows, dim); a run on other sizes reports null."""
    rec = traffic_db.get(key) or {}
    if rec.get("rows") == rows and rec.get("dim") == dim:
        return rec.get("bytes_per_launch")
    return None


def single_query_roofline(ix, wl, rows, k, prof, gprof, traffic_db, key, sample_qn=None):
    """Roofline record of the dominant kernel of the single-query path.  Default: every query makes its own
    SELECTION SCAN PAIR (sampled groups + all rows, scan8_kernel) over the u8 shadow copy, then exact fp32
    re-scoring of the kept candidates; `--opt scan_shadow=1`: the same over the bf16 shadow on the tile kernel;
    `--opt scan_shadow=0` (or a handle without a shadow): the fp32 scan_kernel.  Algorithmic bytes = what the
    kernel must read once: rows * (pitch8 + 4) * (1 + 1/div), rows * pitch16 * 2 * (1 + 1/div), rows * d * 4
    (DESIGN.md 4.1 / 4.2d / 4.2e)."""
    if prof["scan_launches"] == 0 and gprof["gemm_launches"] > 0:
        pairs = gprof["gemm_launches"] / 2
        ms = gprof["gemm_ms"] / max(pairs, 1)
        div = ix.get_option("gemm_sample_div") or min(32, max(4, 1024 // k))
        if ix.get_option("last_single_path") == 2:  # u8 selection scan: 1 byte per element + a 4-byte scale per row
            # one full-pass launch per query (gemm pool) + one sample launch per round of 32 queries (sample pool)
            sprof = ix.profile_read_sample()
            pairs = gprof["gemm_launches"]
            ms = (gprof["gemm_ms"] + sprof["sample_ms"]) / max(pairs, 1)
            pieces = next(p for p in (8, 16, 24, 32, 48, 64, 96, 128, 192, 256) if p * 16 >= wl["dim"])
            per_row = pieces * 16 + 4 + (4 if wl["metric"] == "l2" else 0)
            # (a round's sample launch reads each sampled row once per `qn` queries: scan8_sample4_kernel)
            qn = max(1, sample_qn if sample_qn is not None else ix.get_option("last_sample_qn"))
            alg = rows * per_row * (1.0 + 1.0 / (div * qn))
            ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            t = profiled_traffic(traffic_db, key + "_u8", rows, wl["dim"])
            return {"bound": "hbm", "kernel": "scan8_kernel<phase 0 + phase 1> (one selection scan pair per query over the u8 shadow)",
                    "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": t,
                    "traffic_source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                      "on this configuration)" if t else None,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "launches_timed": int(pairs),
                    "fp32_rows_equivalent_GBps": rows * wl["dim"] * 4 / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                    "sample_queries_per_workgroup": qn,
                    "note": "a launch = the scan pair (sampled groups + all rows) of one query; it reads the 1-byte shadow copy "
                            "and the per-row scales, so the fp32-equivalent rate (rows*d*4 per query, SURVEY 8d) exceeds what any "
                            "fp32 scan can reach; the candidates' exact fp32 re-scoring (rescore_kernel) and two merge launches "
                            "are outside this kernel",
                    "merge_avg_ms": prof["merge_ms"] / max(prof["merge_launches"], 1)}
        pitch16 = (wl["dim"] + 127) // 128 * 128
        alg = rows * pitch16 * 2 * (1.0 + 1.0 / div)
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        t = profiled_traffic(traffic_db, key + "_shadow", rows, wl["dim"])
        return {"bound": "hbm", "kernel": "gemm_bf16w8_kernel<phase 0 + phase 1, shadow> (one selection pass pair per query)",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": t,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "launches_timed": int(pairs),
                "fp32_rows_equivalent_GBps": rows * wl["dim"] * 4 / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                "note": "a launch = the pass pair of one query; it reads the 2-byte shadow copy, so the fp32-equivalent "
                        "rate (rows*d*4 per query, SURVEY 8d) exceeds what any fp32 scan can reach; the candidates' fp32 "
                        "re-scoring (rescore_kernel, ~1 k rows) and two merge launches are outside this kernel",
                "merge_avg_ms": prof["merge_ms"] / max(prof["merge_launches"], 1)}
    ms = prof["scan_ms"] / max(prof["scan_launches"], 1)
    alg = rows * wl["dim"] * 4  # N*d*4 per query (per launch: this rank's rows)
    ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"bound": "hbm", "kernel": "scan_kernel", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS, "traffic": profiled_traffic(traffic_db, key, rows, wl["dim"]),
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "launches_timed": prof["scan_launches"],
            "merge_avg_ms": prof["merge_ms"] / max(prof["merge_launches"], 1)}


def quick_config(name, reuse=None, steps=100, opts=None):
    """Short measurement of another BASELINE config in the same run (outside the timed region):
    pipelined queries/s, kernel average (HIP events) and roofline fraction, single-client p50.
    ``opts``: library options for the duration of the measurement (restored afterwards)."""
    wl = WORKLOADS[name]
    metric_id = _native.METRIC_L2 if wl["metric"] == "l2" else _native.METRIC_COSINE
    ix = reuse or _native.NativeIndex(wl["dim"], metric=metric_id, device_id=0, capacity_rows=wl["rows"])
    saved = {o: ix.get_option(o) for o in (opts or {})}
    try:
        for o, v in (opts or {}).items():
            ix.set_option(o, v)
        if reuse is None:
            ix.fill_synthetic(SEED_CORPUS, 0, wl["rows"], normalize=True)
        k, batch = wl["k"], wl.get("batch", 1)
        nq = steps * batch
        dq = ix.device_queries_synthetic(SEED_QUERY, 0, nq, normalize=True)
        d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)

        def go(first, count):
            if batch > 1:
                for b in range(count):
                    ix.search_batch_device(dq, batch, k, d_idx, d_score, query_offset=(first + b) * batch)
            else:
                ix.search_device(dq, count, k, d_idx, d_score, query_offset=first)

        go(0, min(5, steps))
        ix.synchronize()
        ix.profile(True)
        ix.profile_read()
        ix.profile_read_gemm()
        ix.profile_read_sample()
        t0 = time.perf_counter()
        go(0, steps)
        ix.synchronize()
        el = time.perf_counter() - t0
        prof, gprof = ix.profile_read(), ix.profile_read_gemm()
        ix.profile(False)
        res = {"workload": wl["name"], "queries_per_s": nq / el, "steps": steps}
        if batch > 1:
            rl = batch_roofline(ix, wl, wl["rows"], gprof["gemm_ms"] / steps, k)
            res.update(bound=rl["bound"], kernel=rl["kernel"], achieved=rl["achieved"], unit=rl["unit"], frac=rl["frac"],
                       gemm_ms_per_batch=rl["gemm_ms_per_step"], ms_per_batch=el / steps * 1e3)
            for extra in ("bf16_mfma_frac", "i8_mfma_frac"):
                if extra in rl:
                    res[extra] = rl[extra]
            st = ix.batch_status(batch)  # the device entry point does not repair an overflowed candidate buffer
            res["overflowed_queries_last_batch"] = int(st["overflowed"])
            res["candidates_per_query"] = float(np.mean(st["counts"]))
            # the last timed batch's first queries and answers, for the oracle leg main() runs once the rows are on the host
            o = (steps - 1) * batch  # (every batch of this leg writes its results at the start of the output buffers)
            nver = min(8, batch)
            res["_verify"] = (dq.download(np.float32, (o + nver, ix.pitch))[o:, : wl["dim"]],
                              d_idx.download(np.int64, (nver, k)), d_score.download(np.float32, (nver, k)))
        else:
            rl = single_query_roofline(ix, wl, wl["rows"], k, prof, gprof, {}, "")
            lat = []
            for i in range(min(50, steps)):
                t1 = time.perf_counter()
                go(i, 1)
                ix.synchronize()
                lat.append(time.perf_counter() - t1)
            res.update(bound="hbm", kernel=rl["kernel"].split(" (")[0], kernel_ms=rl["avg_launch_ms"], achieved_GBps=rl["achieved"],
                       frac=rl["frac"], p50_ms=float(np.percentile(lat, 50) * 1e3))
        return res
    except Exception as e:  # an extra must never cost the main result
        return {"workload": wl["name"], "error": str(e)}
    finally:
        for o, v in saved.items():
            ix.set_option(o, v)
        if reuse is None:
            ix.close()


class FileRendezvous:
    """How the ranks that ONE launcher started on this node find each other without torch: a directory named after the
    launcher process (the ranks' common parent: pid + start time, so a recycled pid cannot match a stale directory) and
    the rendezvous port; rank 0 leaves the RCCL unique id there, the others wait for it.  Everything after that
    (barriers, reductions) goes through the RCCL communicator itself."""

    def __init__(self, rank: int, world: int):
        import tempfile

        self.rank, self.world = rank, world
        ppid = os.getppid()
        try:
            with open(f"/proc/{ppid}/stat") as f:
                start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time of the parent, in clock ticks
        except OSError:
            start = "0"
        tag = "_".join(str(x) for x in (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                         ppid, start))
        self.dir = Path(os.environ.get("WDBX_BENCH_RDZV_DIR") or Path(tempfile.gettempdir()) / f"wdbx_bench_{tag}")

    def share_unique_id(self, make) -> bytes:
        f = self.dir / "rccl_unique_id"
        if self.rank == 0:
            self.dir.mkdir(parents=True, exist_ok=True)
            uid = make()
            tmp = self.dir / "rccl_unique_id.tmp"
            tmp.write_bytes(uid)
            os.replace(tmp, f)
            return uid
        deadline = time.time() + 300
        while time.time() < deadline:
            if f.exists():
                uid = f.read_bytes()
                if len(uid) == _native.UNIQUE_ID_BYTES:
                    return uid
            time.sleep(0.02)
        raise RuntimeError(f"rank {self.rank}: no RCCL unique id appeared in {self.dir}")

    def cleanup(self) -> None:
        if self.rank == 0:
            import shutil

            shutil.rmtree(self.dir, ignore_errors=True)


class Watchdog:
    """A collective that never completes (a rank missing, a fabric problem) cannot be interrupted from Python: it would hold
    the GPU until the caller's limit kills the job.  `with Watchdog(seconds, what):` ends THIS process instead, with a message."""

    def __init__(self, seconds: float, what: str):
        import threading

        def fire():
            print(f"[bench] {what} did not complete within {seconds:.0f} s: giving up (exit 3)", file=sys.stderr, flush=True)
            os._exit(3)

        self.t = threading.Timer(seconds, fire)
        self.t.daemon = True

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


class RcclPlumbing:
    """barrier / max / min over the ranks through the shard's own RCCL communicator (no other transport)."""

    def __init__(self, ix, world):
        self.ix, self.world = ix, world

    def gather_f64(self, x: float):
        parts = self.ix.comm_allgather_host(np.float64(x).tobytes(), self.world)
        return [float(np.frombuffer(b, np.float64)[0]) for b in parts]

    def barrier(self) -> None:
        self.ix.synchronize()
        self.gather_f64(0.0)

    def max(self, x: float) -> float:
        return max(self.gather_f64(x))

    def min(self, x: float) -> float:
        return min(self.gather_f64(x))


class NoPlumbing:
    def __init__(self, ix):
        self.ix = ix

    def barrier(self) -> None:
        self.ix.synchronize()

    def max(self, x: float) -> float:
        return x

    min = max


def selection_dtype(ix, batch):
    """What the dominant kernel read: "u8" / "bf16" / "i8" shadow copies (selection only; every returned score is fp32
    from the fp32 rows) or "none" (the fp32 rows themselves)."""
    if batch > 1:
        return {0: "none", 1: "none (bf16 arithmetic on the fp32 rows)", 2: "bf16", 3: "i8"}.get(ix.get_option("last_gemm_family"), "none")
    return {0: "none", 1: "bf16", 2: "u8"}.get(ix.get_option("last_single_path"), "none")


def load_traffic_db():
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if tfile.exists():
        try:
            return json.loads(tfile.read_text())
        except Exception:
            return {}
    return {}


def main_group(args):
    """N GPUs driven by ONE process: the in-process shard group (wdbx_group_attach over one flat index per device,
    wdbx_group_search_resident) -- what `python bench.py --gpus N` runs when no launcher environment is present."""
    n_gpus = max(1, args.gpus)
    wl = dict(WORKLOADS[args.workload])
    if wl.get("batch", 1) > 1:
        sys.exit("the batched MFMA workload is a 1-GPU configuration (BASELINE configs[3])")
    if args.rows:
        wl["rows"] = args.rows
    k = args.k or wl["k"]
    metric_id = _native.METRIC_L2 if wl["metric"] == "l2" else _native.METRIC_COSINE
    ndev = _native.device_count()
    if ndev < 1:
        sys.exit("bench.py needs an AMD GPU (no CPU fallback exists)")
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(n_gpus))
    if len(devices) != n_gpus:
        sys.exit(f"--devices names {len(devices)} devices for --gpus {n_gpus}")
    if max(devices) >= ndev:
        sys.exit(f"--gpus {n_gpus} needs devices {devices} but {ndev} device(s) are visible")
    distinct = len(set(devices)) == len(devices)

    def ranges(scaling):
        if scaling == "strong":
            return [shard_row_range(wl["rows"], n_gpus, r) for r in range(n_gpus)]
        return [(r * wl["rows"], (r + 1) * wl["rows"]) for r in range(n_gpus)]

    spans = ranges(args.scaling)
    total_rows = spans[-1][1]
    shards = []
    for dev, (b, e) in zip(devices, spans):
        ix = _native.NativeIndex(wl["dim"], metric=metric_id, device_id=dev, capacity_rows=max(e - b, 1))
        for o in args.opt:
            name, v = o.split("=")
            ix.set_option(name, int(v))
        ix.fill_synthetic(SEED_CORPUS, b, e - b, normalize=True)  # ingest: untimed
        shards.append(ix)
    # one shard per device: the RCCL exchange.  If the communicators do not come up the run still produces its line --
    # over the device-copy exchange, and says so (`config.transport`, `rccl_nranks` 0): never a silent substitute
    with Watchdog(300, f"the shard group over devices {devices} (ncclCommInitAll + first search)"):
        try:
            grp = _native.NativeGroup.attach(shards, exchange=_native.NativeGroup.EXCHANGE_RCCL if distinct else _native.NativeGroup.EXCHANGE_COPY)
        except _native.HipBackendError as e:
            print(f"[bench] RCCL communicators over devices {devices} failed ({e}); exchanging by device copies", file=sys.stderr)
            grp = _native.NativeGroup.attach(shards, exchange=_native.NativeGroup.EXCHANGE_COPY)
        grp.set_row_bases([b for b, _ in spans])
        info = grp.info()
        nq_total = max(args.warmup + args.steps, args.latency_queries, 1)
        grp.queries_synthetic(SEED_QUERY, 0, nq_total, normalize=True)
        grp.search_resident(0, 1, k)   # (the first collective, under the watchdog; part of the warm-up)
        grp.synchronize()

    def run(first, count):
        if count > 0:
            grp.search_resident(first, count, k)

    run(0, args.warmup)
    grp.synchronize()
    for ix in shards:
        ix.profile(not args.no_profile)
        ix.profile_read(), ix.profile_read_gemm(), ix.profile_read_sample()
    grp.synchronize()
    x0 = grp.stat("exchanges")
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    grp.synchronize()
    elapsed = time.perf_counter() - t0
    exchanges_timed = grp.stat("exchanges") - x0
    profs = [(ix.profile_read(), ix.profile_read_gemm()) for ix in shards]
    res_idx, res_score = grp.results(args.steps, k) if args.steps > 0 else (np.zeros((1, k), np.int64), np.zeros((1, k), np.float32))
    for ix in shards:
        ix.profile(False)
    # second timed leg: the same K queries as K calls of ONE query, back to back, no host synchronisation between them --
    # one exchange (all-gather / device copies) + merge PER QUERY
    per_query = None
    if args.steps > 0:
        grp.synchronize()
        x1 = grp.stat("exchanges")
        t1 = time.perf_counter()
        for i in range(args.steps):
            grp.search_resident(args.warmup + i, 1, k)
        grp.synchronize()
        el1 = time.perf_counter() - t1
        last_idx, last_score = grp.results(1, k)
        per_query = {"queries": args.steps, "exchanges": grp.stat("exchanges") - x1, "queries_per_s": args.steps / el1,
                     "ms_per_query": el1 / args.steps * 1e3,
                     # (ids identical; scores to 1e-6: on shards under 131 072 rows a lone call runs the fp32 scan kernel, a
                     # call of many queries the selection scan + exact re-scoring -- two fp32 summation orders)
                     "last_result_equals_stream_leg": bool(np.array_equal(last_idx[0], res_idx[args.steps - 1])
                                                           and np.allclose(last_score[0], res_score[args.steps - 1], atol=1e-6, rtol=0))}
    if wl["metric"] == "cosine":
        assert np.all(np.diff(res_score, axis=1) <= 0), "scores not descending"
    else:
        assert np.all(np.diff(res_score, axis=1) >= 0), "distances not ascending"
    assert res_idx.max() < total_rows

    # the dominant kernel on every GPU (HIP events on each shard's own stream); the slowest shard bounds the step
    traffic_db = load_traffic_db()
    rls = [single_query_roofline(ix, wl, e - b, k, pr, gp, traffic_db, f"{args.workload}_n{n_gpus}_{args.scaling}")
           for ix, (b, e), (pr, gp) in zip(shards, spans, profs)]
    worst = max(range(n_gpus), key=lambda i: rls[i]["avg_launch_ms"])
    roofline = dict(rls[worst])
    roofline["per_gpu"] = [{"device": d, "rows": e - b, "avg_launch_ms": r["avg_launch_ms"], "achieved": r["achieved"], "frac": r["frac"]}
                           for d, (b, e), r in zip(devices, spans, rls)]
    roofline["note"] = "the slowest shard's kernel (it bounds the step); " + roofline.get("note", "")

    # latency: one query per call, results on the host (single client)
    lat = []
    for i in range(args.latency_queries):
        t1 = time.perf_counter()
        grp.search_resident(i, 1, k)
        grp.results(1, k)
        lat.append(time.perf_counter() - t1)
    # host-side cost of one group call: how long the enqueue alone takes (all shards' threads, exchange, merge launch)
    enq = []
    for i in range(min(50, args.latency_queries)):
        grp.synchronize()
        t1 = time.perf_counter()
        grp.search_resident(i, 1, k)
        enq.append(time.perf_counter() - t1)
    grp.synchronize()

    # the merged answer must equal a shard-by-shard host merge of the shards' own blocking searches
    sharded_check = None
    if args.steps > 0:
        try:
            ncheck = min(4, args.steps)
            dq = _native.DeviceBuffer(shards[0], (args.warmup + ncheck) * shards[0].pitch * 4)
            shards[0]._lib.wdbx_device_fill_synthetic(shards[0]._h, dq.ptr, SEED_QUERY, 0, args.warmup + ncheck, 1)
            qh = dq.download(np.float32, (args.warmup + ncheck, shards[0].pitch))[args.warmup:, : wl["dim"]]
            dq.free()
            per = [ix.search(qh, k) if ix.size() else (np.full((ncheck, k), -1, np.int64), np.zeros((ncheck, k), np.float32))
                   for ix in shards]
            ok = True
            for q in range(ncheck):
                idxs = [np.where(p[0][q] >= 0, p[0][q] + b, -1) for p, (b, _) in zip(per, spans)]
                h_idx, h_score = merge_topk(idxs, [p[1][q] for p in per], k, metric_id)
                ok &= bool(np.array_equal(h_idx, res_idx[q]) and np.allclose(h_score, res_score[q], atol=1e-6, rtol=0))
            sharded_check = "ok" if ok else "MISMATCH"
        except Exception as e:  # report, never hide
            sharded_check = f"error: {e}"

    # strong scaling: also BASELINE configs[4]-style WEAK scaling (every GPU keeps the workload's full row count:
    # 8 x 10M = 80M rows at N=8) as an extra on the same group, outside the timed region
    weak_extra = None
    if n_gpus > 1 and args.scaling == "strong":
        try:
            wspans = ranges("weak")
            for ix, (b, e) in zip(shards, wspans):
                ix.clear()
                ix.fill_synthetic(SEED_CORPUS, b, e - b, normalize=True)
            grp.set_row_bases([b for b, _ in wspans])
            nw = min(100, args.steps)
            grp.search_resident(0, min(10, nw), k)
            grp.synchronize()
            tw = time.perf_counter()
            grp.search_resident(0, nw, k)
            grp.synchronize()
            tw = time.perf_counter() - tw
            widx, _ = grp.results(nw, k)
            weak_extra = {"workload": f"{n_gpus * wl['rows']} x {wl['dim']} fp32 over {n_gpus} shards (weak: {wl['rows']} rows/GPU)",
                          "scaling": "weak", "queries": nw, "queries_per_s": nw / tw, "ms_per_query": tw / nw * 1e3,
                          "rows_scanned_per_s": n_gpus * wl["rows"] * nw / tw,
                          "aggregate_GBps_fp32_equivalent": n_gpus * wl["rows"] * wl["dim"] * 4 * nw / tw / 1e9,
                          "results_span_shards": int(len(np.unique(widx // wl["rows"])))}
        except Exception as e:  # an extra must never cost the main result
            weak_extra = {"error": str(e)}

    sel = selection_dtype(shards[0], 1)
    bytes_resident = [ix.get_option("device_bytes_resident") for ix in shards]
    grp.close()
    for ix in shards:
        ix.close()

    facade = None
    if not args.no_facade and not args.rows:
        facade = facade_latency(wl, k, n_queries=100, num_shards=n_gpus, devices=devices)

    out = {
        "metric": "queries/sec (single-query brute-force top-k scans, whole job)",
        "value": args.steps / elapsed,
        "unit": "queries/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "selection_dtype": sel,
        "data": "synthetic",
        "config": {
            "workload": wl["name"] + f", {n_gpus} shard(s) in one process, " +
                        ("RCCL all-gather merge" if info["rccl_nranks"] else "device-copy exchange (shards share a GPU)") +
                        f"; value = the {args.steps} resident queries enqueued as ONE call ({exchanges_timed} exchange(s) in the timed "
                        "region); per_query_exchange = the same queries one call each",
            "rows_total": total_rows,
            "rows_per_gpu": spans[0][1] - spans[0][0],
            "dim": wl["dim"],
            "metric": wl["metric"],
            "k": k,
            "queries_per_step": 1,
            "parallelism": f"shards{n_gpus}",
            "driver": "in-process shard group (wdbx_group_attach + wdbx_group_search_resident; one host thread per shard)",
            "transport": "rccl" if info["rccl_nranks"] else "device_copies",
            "devices": devices,
            "rccl_nranks": info["rccl_nranks"],
        },
        "roofline": roofline,
        "latency_ms": {
            "p50": float(np.percentile(lat, 50) * 1e3) if lat else None,
            "p99": float(np.percentile(lat, 99) * 1e3) if lat else None,
            "single_client_qps": float(1.0 / np.median(lat)) if lat else None,
            "host_enqueue_p50": float(np.percentile(enq, 50) * 1e3) if enq else None,
        },
        "rows_scanned_per_s": total_rows * args.steps / elapsed,
        "exchanges_in_timed_region": exchanges_timed,
        "per_query_exchange": per_query,
        "device_bytes_resident": {"per_shard": bytes_resident, "total": int(sum(bytes_resident)),
                                  "fp32_rows_total": int(total_rows) * wl["dim"] * 4},
        "sharded_check": sharded_check,
        "rccl": {"rccl_nranks": info["rccl_nranks"], "shards": info["shards"], "communicator": "ncclCommInitAll (one process)"},
        "weak_scaling_extra": weak_extra,
        "timed_region_profiled": not args.no_profile,
        "facade_latency_ms": facade,
        "cpu_baseline": None,  # (timed at N = 1 only)
    }
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or args.mode == "group"):
        return main_group(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    rdzv = None
    # WDBX_BENCH_FORCE_GROUP=1 under `torch.distributed.run --nproc-per-node 1` rehearses the whole
    # N > 1 code path (unique id, RCCL communicator, sharded search, self-check) with one rank
    grouped = world > 1 or bool(os.environ.get("WDBX_BENCH_FORCE_GROUP"))

    wl = dict(WORKLOADS[args.workload])
    if args.rows:
        wl["rows"] = args.rows
    k = args.k or wl["k"]
    metric_id = _native.METRIC_L2 if wl["metric"] == "l2" else _native.METRIC_COSINE

    if args.scaling == "strong":
        begin, end = shard_row_range(wl["rows"], world, rank)
        total_rows = wl["rows"]
    else:
        begin, end = rank * wl["rows"], (rank + 1) * wl["rows"]
        total_rows = wl["rows"] * world
    local_rows = end - begin

    ndev = _native.device_count()
    if ndev < 1:
        sys.exit("bench.py needs an AMD GPU (no CPU fallback exists)")
    device_id = local_rank if local_rank < ndev else 0  # a launcher may expose one device per rank
    ix = _native.NativeIndex(wl["dim"], metric=metric_id, device_id=device_id, capacity_rows=max(local_rows, 1))
    for o in args.opt:
        name, v = o.split("=")
        ix.set_option(name, int(v))
    ix.fill_synthetic(SEED_CORPUS, begin, local_rows, normalize=True)  # ingest: untimed

    transport = "none"
    group = None
    plumb = NoPlumbing(ix)
    if grouped:
        transport = "rccl"
        # no torch: the unique id through the launcher's rendezvous directory, everything else through RCCL.  A rank
        # that cannot join leaves the others waiting inside ncclCommInitRank -- the watchdog ends this process with a
        # message that names the stage.
        rdzv = FileRendezvous(rank, world)
        group = ShardGroup(rank, world, begin, metric_id, local_index=ix)
        with Watchdog(300, f"rank {rank}: the RCCL communicator over {world} ranks (unique id + ncclCommInitRank + first barrier)"):
            uid = rdzv.share_unique_id(_native.NativeIndex.comm_unique_id)
            print(f"[bench] rank {rank}/{world}: unique id shared, entering ncclCommInitRank on device {device_id}", file=sys.stderr, flush=True)
            group.init_rccl(uid)
            plumb = RcclPlumbing(ix, world)
            plumb.barrier()
            print(f"[bench] rank {rank}/{world}: communicator up, first barrier passed", file=sys.stderr, flush=True)

    batch = wl.get("batch", 1)  # queries per step
    if batch > 1 and grouped:
        sys.exit("the batched MFMA workload is a 1-GPU configuration (BASELINE configs[3])")
    nq_total = (args.warmup + args.steps) * batch
    dq = ix.device_queries_synthetic(SEED_QUERY, 0, max(nq_total, args.latency_queries, 1), normalize=True)
    d_idx = ix.alloc(max(nq_total, 1) * k * 8)
    d_score = ix.alloc(max(nq_total, 1) * k * 4)

    barrier = plumb.barrier

    def run(first, count):
        if count <= 0:
            return
        if batch > 1:
            for b in range(count):
                o = (first + b) * batch
                ix._lib.wdbx_index_search_batch_device(ix._h, dq.ptr + o * ix.pitch * 4, batch, k,
                                                       d_idx.ptr + o * k * 8, d_score.ptr + o * k * 4)
            return
        if grouped:
            group.search_device(dq, count, k, d_idx, d_score, query_offset=first)
        else:
            ix.search_device(dq, count, k, d_idx, d_score, query_offset=first)

    run(0, args.warmup)
    barrier()
    ix.profile(not args.no_profile)
    ix.profile_read()
    ix.profile_read_gemm()
    ix.profile_read_sample()
    barrier()
    x0 = ix.get_option("exchanges")
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    ix.synchronize()
    elapsed_local = time.perf_counter() - t0
    exchanges_timed = ix.get_option("exchanges") - x0
    prof = ix.profile_read()
    gprof = ix.profile_read_gemm()
    sample_qn = ix.get_option("last_sample_qn")  # (of the timed rounds: the lone latency queries below reset it)
    ix.profile(False)
    # every rank started behind the same barrier and synchronised its own device: the job's time is the slowest rank's
    elapsed = plumb.max(elapsed_local)

    # the TIMED output, before anything else reuses the buffers: sanity here (sorted, in range); parity against the
    # oracle in verify_timed_queries below and, for every path and edge case, in tests/
    all_idx = d_idx.download(np.int64, (max(nq_total, 1), k))
    all_score = d_score.download(np.float32, (max(nq_total, 1), k))
    lo = args.warmup * batch if batch > 1 else 0  # (single queries write their results from the buffer's start)
    res_idx, res_score = all_idx[lo:lo + max(args.steps * batch, 1)], all_score[lo:lo + max(args.steps * batch, 1)]
    del all_idx, all_score
    if wl["metric"] == "cosine":
        assert np.all(np.diff(res_score, axis=1) <= 0), "scores not descending"
    else:
        assert np.all(np.diff(res_score, axis=1) >= 0), "distances not ascending"
    assert res_idx.max() < total_rows

    # N > 1, second timed leg: the same K queries as K calls of ONE query each, back to back, no host synchronisation
    # between them: one all-gather + merge PER QUERY (every rank issues the same sequence of collectives)
    per_query = None
    if grouped and batch == 1 and args.steps > 0:
        barrier()
        x1 = ix.get_option("exchanges")
        t1 = time.perf_counter()
        for i in range(args.steps):
            run(args.warmup + i, 1)
        ix.synchronize()
        el1 = plumb.max(time.perf_counter() - t1)
        last_idx = d_idx.download(np.int64, (1, k))
        per_query = {"queries": args.steps, "exchanges": ix.get_option("exchanges") - x1, "queries_per_s": args.steps / el1,
                     "ms_per_query": el1 / args.steps * 1e3,
                     "last_result_equals_stream_leg": bool(np.array_equal(last_idx[0], res_idx[args.steps - 1]))}

    # latency: one query at a time, host-synchronised (single client)
    lat = []
    for i in range(args.latency_queries if batch == 1 else 0):
        t1 = time.perf_counter()
        run(i, 1)
        ix.synchronize()
        lat.append(time.perf_counter() - t1)
    barrier()

    # N > 1: every rank must hold the same merged answer, and it must equal the host-side exchange
    sharded_check = None
    if grouped and args.steps > 0:
        try:
            ncheck = min(4, args.steps)
            qh = dq.download(np.float32, (nq_total, ix.pitch))[args.warmup:args.warmup + ncheck, : wl["dim"]]
            run(args.warmup, ncheck)
            barrier()
            r_idx = d_idx.download(np.int64, (ncheck, k))
            r_score = d_score.download(np.float32, (ncheck, k))
            h_idx, h_score = group.search(qh, k)  # blocking local searches + host-side merge of the all-gathered records
            same = bool(np.array_equal(r_idx, h_idx) and np.allclose(r_score, h_score, atol=1e-6, rtol=0))
            sharded_check = "ok" if plumb.min(1.0 if same else 0.0) == 1.0 else "MISMATCH"
        except Exception as e:  # report, never hide
            sharded_check = f"error: {e}"

    # what RCCL itself reports for the communicator the timed region used (evidence that the exchange spanned N ranks)
    comm_info = ix.comm_info() if grouped else None

    # N > 1, strong scaling: also measure BASELINE configs[4]-style WEAK scaling (every rank keeps the workload's full
    # row count: 8 x 10M = 80M rows at N=8) as an extra, outside the timed region.  Same communicator: the shard is
    # refilled with this rank's range of the larger corpus and re-based (wdbx_index_comm_set_row_base).
    weak_extra = None
    if grouped and args.scaling == "strong" and batch == 1:
        try:
            ix.clear()
            wbegin = rank * wl["rows"]
            ix.fill_synthetic(SEED_CORPUS, wbegin, wl["rows"], normalize=True)
            ix.comm_set_row_base(wbegin)
            nw = min(100, args.steps)
            ix.search_device(dq, min(10, nw), k, d_idx, d_score, sharded=True)
            barrier()
            tw = time.perf_counter()
            ix.search_device(dq, nw, k, d_idx, d_score, sharded=True)
            ix.synchronize()
            tw = plumb.max(time.perf_counter() - tw)
            widx = d_idx.download(np.int64, (nw, k))
            weak_extra = {"workload": f"{world * wl['rows']} x {wl['dim']} fp32 over {world} shards (weak: {wl['rows']} rows/GPU)",
                          "scaling": "weak", "queries": nw, "queries_per_s": nw / tw, "ms_per_query": tw / nw * 1e3,
                          "rows_scanned_per_s": world * wl["rows"] * nw / tw,
                          "aggregate_GBps_fp32_equivalent": world * wl["rows"] * wl["dim"] * 4 * nw / tw / 1e9,
                          "results_span_shards": int(len(np.unique(widx // wl["rows"])))}
        except Exception as e:  # an extra must never cost the main result
            weak_extra = {"error": str(e)}

    alg_bytes = local_rows * wl["dim"] * 4  # fp32 corpus bytes of this rank (for the effective-rate fields)
    traffic_db = load_traffic_db()

    if batch > 1:
        roofline = batch_roofline(ix, wl, local_rows, gprof["gemm_ms"] / max(args.steps, 1), k, traffic_db)
        roofline["launches_timed"] = gprof["gemm_launches"]
        roofline["corpus_GBps_effective"] = alg_bytes / (elapsed / max(args.steps, 1)) / 1e9
        # the device entry point leaves an overflowed candidate buffer to the caller: an overflowed query's top-k may be
        # incomplete, so the timed batches only count when none overflowed (last timed batch checked here)
        st = ix.batch_status(batch)
        roofline["overflowed_queries_last_batch"] = int(st["overflowed"])
        roofline["candidates_per_query"] = float(np.mean(st["counts"]))
        assert st["overflowed"] == 0, "candidate buffers overflowed in the timed region: results incomplete"
    else:
        roofline = None
    sel = selection_dtype(ix, batch)
    if roofline is None:
        roofline = single_query_roofline(ix, wl, local_rows, k, prof, gprof, traffic_db, f"{args.workload}_n{world}_{args.scaling}",
                                         sample_qn=sample_qn)
    if batch > 1:
        family = {3: "gemm_i8_kernel", 2: "gemm_bf16w8_kernel", 1: "gemm_bf16w8_kernel"}.get(ix.get_option("last_gemm_family"), "gemm_topk_kernel")
    else:
        family = {2: "scan8_kernel", 1: "gemm_bf16w8_kernel"}.get(ix.get_option("last_single_path"), "scan_kernel")
    out = {
        "metric": "queries/sec (single-query brute-force top-k scans, whole job)" if batch == 1 else
                  "queries/sec (256-query batches, one matrix-core pass per batch, whole job)",
        "value": args.steps * batch / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",  # every returned score is fp32 arithmetic on the fp32 rows
        "selection_dtype": sel,  # what the dominant (selection) kernel read: a u8 / bf16 / i8 shadow copy, or "none"
        "data": "synthetic",
        "config": {
            "workload": wl["name"] + (f", {world} shards, RCCL all-gather merge; value = the {args.steps} resident queries enqueued as "
                                      f"ONE call ({exchanges_timed} exchange(s) in the timed region); per_query_exchange = the same "
                                      "queries one call each" if grouped else ", 1 shard"),
            "rows_total": total_rows,
            "rows_per_gpu": local_rows,
            "dim": wl["dim"],
            "metric": wl["metric"],
            "k": k,
            "queries_per_step": batch,
            "parallelism": f"shards{world}",
            "driver": "one process per GPU (ncclCommInitRank)" if grouped else "one flat index",
            "transport": transport,
        },
        "roofline": roofline,
        "latency_ms": {
            "p50": float(np.percentile(lat, 50) * 1e3) if lat else None,
            "p99": float(np.percentile(lat, 99) * 1e3) if lat else None,
            "single_client_qps": float(1.0 / np.median(lat)) if lat else None,
        },
        "rows_scanned_per_s": total_rows * args.steps / elapsed,
        "exchanges_in_timed_region": exchanges_timed if grouped else None,
        "per_query_exchange": per_query,
        "device_bytes_resident": None,  # (filled below, after every leg has built the copies it uses)
        "sharded_check": sharded_check,
        "rccl": comm_info,  # {"rccl_nranks", "rccl_rank", "row_base"} from ncclCommCount / ncclCommUserRank; null at N = 1
        "weak_scaling_extra": weak_extra,
        "timed_region_profiled": not args.no_profile,
    }
    out["config"]["rccl_nranks"] = comm_info["rccl_nranks"] if comm_info else None
    if rank == 0 and world == 1 and not grouped and args.workload == "t" and not args.no_other_configs:
        # the other BASELINE configs, measured briefly in the same run (they are parity-test cases, not the
        # bench line; exact parity for each lives in tests/test_gpu_parity.py)
        out["other_configs"] = {
            # the SAME corpus and queries on the fp32 scan kernel: SURVEY 8(d)'s literal roofline, N*d*4 bytes per query
            "t_fp32_scan": quick_config("t", reuse=ix, steps=20, opts={"scan_shadow": 0}),
            "c4": quick_config("c4", reuse=ix, steps=60)}
        for name, st in (("c1", 500), ("c2", 200), ("c3", 40)):
            out["other_configs"][name] = quick_config(name, steps=st)
        # SURVEY 8(d) read literally (N*d*4 bytes per query) is the fp32 scan kernel's roofline on this same corpus
        fp = out["other_configs"]["t_fp32_scan"]
        if "frac" in fp and batch == 1:
            out["roofline"]["frac_fp32_rows_kernel"] = fp["frac"]
            out["roofline"]["fp32_rows_kernel"] = {"kernel": "scan_kernel (fp32 rows, scan_shadow=0)", "avg_launch_ms": fp["kernel_ms"],
                                                   "achieved": fp["achieved_GBps"], "algorithmic_bytes_per_launch": local_rows * wl["dim"] * 4,
                                                   "queries_per_s": fp["queries_per_s"]}
    c4_verify = (out.get("other_configs", {}).get("c4") or {}).pop("_verify", None)
    if rank == 0 and world == 1 and not grouped and batch == 1 and not args.no_facade and not args.rows:
        out["facade_latency_ms"] = facade_latency(wl, k)
    if rank == 0 and world == 1 and not grouped and (args.verify > 0 or not args.no_cpu_baseline):
        rows_h, whole = host_rows(ix, wl["rows"])
        if args.verify > 0 and args.steps > 0:
            nver = min(args.verify, args.steps * batch)
            qh = dq.download(np.float32, (args.warmup * batch + nver, ix.pitch))[args.warmup * batch:, : wl["dim"]]
            out["parity"] = verify_timed_queries(rows_h, whole, ix, qh, res_idx[:nver], res_score[:nver], k, metric_id)
        if c4_verify is not None:  # the c4 leg's last timed batch against the oracle (same corpus: it reuses this index)
            out["other_configs"]["c4"]["parity"] = verify_timed_queries(rows_h, whole, ix, c4_verify[0], c4_verify[1], c4_verify[2],
                                                                        WORKLOADS["c4"]["k"], metric_id)
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(rows_h, whole, wl, k, metric_id, args.cpu_seconds)
        del rows_h
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and not grouped and not args.no_live_traffic:
        # roofline.traffic of this very command line: two PMC child passes (children, never an exec), after every timed leg so
        # that nothing they do to clocks or temperatures touches a timed region
        live, live_err = live_traffic(args, batch)
        apply_live_traffic(out["roofline"], live, live_err, family)
    # what the handle holds on the device (fp32 rows + every shadow copy a leg of this run built + scratch): the memory
    # price of the selection paths is on the record next to their speed
    out["device_bytes_resident"] = {"total": ix.get_option("device_bytes_resident"), "fp32_rows": int(local_rows) * wl["dim"] * 4,
                                    "u8_shadow": ix.get_option("shadow8_bytes"), "i8_shadow": ix.get_option("shadowg_bytes"),
                                    "bf16_shadow": ix.get_option("shadow_bytes")}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if grouped:
        barrier()  # nobody tears the communicator down under a rank that is still inside a collective
    ix.close()
    if rdzv is not None:
        rdzv.cleanup()


if __name__ == "__main__":
    main()
