#!/usr/bin/env python3
"""Staged preflight of everything the N > 1 GPU path does that a 1-GPU box can never execute.

    python tools/preflight_multigpu.py [--devices 0,1,...] [--out gpurun_out/preflight_multigpu.json]

Plain `python`, no launcher, < 60 s on an 8-GPU node.  Every stage runs in a CHILD process under its own time limit and
prints one line; a stage that hangs or dies is recorded under its NAME (what it was doing, its last output) instead of
surfacing later as a watchdog's `exit 3` in the middle of the scaling bench.  The stages are exactly the code that has
never run with more than one rank (missing #1):

  devices        device count, names, memory, the peer-access matrix (hipDeviceCanAccessPeer)
  comm_init_all  one flat index per device + ncclCommInitAll over them (wdbx_group_attach_ex, exchange = RCCL or fail)
  group_rccl     S host threads x ncclAllGather: lone query, a 256-query call, a masked call, the resident entry point --
                 each compared with the single-index answer over the same rows (ids equal, scores to 1e-6)
  group_copy     the same group over peer-access / mapped-staging exchange (exchange = device copies) -- the fallback the
                 bench takes when the communicators do not come up
  per_process    N fresh child processes, one per device, through bench.py's own launcher path: FileRendezvous (the RCCL
                 unique id through a directory named after the common parent) -> ncclCommInitRank -> sharded searches ->
                 `sharded_check`; the children's JSON line is parsed

With fewer than two distinct devices (`--devices 0,0`: two shards on one GPU) the RCCL stages are reported as `skipped`
and the copy-exchange and single-rank launcher stages still run, so the tool itself is exercised on a 1-GPU box.
The reference shape all of this replaces: the shard loop + list.sort merge of wdbx/core/vector_store.py:323-345.
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wdbx-py_amd"))

ROWS_PER_SHARD = 300_000   # above the selection scan's 196 608-row floor for lone queries, small enough for seconds
DIM, K = 384, 10
SEED_CORPUS, SEED_QUERY = 0xC0FFEE, 0xBEEF


def _say(msg):
    print(msg, flush=True)


# ------------------------------------------------------------------------------------------------ stages (run in children)
def stage_devices(devs):
    import ctypes

    from wdbx_amd import _native

    n = _native.device_count()
    out = {"visible_devices": n, "asked_for": devs}
    if n < 1:
        raise RuntimeError("no AMD GPU visible")
    if max(devs) >= n:
        raise RuntimeError(f"devices {devs} asked for, {n} visible")
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        distinct = sorted(set(devs))
        matrix = {}
        for a in distinct:
            for b in distinct:
                if a != b:
                    can = ctypes.c_int(0)
                    rc = hip.hipDeviceCanAccessPeer(ctypes.byref(can), a, b)
                    matrix[f"{a}->{b}"] = int(can.value) if rc == 0 else f"error {rc}"
        out["peer_access"] = matrix
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        mem = {}
        for a in distinct:
            if hip.hipSetDevice(a) == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0:
                mem[str(a)] = {"free_GiB": round(free.value / 2**30, 1), "total_GiB": round(total.value / 2**30, 1)}
        out["memory"] = mem
    except OSError as e:
        out["peer_access"] = f"libamdhip64.so not loadable from Python: {e}"
    return out


def _make_shards(devs, native):
    shards, bases = [], []
    for s, dev in enumerate(devs):
        ix = native.NativeIndex(DIM, device_id=dev, capacity_rows=ROWS_PER_SHARD)
        ix.fill_synthetic(SEED_CORPUS, s * ROWS_PER_SHARD, ROWS_PER_SHARD, normalize=True)
        shards.append(ix)
        bases.append(s * ROWS_PER_SHARD)
    return shards, bases


def _group_checks(devs, exchange, out):
    """The group over `devs` with the given exchange against ONE index holding all the rows: lone query, 256-query call,
    masked call, resident entry point."""
    import numpy as np

    from wdbx_amd import _native as native

    t0 = time.time()
    shards, bases = _make_shards(devs, native)
    out["fill_s"] = round(time.time() - t0, 2)
    whole = native.NativeIndex(DIM, device_id=devs[0], capacity_rows=ROWS_PER_SHARD * len(devs))
    whole.fill_synthetic(SEED_CORPUS, 0, ROWS_PER_SHARD * len(devs), normalize=True)
    t0 = time.time()
    grp = native.NativeGroup.attach(shards, exchange=exchange)
    out["attach_s"] = round(time.time() - t0, 2)
    try:
        grp.set_row_bases(bases)
        info = grp.info()
        out["info"] = info
        if exchange == native.NativeGroup.EXCHANGE_RCCL and info["rccl_nranks"] != len(devs):
            raise RuntimeError(f"ncclCommCount says {info['rccl_nranks']} ranks for {len(devs)} shards")
        dq = whole.device_queries_synthetic(SEED_QUERY, 0, 256, normalize=True)
        queries = dq.download(np.float32, (256, whole.pitch))[:, :DIM].copy()

        def same(got, want, what):
            ok = bool(np.array_equal(got[0], want[0]) and np.allclose(got[1], want[1], atol=1e-6, rtol=0))
            out[what] = "ok" if ok else "MISMATCH"
            if not ok:
                bad = int(np.argmax(np.any(got[0] != want[0], axis=1)))
                raise RuntimeError(f"{what}: group answer differs from the single index (first at query {bad}: "
                                   f"{got[0][bad].tolist()} vs {want[0][bad].tolist()})")

        t0 = time.time()
        same(grp.search(queries[0], K), whole.search(queries[0], K), "lone_query")
        out["first_search_s"] = round(time.time() - t0, 2)
        lat = []
        for i in range(1, 9):
            t1 = time.perf_counter()
            got = grp.search(queries[i], K)
            lat.append(time.perf_counter() - t1)
            same(got, whole.search(queries[i], K), "lone_query")
        out["lone_query_ms_p50"] = round(sorted(lat)[len(lat) // 2] * 1e3, 3)
        same(grp.search(queries, K), whole.search(queries, K), "call_of_256_queries")
        # a row mask per shard (metadata filter push-down): every third row allowed
        n_all = ROWS_PER_SHARD * len(devs)
        allowed = (np.arange(n_all) % 3) == 0
        masks = [native.pack_row_mask(allowed[b:b + ROWS_PER_SHARD]) for b in bases]
        got = grp.search_merged(queries[:4], K, K, mask_words=masks)
        want = whole.search(queries[:4], K, mask_words=native.pack_row_mask(allowed))
        same(got, want, "masked_call")
        # the resident entry point bench.py drives: one call of 40 queries, then 8 calls of one
        grp.queries_synthetic(SEED_QUERY, 0, 64, normalize=True)
        x0 = grp.stat("exchanges")
        grp.search_resident(0, 40, K)
        grp.synchronize()
        out["exchanges_for_one_call_of_40"] = grp.stat("exchanges") - x0
        same(grp.results(40, K), _single(whole, queries[:40]), "resident_stream")  # (one scan per query, like single calls)
        for i in range(8):
            grp.search_resident(40 + i, 1, K)
        grp.synchronize()
        same(grp.results(1, K), whole.search(queries[47], K), "resident_one_query_calls")
    finally:
        grp.close()
        for ix in shards:
            ix.close()
        whole.close()
    return out


def _single(ix, queries):
    import numpy as np

    res = [ix.search(q, K) for q in queries]
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def stage_comm_init_all(devs):
    from wdbx_amd import _native as native

    shards, _ = _make_shards(devs, native)
    try:
        t0 = time.time()
        grp = native.NativeGroup.attach(shards, exchange=native.NativeGroup.EXCHANGE_RCCL)
        dt = time.time() - t0
        info = grp.info()
        grp.close()
    finally:
        for ix in shards:
            ix.close()
    if info["rccl_nranks"] != len(devs):
        raise RuntimeError(f"ncclCommCount says {info['rccl_nranks']} ranks for {len(devs)} shards")
    return {"ncclCommInitAll_s": round(dt, 2), "info": info}


def stage_group_rccl(devs):
    from wdbx_amd import _native as native

    return _group_checks(devs, native.NativeGroup.EXCHANGE_RCCL, {})


def stage_group_copy(devs):
    from wdbx_amd import _native as native

    return _group_checks(devs, native.NativeGroup.EXCHANGE_COPY, {})


STAGES = {"devices": stage_devices, "comm_init_all": stage_comm_init_all, "group_rccl": stage_group_rccl,
          "group_copy": stage_group_copy}


# ------------------------------------------------------------------------------------------------ the driver
def run_child(stage, devs, limit):
    cmd = [sys.executable, str(Path(__file__).resolve()), "--stage", stage, "--devices", ",".join(map(str, devs))]
    t0 = time.time()
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True)
    except subprocess.TimeoutExpired as e:
        tail = ((e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""))[-400:]
        err = ((e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))[-400:]
        return {"status": "HUNG", "seconds": round(time.time() - t0, 1), "limit_s": limit, "stdout_tail": tail, "stderr_tail": err}
    rec = {"status": "ok" if p.returncode == 0 else "FAILED", "seconds": round(time.time() - t0, 1), "rc": p.returncode}
    for line in p.stdout.splitlines():
        if line.startswith("RESULT "):
            rec["result"] = json.loads(line[7:])
    if p.returncode != 0:
        rec["stderr_tail"] = p.stderr[-600:]
        rec["stdout_tail"] = p.stdout[-300:]
    return rec


def stage_per_process(devs, limit):
    """bench.py under a launcher's environment, one fresh process per device (this process plays the launcher: the ranks'
    common parent, which names the rendezvous directory)."""
    world = len(devs)
    port = str(29600 + os.getpid() % 300)
    procs = []
    t0 = time.time()
    for rank, dev in enumerate(devs):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(dev), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        if world == 1:
            env["WDBX_BENCH_FORCE_GROUP"] = "1"
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "16", "--warmup", "4",
               "--rows", str(ROWS_PER_SHARD * world), "--latency-queries", "8", "--no-cpu-baseline", "--no-facade",
               "--no-other-configs", "--no-live-traffic", "--verify", "0"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    rec = {"world": world, "ranks": []}
    status = "ok"
    for rank, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=max(1.0, limit - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            status = "HUNG"
            rec["ranks"].append({"rank": rank, "status": "HUNG (killed)", "stderr_tail": se[-500:]})
            continue
        r = {"rank": rank, "rc": p.returncode, "stderr_tail": se[-400:]}
        if p.returncode != 0:
            status = "FAILED" if status == "ok" else status
        if rank == 0 and p.returncode == 0:
            try:
                line = json.loads(so.strip().splitlines()[-1])
                r["line"] = {k: line.get(k) for k in ("value", "n_gpus", "sharded_check", "exchanges_in_timed_region", "per_query_exchange",
                                                       "rccl")}
                if line.get("sharded_check") != "ok" or (line.get("rccl") or {}).get("rccl_nranks") != world:
                    status = "FAILED"
            except Exception as e:
                r["parse_error"] = str(e)
                status = "FAILED"
        rec["ranks"].append(r)
    rec["status"] = status
    rec["seconds"] = round(time.time() - t0, 1)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="")
    ap.add_argument("--stage", default="")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "preflight_multigpu.json"))
    ap.add_argument("--limit", type=float, default=120.0, help="seconds per stage")
    args = ap.parse_args()
    if args.stage:  # a child: run one stage, print its result
        devs = [int(d) for d in args.devices.split(",")]
        res = STAGES[args.stage](devs)
        print("RESULT " + json.dumps(res), flush=True)
        return 0

    from wdbx_amd import _native

    ndev = _native.device_count()
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(max(ndev, 1)))
    distinct = len(set(devs)) == len(devs) and len(devs) >= 2
    report = {"devices": devs, "visible_devices": ndev, "stages": {}}
    t_all = time.time()

    def stage(name, fn):
        _say(f"[preflight] {name} ...")
        rec = fn()
        report["stages"][name] = rec
        _say(f"[preflight] {name}: {rec['status']} ({rec.get('seconds', 0)} s)" +
             ("" if rec["status"] in ("ok", "skipped") else f"  <-- {rec.get('stderr_tail', rec.get('reason', ''))[-300:]!r}"))
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(report, indent=1) + "\n")   # (after every stage: a dying run leaves its record)
        return rec["status"] == "ok"

    stage("devices", lambda: run_child("devices", devs, args.limit))
    skip = {"status": "skipped", "reason": "needs >= 2 distinct devices (RCCL takes one rank per device)"}
    stage("comm_init_all", (lambda: run_child("comm_init_all", devs, args.limit)) if distinct else (lambda: dict(skip)))
    stage("group_rccl", (lambda: run_child("group_rccl", devs, args.limit)) if distinct else (lambda: dict(skip)))
    if len(devs) >= 2:
        stage("group_copy", lambda: run_child("group_copy", devs, args.limit))
    # the launcher path: all distinct devices as ranks; on shared devices one rank (a 1-rank communicator)
    stage("per_process", lambda: stage_per_process(devs if distinct else devs[:1], args.limit))
    bad = [n for n, r in report["stages"].items() if r["status"] not in ("ok", "skipped")]
    report["total_seconds"] = round(time.time() - t_all, 1)
    report["verdict"] = "ok" if not bad else "FAILED at: " + ", ".join(bad)
    Path(args.out).write_text(json.dumps(report, indent=1) + "\n")
    _say(f"[preflight] {report['verdict']} in {report['total_seconds']} s -> {args.out}")
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main())
