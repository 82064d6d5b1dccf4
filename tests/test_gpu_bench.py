"""bench.py as the driver starts it, on small corpora: the JSON contract of every driver (plain index, in-process shard
group with 1 RCCL rank, several shards on one GPU, one rank under the real launcher without torch in the worker)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
COMMON = ["--rows", "400000", "--steps", "24", "--warmup", "6", "--latency-queries", "8", "--no-facade", "--no-other-configs",
          "--no-cpu-baseline"]


def _run(cmd, env=None, timeout=300):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _check_contract(d, n_gpus):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == 24 and d["warmup"] == 6 and d["unit"] == "queries/s" and d["value"] > 0
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(d["ms_per_step"] * d["value"] / 1e3 - 1.0) < 1e-6


def test_plain_index_line():
    d = _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON + ["--verify", "2", "--traffic-steps", "8"], timeout=900)
    _check_contract(d, 1)
    assert d["selection_dtype"] == "u8" and d["parity"]["parity_check"] == "ok" and d["config"]["transport"] == "none"
    assert d["exchanges_in_timed_region"] is None and d["per_query_exchange"] is None     # (one shard: nothing to exchange)
    b = d["device_bytes_resident"]
    assert b["fp32_rows"] == 400_000 * 384 * 4 and b["u8_shadow"] > 400_000 * 384 and b["total"] >= b["fp32_rows"] + b["u8_shadow"]
    # roofline.traffic is measured for this very command line: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE)
    r = d["roofline"]
    assert "traffic_live_error" not in r, r.get("traffic_live_error")
    assert r["traffic_source"].startswith("live: rocprofv3 --pmc") and r["traffic_detail"]["family"] == "scan8_kernel"
    # (400 k x 384 bytes sit in the 256 MiB Infinity Cache between queries: the memory-side counters may see less than the
    # algorithmic bytes, never much more)
    assert 0 < r["traffic"] < 1.1 * r["algorithmic_bytes_per_launch"]
    assert r["traffic_detail"]["launches"] >= 2 and r["traffic_detail"]["units"] == 12   # (a round of queries is ONE grid on a small shard)


def test_live_traffic_can_be_switched_off_and_never_costs_the_line():
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--no-live-traffic"] + COMMON)
    _check_contract(d, 1)
    assert "traffic_detail" not in d["roofline"] and d["roofline"]["traffic"] is None   # (no committed record of this size)
    env = dict(os.environ, PATH="/nonexistent")   # no rocprofv3: the line still comes out, and says why traffic is missing
    d = _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON, env=env)
    _check_contract(d, 1)
    assert "rocprofv3" in d["roofline"]["traffic_live_error"]


def test_in_process_group_with_one_rccl_rank():
    d = _run([sys.executable, "bench.py", "--mode", "group"] + COMMON)
    _check_contract(d, 1)
    assert d["config"]["rccl_nranks"] == 1 and d["config"]["transport"] == "rccl" and d["sharded_check"] == "ok"
    assert d["latency_ms"]["host_enqueue_p50"] > 0
    _check_exchange_record(d)


def _check_exchange_record(d):
    """An N > 1 line says how its exchange was amortised: the K timed queries were ONE call = one
    all-gather + merge (`exchanges_in_timed_region`), and a second timed leg ran the same queries one call each."""
    assert d["exchanges_in_timed_region"] == 1 and "1 exchange(s) in the timed region" in d["config"]["workload"]
    pq = d["per_query_exchange"]
    assert pq["queries"] == 24 and pq["exchanges"] == 24 and pq["queries_per_s"] > 0 and pq["last_result_equals_stream_leg"] is True
    assert d["device_bytes_resident"]["total"] > 400_000 * 384 * 4   # the fp32 rows and the u8 shadow at least


def test_gpus_n_as_typed_with_shards_sharing_the_gpu():
    """`python bench.py --gpus 3` needs three devices; `--devices 0,0,0` rehearses the same code (threads, exchange by device
    copies, strong split, weak extra) on one."""
    d = _run([sys.executable, "bench.py", "--gpus", "3", "--devices", "0,0,0"] + COMMON)
    _check_contract(d, 3)
    assert d["config"]["rows_per_gpu"] == 133334 and d["config"]["transport"] == "device_copies" and d["config"]["rccl_nranks"] == 0
    assert d["sharded_check"] == "ok" and len(d["roofline"]["per_gpu"]) == 3
    _check_exchange_record(d)
    w = d["weak_scaling_extra"]
    assert w["scaling"] == "weak" and w["results_span_shards"] >= 2 and w["queries_per_s"] > 0


def test_gpus_n_without_enough_devices_says_so():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "64"] + COMMON, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "device(s) are visible" in p.stderr


def test_one_rank_under_the_launcher_without_torch_in_the_worker():
    env = dict(os.environ, WDBX_BENCH_FORCE_GROUP="1")
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", "29517", "bench.py", "--gpus", "1"] + COMMON, env=env, timeout=600)
    _check_contract(d, 1)
    assert d["rccl"]["rccl_nranks"] == 1 and d["config"]["transport"] == "rccl" and d["sharded_check"] == "ok"
    assert d["config"]["driver"].startswith("one process per GPU")
    _check_exchange_record(d)
