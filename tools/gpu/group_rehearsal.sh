#!/bin/bash
# The shard sizes of a 10 M-row corpus split over 1, 2, 4 and 8 GPUs, on ONE GPU: the plain index, the in-process shard group
# (S = 1, RCCL 1-rank communicator) and one rank under the launcher's environment (no torch in the worker) -- what each GPU of
# an N-GPU strong-scaling run has to do per query, incl. the per-query-exchange leg of the N > 1 lines.
#   usage (on the GPU box): tools/gpu/group_rehearsal.sh <outdir> [steps]
set -o pipefail
O=${1:-gpurun_out/group_rehearsal}; STEPS=${2:-400}; mkdir -p $O
for rows in 10000000 5000000 2500000 1250000; do
  timeout -k 10 200 python bench.py --rows $rows --steps $STEPS --warmup 40 --no-other-configs --no-cpu-baseline --no-facade --no-live-traffic > $O/index_$rows.json 2> $O/index_$rows.err || exit 1
  timeout -k 10 200 python bench.py --mode group --rows $rows --steps $STEPS --warmup 40 --no-facade > $O/group_$rows.json 2> $O/group_$rows.err || exit 1
  WDBX_BENCH_FORCE_GROUP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 \
      python bench.py --gpus 1 --rows $rows --steps $STEPS --warmup 40 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic > $O/launcher_$rows.json 2> $O/launcher_$rows.err || exit 1
done
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for kind in ("index", "group", "launcher"):
    for rows in (10000000, 5000000, 2500000, 1250000):
        d = json.loads(open(f"{O}/{kind}_{rows}.json").read().strip().splitlines()[-1])
        pq = d.get("per_query_exchange") or {}
        print(kind, rows, round(d["value"], 1), "q/s", round(d["ms_per_step"], 4), "ms/step  kernel", round(d["roofline"]["avg_launch_ms"], 4), "frac",
              round(d["roofline"]["frac"], 3), "exchanges", d.get("exchanges_in_timed_region"), "per-query-exchange q/s", pq.get("queries_per_s") and round(pq["queries_per_s"], 1),
              d.get("sharded_check"), d["config"]["transport"])
PY
