# the shard sizes of a 10 M-row corpus split over 1, 2, 4 and 8 GPUs, on ONE GPU: the plain index, the in-process shard group
# (S = 1, RCCL 1-rank communicator) and one rank under the launcher (no torch in the worker) -- what each GPU of an N-GPU
# strong-scaling run has to do per query
set -o pipefail
O=gpurun_out/r03/group_rehearsal; mkdir -p $O
for rows in 10000000 5000000 2500000 1250000; do
  timeout -k 10 200 python bench.py --rows $rows --steps 400 --warmup 40 --no-other-configs --no-cpu-baseline --no-facade --no-live-traffic > $O/index_$rows.json 2> $O/index_$rows.err || exit 1
  timeout -k 10 200 python bench.py --mode group --rows $rows --steps 400 --warmup 40 --no-facade > $O/group_$rows.json 2> $O/group_$rows.err || exit 1
  WDBX_BENCH_FORCE_GROUP=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 1 --rows $rows --steps 400 --warmup 40 > $O/launcher_$rows.json 2> $O/launcher_$rows.err || exit 1
done
python3 - <<'PY'
import json, glob
for kind in ("index", "group", "launcher"):
    for rows in (10000000, 5000000, 2500000, 1250000):
        d = json.loads(open(f"gpurun_out/r03/group_rehearsal/{kind}_{rows}.json").read().strip().splitlines()[-1])
        print(kind, rows, round(d["value"], 1), "q/s", round(d["ms_per_step"], 4), "ms/step  kernel", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 3), d.get("sharded_check"), d["config"]["transport"])
PY
