# the N > 1 code path (uid broadcast, RCCL communicator, sharded search + all-gather merge, self-check) with one
# rank, at the shard sizes of a 10M-row corpus split over 1, 2, 4 and 8 GPUs
set -e
O=gpurun_out/group; mkdir -p $O
for rows in 10000000 5000000 2500000 1250000; do
  WDBX_BENCH_FORCE_GROUP=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 1 --rows $rows --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > $O/g1_$rows.json 2> $O/g1_$rows.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/group/g1_*.json"), key=lambda x: -int(x.split("_")[-1].split(".")[0])):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(d["config"]["rows_per_gpu"], round(d["value"], 1), "q/s", round(d["ms_per_step"], 4), "ms", d["roofline"]["kernel"][:24], round(d["roofline"]["avg_launch_ms"], 4), d["sharded_check"], d["config"]["transport"])
PY
