#!/bin/bash
# config 1 (10 k x 384) and a 65 k-row shard: the fp32 scan's grid size (option scan_blocks) against the default
set -o pipefail
O=gpurun_out/r03/c1_blocks
mkdir -p $O
for rows in 10000 65000; do
for b in 0 160 80 40 20; do
  timeout -k 10 200 python bench.py --workload c1 --rows $rows --steps 2000 --warmup 100 --latency-queries 200 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --verify 0 --opt scan_blocks=$b > $O/c1_${rows}_$b.json 2> $O/err.log || exit $?
  python - $O/c1_${rows}_$b.json $rows $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("rows", sys.argv[2], "blocks", sys.argv[3], round(d["value"]), "q/s  p50", round(d["latency_ms"]["p50"] * 1e3, 1), "us  kernel", round(d["roofline"]["avg_launch_ms"] * 1e3, 2), "us  merge", round(d["roofline"].get("merge_avg_ms", 0) * 1e3, 2))
PY
done
done
