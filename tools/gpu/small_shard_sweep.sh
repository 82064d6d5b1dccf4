#!/bin/bash
# Workgroups per CU of the u8 selection scan's full pass (option scan8_wgs) by shard size, alternating, REPS runs each: on
# small shards a pass is short, and more (shorter) workgroups per pass let the tail of one pass overlap the start of the next.
#   usage (on the GPU box): tools/gpu/small_shard_sweep.sh <outdir> "<rows...>" "<wgs...>" [reps]
O=${1:-gpurun_out/small_shard}; ROWS=${2:-"1250000 2500000 10000000"}; WGS=${3:-"2 4 6"}; REPS=${4:-3}
mkdir -p $O
for rows in $ROWS; do
 for rep in $(seq 1 $REPS); do
  for w in $WGS; do
   python bench.py --rows $rows --steps 640 --warmup 64 --no-other-configs --no-cpu-baseline --no-facade --no-live-traffic --verify 0 --latency-queries 0 --opt scan8_wgs=$w > $O/wgs_${w}_${rows}_$rep.json 2>/dev/null
  done
 done
done
python3 - $O "$ROWS" "$WGS" $REPS <<'PY' | tee $O/summary.txt
import json, sys, statistics
O, rows_l, wgs_l, reps = sys.argv[1], sys.argv[2].split(), sys.argv[3].split(), int(sys.argv[4])
for rows in rows_l:
    for w in wgs_l:
        v, km = [], []
        for rep in range(1, reps + 1):
            d = json.loads(open(f"{O}/wgs_{w}_{rows}_{rep}.json").read().strip().splitlines()[-1])
            v.append(d["value"]); km.append(d["roofline"]["avg_launch_ms"])
        print(f"rows {rows:>9} wgs {w}: q/s {[round(x, 1) for x in v]} median {statistics.median(v):.1f}; kernel ms median {statistics.median(km):.5f}")
PY
