#!/bin/bash
# durations of the sample launches of full rounds on the headline workload: plain form vs several queries per workgroup
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/sample4_trace
mkdir -p $O
cd /tmp
for m in 0 2; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -o t -- python3 $R/bench.py --steps 128 --warmup 32 --latency-queries 0 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --verify 0 --opt scan8_sample4=$m > $O/t_$m.json 2> $O/trace.err || exit $?
  python3 - $(find $O/trace$m -name '*kernel_trace.csv' | head -1) $m <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "scan8" in n:
        d[n.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print("mode", sys.argv[2], k[:60], "calls", len(v), "median us", round(v[len(v) // 2], 1), "max", round(v[-1], 1))
PY
  rm -rf $O/trace$m
done
