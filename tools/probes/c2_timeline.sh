#!/bin/bash
# timeline of lone device-resident queries on 1 M x 384 (config 2): kernel start / end stamps of the last few queries
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/c2_timeline
mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload c2 --steps 8 --warmup 8 --latency-queries 30 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --verify 0 > $O/c2.json 2> $O/trace.err || exit $?
python3 - $(find $O/trace -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-28:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f'{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:7.1f} us  {r["Kernel_Name"][:60]}')
    prev_end = e
PY
rm -rf $O/trace
