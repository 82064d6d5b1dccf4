#!/bin/bash
# kernel trace of config 3 (10 M x 768, L2, top-100, single queries)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/c3_trace
mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload c3 --steps 64 --warmup 8 --latency-queries 0 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --verify 0 > $O/c3_under_rocprof.json 2> $O/trace.err || exit $?
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/c3_kernel_stats.csv
python3 - $O/c3_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), r["AverageNs"].rjust(14), r["MinNs"].rjust(10), r["Percentage"])
PY
rm -rf $O/trace
