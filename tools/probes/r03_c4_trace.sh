#!/bin/bash
# kernel trace of the c4 batch on the current code: what is left around the tile pass pair
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/refine
mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload c4 --steps 40 --warmup 10 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic > $O/c4_under_rocprof.json 2> $O/trace.err || exit $?
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/c4_kernel_stats.csv
python3 - $O/c4_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), r["AverageNs"].rjust(14), r["Percentage"])
PY
rm -rf $O/trace
