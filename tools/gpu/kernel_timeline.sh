#!/bin/bash
# Kernel timeline of a Python program: start / gap-to-previous / duration of the LAST <n> kernels of the run, from a rocprofv3
# kernel trace.  Shows where a lone query's time goes: kernels vs launch gaps vs the host's share (wall clock minus the span).
#   usage (on the GPU box): tools/gpu/kernel_timeline.sh <outdir> <n> <script.py> [arguments...]
#   e.g.  tools/gpu/kernel_timeline.sh gpurun_out/r04/c2_timeline 24 bench.py --workload c2 --steps 8 --warmup 8 --latency-queries 30 \
#             --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --verify 0
#         tools/gpu/kernel_timeline.sh gpurun_out/r04/lone_blocking_1m 12 tools/probes/lone_blocking.py 1000000
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; N=$2; PROG=$3; shift 3
mkdir -p $O; O=$(cd $O && pwd)
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/$PROG "$@" > $O/program.out 2> $O/trace.err || exit $?
python3 - $(find $O/trace -name '*kernel_trace.csv' | head -1) $N > $O/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]):]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f'{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:7.1f} us  {r["Kernel_Name"][:70]}')
    prev_end = e
PY
cat $O/timeline.txt
tail -3 $O/program.out
rm -rf $O/trace
