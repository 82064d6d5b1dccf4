#!/bin/bash
# effective clock (GRBM_GUI_ACTIVE / 8 / duration) and matrix-pipe busy share per gemm_i8_kernel variant: rocprofv3 PMC pass over
# tools/probes/c4_i8_ab.py.  usage: clock_by_variant.sh <outdir> <variants>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; V=$2
mkdir -p $O
O=$(cd $O && pwd)   # (absolute: the pass runs from /tmp)
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/raw -o p -- python3 $R/tools/probes/c4_i8_ab.py 10000000 384 $V 1 > $O/run.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
dur = {}
for f in glob.glob(O + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(dict)
for f in glob.glob(O + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for d, (name, ns) in dur.items():
    if "gemm_i8_kernel<1" in name and d in cnt:
        c = cnt[d]
        ghz = c.get("GRBM_GUI_ACTIVE", 0) / 8 / ns
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024)
        agg[name.split("(")[0]].append((ns / 1e6, ghz, busy, c.get("SQ_INSTS_MFMA", 0)))
with open(O + "/summary.txt", "w") as fh:
    for name, v in sorted(agg.items()):
        n = len(v)
        line = "%-40s n=%d  ms=%.3f  clock_GHz=%.3f  mfma_busy=%.3f  mfma_insts=%.3g" % (name, n, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, sum(x[2] for x in v) / n, sum(x[3] for x in v) / n)
        print(line); fh.write(line + "\n")
PY
rm -rf $O/raw
