#!/bin/bash
# rocprofv3 record of one bench.py command line: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in passes of their own
# (MI355X_MICROARCH.md: counters in their own runs, gfx950 FETCH_SIZE counts half of wide streaming reads).
# usage (on the GPU box): tools/gpu/profile_bench.sh <outdir> [bench.py arguments...]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; shift
mkdir -p $O
O=$(cd $O && pwd)   # (absolute: the passes run from /tmp)
cd /tmp
ARGS="--no-other-configs --no-cpu-baseline --no-facade --no-live-traffic $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- python3 $R/bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/stats.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- python3 $R/bench.py $ARGS > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- python3 $R/bench.py $ARGS > $O/write.json 2> $O/write.err
cd $R
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
out = {}
for grp in ("fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(kn, {})[c] = {"n": len(v), "mean_KiB": sum(v) / len(v), "max_KiB": max(v)}
json.dump(out, open(f"{O}/pmc_summary.json", "w"), indent=1)
for kn, cs in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {"mean_KiB": 0})["mean_KiB"])[:6]:
    print(kn, {c: (round(v["mean_KiB"], 1), v["n"]) for c, v in cs.items()})
PY
head -8 $O/kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write
