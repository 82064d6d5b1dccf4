#!/bin/bash
# rocprofv3 records of BASELINE config 4 on the i8 tiles: kernel trace stats, FETCH_SIZE / WRITE_SIZE (separate passes) and the
# SQ counter groups of pmc_probe.sh for gemm_i8_kernel<1, 4>.  usage (on the GPU box): tools/probes/c4_i8_profile.sh [outdir] [opts]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/r02/c4_i8}
export WDBX_OPTS=${2:-}
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python3 $R/tools/bench_batch.py 10000000 384 256 10 20 > $O/bench_under_rocprof.json 2> $O/stats.err
cp $O/stats/*/*kernel_stats.csv $O/c4_i8_kernel_stats.csv 2>/dev/null || cp $O/stats/*kernel_stats.csv $O/c4_i8_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o c4 -- python3 $R/tools/bench_batch.py 10000000 384 256 10 10 > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o c4 -- python3 $R/tools/bench_batch.py 10000000 384 256 10 10 > $O/write.json 2> $O/write.err
bash $R/tools/probes/pmc_probe.sh $O/sq "gemm_i8_kernel<1" python3 $R/tools/bench_batch.py 10000000 384 256 10 10 > $O/sq_summary.txt 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
out = {}
for grp in ("fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(kn, {})[c] = {"n": len(v), "mean": sum(v) / len(v), "max": max(v)}
json.dump(out, open(f"{O}/c4_i8_pmc_summary.json", "w"), indent=1)
for kn, cs in out.items():
    if "gemm_i8" in kn or "rescore" in kn:
        print(kn, {c: (round(v["mean"], 1), v["n"]) for c, v in cs.items()})
PY
head -14 $O/c4_i8_kernel_stats.csv
cat $O/sq_summary.txt | tail -32
rm -rf $O/stats $O/fetch $O/write $O/sq/g*
