# rocprofv3 record of the default bench line (workload t): kernel stats, then FETCH_SIZE / WRITE_SIZE passes
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tprof
mkdir -p $O
cd /tmp
CMD="python3 $R/bench.py --no-other-configs --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- $CMD > $O/stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- $CMD > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- $CMD > $O/write.json 2> $O/write.err
cd $R
python3 - <<'PY'
import csv, glob, collections, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/tprof"
out = {}
for grp in ("fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(kn, {})[c] = {"n": len(v), "mean": sum(v) / len(v), "max": max(v)}
json.dump(out, open(f"{O}/pmc_summary.json", "w"), indent=1)
for kn, cs in out.items():
    print(kn, {c: (round(v["mean"], 1), v["n"]) for c, v in cs.items()})
PY
head -14 $O/stats/*kernel_stats.csv
cat $O/stats.json
