set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c4prof
mkdir -p $O
cd /tmp
CMD="python3 $R/bench.py --workload c4 --steps 20 --warmup 3 --no-other-configs --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- $CMD > $O/stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o c4 -- $CMD > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o c4 -- $CMD > $O/write.json 2> $O/write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -o c4 -- $CMD > $O/sq.json 2> $O/sq.err
cd $R
python3 - <<'PY'
import csv, glob, collections, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/c4prof"
out = {}
for grp in ("fetch", "write", "sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0][:60]
            agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(kn, {})[c] = {"n": len(v), "mean": sum(v) / len(v), "max": max(v)}
json.dump(out, open(f"{O}/pmc_summary.json", "w"), indent=1)
for kn, cs in out.items():
    if "gemm" in kn or "rescore" in kn:
        print(kn, {c: (round(v["mean"], 1), round(v["max"], 1), v["n"]) for c, v in cs.items()})
PY
head -12 $O/stats/*kernel_stats.csv
