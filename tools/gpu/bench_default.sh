#!/bin/bash
# The driver's default command line with its wall clock: `python bench.py` -> <out>/bench_default.json (+ .err, wall_seconds.txt)
# and a short digest on stdout.     usage (on the GPU box): tools/gpu/bench_default.sh <outdir> [bench.py arguments...]
O=$1; shift
mkdir -p $O
SECONDS=0
python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err
rc=$?
echo "rc=$rc wall_seconds=$SECONDS" | tee $O/wall_seconds.txt
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d["roofline"], d["cpu_baseline"]
print("value", round(d["value"], 1), d["unit"], "frac", round(r["frac"], 4), "fp32 kernel", round(r["frac_fp32_rows_kernel"], 4),
      "traffic/alg", round(r.get("traffic_over_algorithmic") or 0, 4), "cpu", round(c["value"], 2), "on", c["cores"], "parity", d["parity"]["parity_check"])
for k, v in d.get("other_configs", {}).items():
    print(" ", k, round(v.get("queries_per_s", 0), 1), "q/s frac", v.get("frac"), "p50_ms", v.get("p50_ms"))
print("  latency", d["latency_ms"], "facade", d["facade_latency_ms"]["WDBX.vector_search"])
PY
