#!/bin/bash
# live PMC traffic after the timed legs: bench tests, then the default line twice
set -o pipefail
O=gpurun_out/r03
mkdir -p $O/final
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q > $O/final/bench_tests.log 2>&1; rc=$?
tail -3 $O/final/bench_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in c d; do
timeout -k 10 600 python bench.py > $O/bench_default_$i.json 2> $O/final/bench_default_$i.err; echo "bench default rc=$?"
python - $O/bench_default_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"], 1), "q/s frac", round(r["frac"], 4), "fp32", round(r["frac_fp32_rows_kernel"], 4), "c4", round(d["other_configs"]["c4"]["frac"], 4),
      round(d["other_configs"]["c4"]["queries_per_s"]), "c2", round(d["other_configs"]["c2"]["frac"], 3), "c3", round(d["other_configs"]["c3"]["frac"], 3), "parity", d["parity"]["parity_check"], d["other_configs"]["c4"]["parity"]["parity_check"], "traffic x", r.get("traffic_over_algorithmic"), r.get("traffic_live_error"))
PY
done
