#!/bin/bash
# headline workload: the sample stride of the single-query selection scan (option gemm_sample_div; default 32 at k = 10)
set -o pipefail
O=gpurun_out/r03/sample_div
mkdir -p $O
for rep in 1 2; do
for dv in 0 48 64 96 128; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --latency-queries 50 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --opt gemm_sample_div=$dv > $O/t_div${dv}_$rep.json 2> $O/err.log || exit $?
  python - $O/t_div${dv}_$rep.json $dv <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("div", sys.argv[2], round(d["value"], 1), "q/s  ms/step", round(d["ms_per_step"], 4), " p50", round(d["latency_ms"]["p50"], 4), "parity", d["parity"]["parity_check"])
PY
done
done
