#!/bin/bash
# config 4: the sample stride of the batched i8 path (option gemm_sample_div; default 32 at k = 10), now that the second
# selection stage makes extra candidates cheap
set -o pipefail
O=gpurun_out/r03/c4_sample_div_b
mkdir -p $O
for rep in 1 2; do
for dv in 0 24 16 12; do
  timeout -k 10 300 python bench.py --workload c4 --steps 60 --warmup 10 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --opt gemm_sample_div=$dv > $O/c4_div${dv}_$rep.json 2> $O/err.log || exit $?
  python - $O/c4_div${dv}_$rep.json $dv <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("div", sys.argv[2], round(d["value"]), "q/s  ms/batch", round(d["ms_per_step"], 4), " pair", round(r["gemm_ms_per_step"], 4), "frac", round(r["frac"], 4), "cand", round(r["candidates_per_query"], 1), d["parity"]["parity_check"])
PY
done
done
