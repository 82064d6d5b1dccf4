#!/bin/bash
# second selection stage (refine_pairs_kernel): tests, then c4 with and without it, alternating
set -o pipefail
O=gpurun_out/r03/refine
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_stage" > $O/tests.log 2>&1
rc=$?
tail -5 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
  for r in 0 1; do
    timeout -k 10 300 python bench.py --workload c4 --steps 60 --warmup 10 --no-cpu-baseline --no-facade --no-other-configs --opt gemm8_refine=$r > $O/c4_refine${r}_$i.json 2> $O/c4_refine${r}_$i.err || exit $?
    python - $O/c4_refine${r}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), "q/s", round(d["ms_per_step"], 4), "ms/batch", "pair", round(d["roofline"].get("gemm_ms_per_step", 0), 4), "cand", d["roofline"].get("candidates_per_query"), d.get("parity", {}).get("parity_check"))
PY
  done
done
