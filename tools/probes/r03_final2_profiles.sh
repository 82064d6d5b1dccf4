#!/bin/bash
# rocprofv3 records (kernel trace + FETCH/WRITE passes) of the three bench command lines, final code
set -o pipefail
O=gpurun_out/r03
mkdir -p $O/final
bash tools/probes/bench_profile.sh $O/c4_final --workload c4 --steps 20 --warmup 5 > $O/final/c4_profile.log 2>&1 || { tail -5 $O/final/c4_profile.log; exit 1; }
bash tools/probes/bench_profile.sh $O/t_u8_final --steps 100 --warmup 20 > $O/final/t_u8_profile.log 2>&1 || { tail -5 $O/final/t_u8_profile.log; exit 1; }
bash tools/probes/bench_profile.sh $O/fp32_scan_final --steps 60 --warmup 10 --opt scan_shadow=0 > $O/final/fp32_profile.log 2>&1 || { tail -5 $O/final/fp32_profile.log; exit 1; }
tail -14 $O/final/c4_profile.log; tail -10 $O/final/t_u8_profile.log; tail -10 $O/final/fp32_profile.log
# (a second default line of the same code: the boxes differ by ~2 %)
timeout -k 10 600 python bench.py > $O/bench_default_b.json 2> $O/final/bench_default_b.err; echo "bench default rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/bench_default_b.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"], 1), "q/s frac", round(r["frac"], 4), "fp32", round(r["frac_fp32_rows_kernel"], 4), "c4", round(d["other_configs"]["c4"]["frac"], 4),
      round(d["other_configs"]["c4"]["queries_per_s"]), "c2", round(d["other_configs"]["c2"]["frac"], 3), "c3", round(d["other_configs"]["c3"]["frac"], 3), "parity", d["parity"]["parity_check"], d["other_configs"]["c4"]["parity"]["parity_check"], "traffic x", r.get("traffic_over_algorithmic"))
PY
