#!/bin/bash
# final check of the committed code (second half of round 3): whole GPU suite, smoke, default bench line (live PMC traffic),
# c4 with / without the second selection stage, and rocprofv3 records (kernel trace + FETCH/WRITE passes) of the three
# bench command lines whose kernels changed
set -o pipefail
O=gpurun_out/r03
mkdir -p $O/c4_i8 $O/final
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gputests_full.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/gputests_full.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; src=$?; echo "smoke rc=$src"; tail -1 $O/smoke.log
if [ $src -eq 124 ] || [ $src -eq 137 ]; then exit $src; fi
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; brc=$?; echo "bench default rc=$brc"
if [ $brc -eq 124 ] || [ $brc -eq 137 ]; then exit $brc; fi
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"], 1), "q/s frac", round(r["frac"], 4), "fp32", round(r["frac_fp32_rows_kernel"], 4), "c4", round(d["other_configs"]["c4"]["frac"], 4),
      round(d["other_configs"]["c4"]["queries_per_s"]), "c2", round(d["other_configs"]["c2"]["frac"], 3), "parity", d["parity"]["parity_check"], d["other_configs"]["c4"]["parity"]["parity_check"])
print("traffic", r.get("traffic"), r.get("traffic_over_algorithmic"), r.get("traffic_live_error"), (r.get("traffic_source") or "")[:50])
PY
for i in 1 2; do
  for r in 0 1; do
    timeout -k 10 300 python bench.py --workload c4 --steps 60 --warmup 10 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --opt gemm8_refine=$r > $O/c4_i8/refine_ab_${r}_$i.json 2> $O/final/ab.err || exit $?
  done
done
timeout -k 10 500 python bench.py --workload c4 --steps 60 --warmup 10 --no-cpu-baseline --no-facade --no-other-configs > $O/c4_i8/c4_live.json 2> $O/final/c4_live.err || exit $?
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03/c4_i8/refine_ab_*.json")) + ["gpurun_out/r03/c4_i8/c4_live.json"]:
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"], 4), "pair", round(r.get("gemm_ms_per_step", 0), 4), "frac", round(r["frac"], 4), "cand", r.get("candidates_per_query"), "traffic", r.get("traffic_over_algorithmic"))
PY
bash tools/probes/bench_profile.sh $O/c4_final --workload c4 --steps 20 --warmup 5 > $O/final/c4_profile.log 2>&1 || { tail -5 $O/final/c4_profile.log; exit 1; }
bash tools/probes/bench_profile.sh $O/t_u8_final --steps 100 --warmup 20 > $O/final/t_u8_profile.log 2>&1 || { tail -5 $O/final/t_u8_profile.log; exit 1; }
bash tools/probes/bench_profile.sh $O/fp32_scan_final --steps 60 --warmup 10 --opt scan_shadow=0 > $O/final/fp32_profile.log 2>&1 || { tail -5 $O/final/fp32_profile.log; exit 1; }
grep -h "gemm_i8_kernel<1\|scan8_kernel<8, 3, 0, 1\|scan_kernel<8, 12" $O/c4_final/kernel_stats.csv $O/t_u8_final/kernel_stats.csv $O/fp32_scan_final/kernel_stats.csv | cut -c1-120
