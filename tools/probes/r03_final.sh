# final check of the committed code: whole GPU suite, smoke, default bench line
set -o pipefail
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gputests_full.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/gputests_full.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench default rc=$rc"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"], 1), "q/s frac", round(d["roofline"]["frac"], 4), "fp32", round(d["roofline"]["frac_fp32_rows_kernel"], 4), "c4", round(d["other_configs"]["c4"]["frac"], 4),
      round(d["other_configs"]["c4"]["queries_per_s"]), "c2", round(d["other_configs"]["c2"]["frac"], 3), "parity", d["parity"]["parity_check"], d["other_configs"]["c4"]["parity"]["parity_check"])
PY
