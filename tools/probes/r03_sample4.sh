#!/bin/bash
# several queries per workgroup in the sample pass: its test, then the headline and config 3 with / without it, alternating
set -o pipefail
O=gpurun_out/r03/sample4
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sample_pass_for_several or t_10m or c3_full" > $O/tests.log 2>&1; rc=$?
tail -12 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2 3; do
for r in 0 1; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --latency-queries 20 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --opt scan8_sample4=$r > $O/t_s4${r}_$rep.json 2> $O/err.log || exit $?
  timeout -k 10 300 python bench.py --workload c3 --steps 96 --warmup 8 --latency-queries 20 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic --opt scan8_sample4=$r > $O/c3_s4${r}_$rep.json 2> $O/err.log || exit $?
  python - $O/t_s4${r}_$rep.json $O/c3_s4${r}_$rep.json $r <<'PY'
import json, sys
for f in sys.argv[1:3]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("sample4", sys.argv[3], f.split("/")[-1], round(d["value"], 1), "q/s  ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "kernel", round(d["roofline"]["avg_launch_ms"], 4), d["parity"]["parity_check"])
PY
done
done
