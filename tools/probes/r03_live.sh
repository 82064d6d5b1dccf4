#!/bin/bash
# bench.py with live PMC traffic: its tests, then the c4 line
set -o pipefail
O=gpurun_out/r03/live
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -m gpu -x -q -k "bench or plain or live or gpus or launcher or rank or single or u8 or t_10m or c2 or c3 or lone" > $O/tests.log 2>&1
rc=$?
tail -5 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --workload c4 --steps 60 --warmup 10 --no-cpu-baseline --no-facade --no-other-configs > $O/c4_live.json 2> $O/c4_live.err || exit $?
python - $O/c4_live.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("c4", round(d["value"]), "q/s", round(d["ms_per_step"], 4), "frac", round(r["frac"], 4), "traffic", r.get("traffic"), r.get("traffic_over_algorithmic"), r.get("traffic_live_error"), r.get("traffic_source", "")[:60])
PY
