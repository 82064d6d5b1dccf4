#!/bin/bash
# wave-level threshold search in the lone query's full pass: tests, timeline, blocking latency by size
set -o pipefail
O=gpurun_out/r03/lone2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -m gpu -x -q -k "lone or single or c2 or c1 or u8 or small or facade_on or latency or zero_copy or host_select or defer" > $O/tests.log 2>&1; rc=$?
tail -4 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/probes/c2_timeline.sh | tail -12
timeout -k 10 200 python tools/probes/lone_query_latency.py 2>&1 | tee $O/lone_latency.txt
timeout -k 10 300 python bench.py --workload c2 --steps 200 --warmup 20 --latency-queries 100 --no-cpu-baseline --no-facade --no-other-configs --no-live-traffic > $O/c2.json 2> $O/err.log || exit $?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/lone2/c2.json").read().strip().splitlines()[-1])
print("c2", round(d["value"]), "q/s p50", round(d["latency_ms"]["p50"] * 1e3, 1), "us", d["parity"]["parity_check"])
PY
