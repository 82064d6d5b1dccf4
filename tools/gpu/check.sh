#!/bin/bash
# One GPU-box pass over everything a change can break: the GPU suite, smoke(), the staged multi-GPU preflight (on one GPU:
# its copy-exchange and single-rank launcher stages), a 2-shard group bench line.  Logs under gpurun_out/$TAG/.
#   gpurun --timeout 1150 -- 'bash tools/gpu/check.sh r04/check1'
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python -m pytest tests -m gpu -x -q > "$OUT/gputests.log" 2>&1
echo "gpu suite rc=$?"; tail -4 "$OUT/gputests.log"
python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
python tools/preflight_multigpu.py --devices 0,0 --out "$OUT/preflight_one_gpu.json" > "$OUT/preflight_one_gpu.log" 2>&1
echo "preflight rc=$?"; tail -7 "$OUT/preflight_one_gpu.log"
python bench.py --gpus 2 --devices 0,0 --steps 40 --warmup 5 --rows 2500000 --no-facade > "$OUT/bench_group2_one_gpu.json" 2> "$OUT/bench_group2_one_gpu.err"
echo "group bench rc=$?"; tail -2 "$OUT/bench_group2_one_gpu.err"
python - "$OUT/bench_group2_one_gpu.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "exchanges_in_timed_region", "per_query_exchange", "sharded_check")})
    print(d["config"]["workload"])
except Exception as e:
    print("no bench line:", e)
PY
