# round 3, run 9: lone blocking queries with the host-side final select + in-kernel threshold: parity suite, then latency A/B
set -o pipefail
mkdir -p gpurun_out/r03/lone
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "not config_c3_full and not config_c4 and not config_c5 and not soak and not test_gpu_bench" > gpurun_out/r03/gputests9.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r03/gputests9.log
if [ $rc -ne 0 ]; then exit $rc; fi
python - > gpurun_out/r03/lone/lone_latency.json <<'PY'
import json, sys, time
import numpy as np
sys.path[:0] = ["wdbx-py_amd", "oracle"]
import wdbx_oracle as O
from wdbx_amd import _native
out = {}
for n, d, k, metric in ((250_000, 384, 10, 0), (1_000_000, 384, 10, 0), (2_500_000, 384, 10, 0), (10_000_000, 384, 10, 0), (1_000_000, 768, 100, 1)):
    ix = _native.NativeIndex(d, metric=metric, capacity_rows=n)
    ix.fill_synthetic(O.SEED_CORPUS, 0, n, True)
    qs = O.normalize_rows_fast(O.synth_rows(O.SEED_QUERY, 0, 64, d))
    res = {}
    for mode in (1, 0, 1, 0):
        ix.set_option("lone_host_select", mode)
        for q in qs[:8]:
            ix.search(q, k)
        lat = []
        for r in range(4):
            for q in qs:
                t0 = time.perf_counter(); ix.search(q, k); lat.append(time.perf_counter() - t0)
        res.setdefault("host_select" if mode else "merge_launches", []).append(round(float(np.percentile(lat, 50)) * 1e6, 1))
    ix.set_option("lone_host_select", 1)
    a = [ix.search(q, k) for q in qs[:16]]
    ix.set_option("lone_host_select", 0)
    b = [ix.search(q, k) for q in qs[:16]]
    res["identical"] = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    out[f"{n}x{d} k{k} {'l2' if metric else 'cos'}"] = res
    ix.close()
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r03/lone/lone_latency.json
