# round-1 measurement pass (one box): writes gpurun_out/final/
O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 python3 bench.py --opt scan_shadow=0 --no-other-configs --no-cpu-baseline > $O/bench_default_fp32_scan.json 2>> $O/err.txt
for w in c1 c2 c3 c4; do timeout -k 10 300 python3 bench.py --workload $w --no-other-configs --no-cpu-baseline > $O/bench_$w.json 2>> $O/err.txt; done
timeout -k 10 300 python3 bench.py --workload c4 --opt gemm_bf16=0 --steps 10 --warmup 2 --no-other-configs --no-cpu-baseline > $O/bench_c4_fp32_tiles.json 2>> $O/err.txt
timeout -k 10 600 python3 tools/bench_dims_shadow.py > $O/bench_dims_shadow.txt 2>> $O/err.txt
timeout -k 10 300 python3 tools/bench_facade.py > $O/facade_latency.json 2>> $O/err.txt
timeout -k 10 400 python3 tools/bench_async.py > $O/bench_async_facade.json 2>> $O/err.txt
tail -c 600 $O/bench_async_facade.json; echo
tail -n 3 $O/bench_dims_shadow.txt | cut -c1-600
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms/step", d["roofline"]["kernel"][:28], round(d["roofline"]["frac"], 3))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
