set -e
O=gpurun_out/r01b; mkdir -p $O
for nq in 16 64 128 256 1024; do timeout -k 10 300 python3 tools/bench_batch.py 10000000 384 $nq 10 5 > $O/bench_batch_bf16_nq$nq.json 2>> $O/err.txt; done
timeout -k 10 300 python3 tools/bench_batch.py 10000000 768 256 100 5 1 > $O/bench_batch_bf16_l2_768_k100.json 2>> $O/err.txt
WDBX_OPTS=gemm_bf16=1 timeout -k 10 300 python3 tools/bench_batch.py 10000000 384 256 10 5 > $O/bench_batch_bf16_noshadow.json 2>> $O/err.txt
WDBX_OPTS=gemm_bf16=0 timeout -k 10 300 python3 tools/bench_batch.py 10000000 384 256 10 5 > $O/bench_batch_fp32_tiles.json 2>> $O/err.txt
timeout -k 10 400 python3 tools/bench_async.py > $O/bench_async_facade.json 2>> $O/err.txt
timeout -k 10 300 python3 bench.py --workload c4 --steps 20 --warmup 3 --no-other-configs > $O/bench_c4.json 2>> $O/err.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r01b/bench_batch*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], round(d["ms_per_batch"], 3), round(d["queries_per_s"]), round(d["gemm_ms_per_batch"], 3), d["gemm_family"], d["agreement_with_fp32_tiles"])
PY
tail -c 1500 $O/bench_async_facade.json
