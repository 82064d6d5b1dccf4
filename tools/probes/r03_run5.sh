# round 3, run 5: the whole GPU suite on the current build (L2 on the i8 tiles, compaction, group, one-grid scan8), then L2 batch timings
set -o pipefail
mkdir -p gpurun_out/r03/batch_sizes
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests5.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r03/gputests5.log
if [ $rc -ne 0 ]; then exit $rc; fi
# config 3 in batches of 256: L2 on the i8 tiles (default) and on the bf16 tiles (gemm_l2_i8=0)
timeout -k 10 200 python tools/bench_batch.py 10000000 768 256 100 10 1 > gpurun_out/r03/batch_sizes/bench_batch_c3_l2_256_i8.json 2> gpurun_out/r03/batch_sizes/c3_i8.err; echo "c3 i8 rc=$?"
WDBX_OPTS=gemm_l2_i8=0 timeout -k 10 200 python tools/bench_batch.py 10000000 768 256 100 10 1 > gpurun_out/r03/batch_sizes/bench_batch_c3_l2_256_bf16.json 2> gpurun_out/r03/batch_sizes/c3_bf16.err; echo "c3 bf16 rc=$?"
grep -h "ms_per_batch\|queries_per_s\|ids_equal\|family" gpurun_out/r03/batch_sizes/bench_batch_c3_l2_256_*.json
