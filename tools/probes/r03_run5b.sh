set -o pipefail
mkdir -p gpurun_out/r03/batch_sizes
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fuzz.py -k "not config_t and not config_c3_full and not config_c4 and not soak and not test_cosine_search and not every_kernel_variant" > gpurun_out/r03/gputests5b.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r03/gputests5b.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/bench_batch.py 10000000 768 256 100 10 1 > gpurun_out/r03/batch_sizes/bench_batch_c3_l2_256_i8.json 2> gpurun_out/r03/batch_sizes/c3_i8.err; echo "c3 i8 rc=$?"
WDBX_OPTS=gemm_l2_i8=0 timeout -k 10 200 python tools/bench_batch.py 10000000 768 256 100 10 1 > gpurun_out/r03/batch_sizes/bench_batch_c3_l2_256_bf16.json 2> gpurun_out/r03/batch_sizes/c3_bf16.err; echo "c3 bf16 rc=$?"
grep -h "ms_per_batch\|queries_per_s\|ids_equal\|family" gpurun_out/r03/batch_sizes/bench_batch_c3_l2_256_*.json
