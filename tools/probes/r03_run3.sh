# round 3, run 3: group tests (parity + facade), i8 tile kernel A/B (product / thresholds ahead of time / + 6-deep ring), group host cost again
set -o pipefail
mkdir -p gpurun_out/r03/group_host gpurun_out/r03/c4_i8
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r03/$name.json 2> gpurun_out/r03/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -m gpu -x -q -k "group or exception_barrier" > gpurun_out/r03/gputests3.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r03/gputests3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
run c4_i8/ab_0_6_7 python tools/probes/c4_i8_ab.py 10000000 384 0,6,7 6
cat gpurun_out/r03/c4_i8/ab_0_6_7.json | grep -v all; tail -3 gpurun_out/r03/c4_i8/ab_0_6_7.err
run group_host/group_1250000_b python bench.py --mode group --rows 1250000 --steps 400 --warmup 40 --no-facade
run group_host/group_8shards_one_gpu_10m_b python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 10
grep -h -o '"value": [0-9.]*\|"host_enqueue_p50": [0-9.]*\|"p50": [0-9.]*' gpurun_out/r03/group_host/*_b.json
