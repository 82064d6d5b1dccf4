set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -m gpu -x -q > gpurun_out/r03/gputests8.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r03/gputests8.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r03/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03/smoke.log
