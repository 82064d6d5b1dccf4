set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests1.log 2>&1; rc=$?
echo "pytest rc=$rc" >> gpurun_out/r03/gputests1.log
tail -5 gpurun_out/r03/gputests1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_default1.json 2> gpurun_out/r03/bench_default1.err; rc=$?
echo "bench default rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --mode group --steps 200 --warmup 20 --no-facade > gpurun_out/r03/bench_group1.json 2> gpurun_out/r03/bench_group1.err; rc=$?
echo "bench group rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --gpus 4 --devices 0,0,0,0 --steps 100 --warmup 10 --no-facade > gpurun_out/r03/bench_g4copy.json 2> gpurun_out/r03/bench_g4copy.err; rc=$?
echo "bench g4 rc=$rc"
tail -c 600 gpurun_out/r03/bench_group1.err gpurun_out/r03/bench_g4copy.err
