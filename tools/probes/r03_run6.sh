# round 3, run 6: the config-5-shaped parity test (80 M rows in 8 shards on one GPU), rounds of 64 vs 32 queries
set -o pipefail
mkdir -p gpurun_out/r03/round64
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r03/$name.json 2> gpurun_out/r03/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config_c5 or group" > gpurun_out/r03/gputests6.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r03/gputests6.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for rows in 1250000 10000000; do
  run round64/r32_$rows python bench.py --rows $rows --steps 512 --warmup 64 --no-cpu-baseline --no-other-configs --no-facade
  run round64/r64_$rows python bench.py --rows $rows --steps 512 --warmup 64 --no-cpu-baseline --no-other-configs --no-facade --opt exchange_batch=64
done
grep -h -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/r03/round64/*.json
