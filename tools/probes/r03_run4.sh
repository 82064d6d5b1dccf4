# round 3, run 4: one grid per round for the u8 phase-1 passes (A/B against one launch per query), block-form tile epilogue A/B
set -o pipefail
mkdir -p gpurun_out/r03/scan8_grid gpurun_out/r03/c4_i8
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r03/$name.json 2> gpurun_out/r03/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group or u8 or single_query or config_t or config_c2 or config_c3 or selection or masked" > gpurun_out/r03/gputests4.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r03/gputests4.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
run c4_i8/ab_0_6 python tools/probes/c4_i8_ab.py 10000000 384 0,6 6
grep -v "^   [01]\." gpurun_out/r03/c4_i8/ab_0_6.json; tail -3 gpurun_out/r03/c4_i8/ab_0_6.err
for rows in 10000000 1250000; do
  run scan8_grid/one_grid_$rows python bench.py --rows $rows --steps 400 --warmup 40 --no-cpu-baseline --no-other-configs --no-facade
  run scan8_grid/per_query_$rows python bench.py --rows $rows --steps 400 --warmup 40 --no-cpu-baseline --no-other-configs --no-facade --opt scan8_per_query=1
done
run scan8_grid/one_grid_c2 python bench.py --workload c2 --steps 400 --warmup 40 --no-cpu-baseline --no-facade
run scan8_grid/one_grid_c3 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-facade
grep -h -o '"value": [0-9.]*\|"frac": [0-9.]*, "traffic\|"avg_launch_ms": [0-9.]*' gpurun_out/r03/scan8_grid/*.json
