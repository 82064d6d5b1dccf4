# round 3 records, part 2: the i8 tile kernel (config 4) under rocprofv3 (kernel trace, FETCH/WRITE, SQ counters, clock by
# variant), whole-batch timings, then the whole GPU suite as one record
set -o pipefail
O=gpurun_out/r03
mkdir -p $O/c4_i8
timeout -k 10 900 bash tools/probes/c4_i8_profile.sh $PWD/$O/c4_i8 > $O/c4_i8_profile.log 2>&1; rc=$?; echo "c4 profile rc=$rc"; tail -5 $O/c4_i8_profile.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 bash tools/probes/clock_by_variant.sh $PWD/$O/c4_i8/clock 0,6 > $O/c4_i8/clock_by_variant.log 2>&1; rc=$?; echo "clock rc=$rc"; cat $O/c4_i8/clock/summary.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python tools/bench_batch.py 10000000 384 256 10 20 > $O/c4_i8/bench_batch_256.json 2> $O/c4_i8/bench_batch_256.err; echo "batch rc=$?"
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gputests_full.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/gputests_full.log
