# round 3 records, part 4 (final code after the last kernel change): whole GPU suite, default bench line, the i8 tile
# kernel under rocprofv3 (kernel trace, FETCH/WRITE, SQ counters), clock table of the default form against the round-2 form
set -o pipefail
O=gpurun_out/r03
mkdir -p $O/c4_i8
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gputests_full.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/gputests_full.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench default rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 900 bash tools/probes/c4_i8_profile.sh $PWD/$O/c4_i8 > $O/c4_i8_profile.log 2>&1; rc=$?; echo "c4 profile rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
head -3 $O/c4_i8/c4_i8_kernel_stats.csv | cut -c1-160
timeout -k 10 300 bash tools/probes/clock_by_variant.sh $PWD/$O/c4_i8/clock 0,13 > $O/c4_i8/clock_by_variant.log 2>&1; rc=$?; echo "clock rc=$rc"; cat $O/c4_i8/clock/summary.txt
timeout -k 10 200 python tools/bench_batch.py 10000000 384 256 10 20 > $O/c4_i8/bench_batch_256.json 2> $O/c4_i8/bench_batch_256.err; echo "batch rc=$?"
timeout -k 10 200 python tools/bench_batch.py 10000000 768 256 10 10 > $O/batch_sizes/bench_batch_768_cos_256.json 2> $O/batch_sizes/b768.err; echo "768 rc=$?"
grep -o '"ms_per_batch": [0-9.]*' $O/batch_sizes/bench_batch_768_cos_256.json
