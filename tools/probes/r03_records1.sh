# round 3 records, part 1: the default bench line, rocprofv3 kernel trace + FETCH/WRITE PMC of the default (u8) line and of
# the fp32 scan kernel (SURVEY 8d's literal N*d*4 bytes), the c4 (i8 tiles) profile
set -o pipefail
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench default rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 500 bash tools/probes/bench_profile.sh $PWD/$O/t_u8 > $O/t_u8_profile.log 2>&1; rc=$?; echo "t_u8 profile rc=$rc"; tail -12 $O/t_u8_profile.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 500 bash tools/probes/bench_profile.sh $PWD/$O/fp32_scan --opt scan_shadow=0 > $O/fp32_scan_profile.log 2>&1; rc=$?; echo "fp32 profile rc=$rc"; tail -12 $O/fp32_scan_profile.log
