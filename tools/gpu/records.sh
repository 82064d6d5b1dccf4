#!/bin/bash
# The end-of-round records, taken on the round's last code in one GPU-box call and copied into profiles/<round>/ (tracked):
#   bench_default.json            `python bench.py` as the driver runs it (every leg: other configs, parity, CPU baseline, live PMC traffic)
#   t_u8_final/                   rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the headline line (u8 selection scan)
#   fp32_scan_final/              the same for --opt scan_shadow=0 (SURVEY 8d read literally: N*d*4 bytes per query)
#   c4_final/                     the same for --workload c4 (int8 tiles, 256-query batches)
#   usage (on the GPU box): tools/gpu/records.sh r04      (then, here: python tools/gpu/collect_records.py r04)
set -o pipefail
RND=${1:-r04}
O=gpurun_out/$RND; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "default bench rc=$?"
bash tools/gpu/profile_bench.sh $O/t_u8_final --steps 200 --warmup 20 > $O/t_u8_final.log 2>&1 || echo "t_u8 profile rc=$?"
bash tools/gpu/profile_bench.sh $O/fp32_scan_final --steps 150 --warmup 20 --opt scan_shadow=0 > $O/fp32_scan_final.log 2>&1 || echo "fp32 profile rc=$?"
bash tools/gpu/profile_bench.sh $O/c4_final --workload c4 --steps 20 --warmup 5 > $O/c4_final.log 2>&1 || echo "c4 profile rc=$?"
tail -c 600 $O/bench_default.json
