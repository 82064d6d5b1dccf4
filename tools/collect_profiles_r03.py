#!/usr/bin/env python3
"""Copy the round-3 measurement records from gpurun_out/r03 (scratch) into profiles/r03 (tracked)."""
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC, DST = ROOT / "gpurun_out" / "r03", ROOT / "profiles" / "r03"
WANT = [
    "bench_default.json",
    "t_u8/kernel_stats.csv", "t_u8/pmc_summary.json", "t_u8/bench_under_rocprof.json",
    "fp32_scan/kernel_stats.csv", "fp32_scan/pmc_summary.json", "fp32_scan/bench_under_rocprof.json",
    "c4_i8/c4_i8_kernel_stats.csv", "c4_i8/c4_i8_pmc_summary.json", "c4_i8/sq_summary.txt", "c4_i8/bench_under_rocprof.json",
    "c4_i8/bench_batch_256.json", "c4_i8/ab_0_6_7.json", "c4_i8/ab_0_6.json", "c4_i8/clock/summary.txt",
    "batch_sizes/bench_batch_c3_l2_256_i8.json", "batch_sizes/bench_batch_c3_l2_256_bf16.json",
    "scan8_grid/one_grid_10000000.json", "scan8_grid/per_query_10000000.json", "scan8_grid/one_grid_1250000.json",
    "scan8_grid/per_query_1250000.json", "scan8_grid/one_grid_c2.json", "scan8_grid/one_grid_c3.json",
    "group_host/index_1250000.json", "group_host/group_1250000.json", "group_host/group_1250000_b.json",
    "group_host/launcher_1rank_1250000.json", "group_host/group_8shards_one_gpu_10m.json", "group_host/group_8shards_one_gpu_10m_b.json",
    "bench_group1.json", "bench_g4copy.json", "gputests_full.log", "smoke.log", "bench_default_second_run.json", "bench_default_earlier_box.json",
    # second half of the round (second selection stage, listed repairs, block threshold search, live PMC traffic): final code
    "c4_i8/refine_ab_0_1.json", "c4_i8/refine_ab_1_1.json", "c4_i8/refine_ab_0_2.json", "c4_i8/refine_ab_1_2.json", "c4_i8/c4_live.json",
    "refine/c4_kernel_stats.csv",
    "c4_final/kernel_stats.csv", "c4_final/pmc_summary.json", "c4_final/bench_under_rocprof.json",
    "t_u8_final/kernel_stats.csv", "t_u8_final/pmc_summary.json", "t_u8_final/bench_under_rocprof.json",
    "fp32_scan_final/kernel_stats.csv", "fp32_scan_final/pmc_summary.json", "fp32_scan_final/bench_under_rocprof.json",
]
for rel in WANT:
    src = SRC / rel
    if not src.exists():
        print("missing", rel)
        continue
    dst = DST / rel
    dst.parent.mkdir(parents=True, exist_ok=True)
    shutil.copyfile(src, dst)
print("copied to", DST)
