#!/usr/bin/env python3
"""Copy a round's measurement records from gpurun_out/<round>/ (scratch, merged back by gpurun) into profiles/<round>/
(tracked): everything but raw traces and stderr logs of successful runs.    python tools/gpu/collect_records.py r04 [subdir ...]"""
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
only = sys.argv[2:]
SRC, DST = ROOT / "gpurun_out" / rnd, ROOT / "profiles" / rnd
KEEP = {".json", ".csv", ".txt", ".log"}
n = 0
for src in sorted(SRC.rglob("*")):
    rel = src.relative_to(SRC)
    if not src.is_file() or src.suffix not in KEEP or src.stat().st_size > 2_000_000:
        continue
    if only and rel.parts[0] not in only and str(rel) not in only:
        continue
    dst = DST / rel
    dst.parent.mkdir(parents=True, exist_ok=True)
    shutil.copyfile(src, dst)
    n += 1
print(f"copied {n} files to {DST}")
