#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (`make -C wdbx-py_amd/csrc asm`) into one line
per instantiated kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS.

    python tools/resource_usage.py [wdbx-py_amd/csrc/wdbx_hip.resource.txt] > profiles/r02/resource_usage.txt

Exit code 1 when any kernel uses scratch (so the build can assert "no spills")."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "wdbx-py_amd" / "csrc" / "wdbx_hip.resource.txt"
blocks = re.split(r"remark: [^\n]*Function Name: ", src.read_text())[1:]
names = [b.split("\n")[0].strip() for b in blocks]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
except Exception:
    dem = names


def field(block, key):
    m = re.search(re.escape(key) + r": (\d+)", block)
    return int(m.group(1)) if m else -1


rows = []
for name, block in zip(dem, blocks):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((GemmArgs|ScanArgs|Scan8Args|MergeArgs|Gemm8Args)\)$", "", name)
    rows.append((name, field(block, "VGPRs"), field(block, "AGPRs"), field(block, "SGPRs"),
                 field(block, "ScratchSize [bytes/lane]"), field(block, "Occupancy [waves/SIMD]"),
                 field(block, "LDS Size [bytes/block]")))
rows.sort()
spill = [r for r in rows if r[4] > 0]
print(f"# {len(rows)} kernels, {len(spill)} with scratch; columns: VGPR AGPR SGPR scratch_B_per_lane waves_per_SIMD static_LDS_B  kernel")
for r in rows:
    print(f"{r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:5d} {r[5]:2d} {r[6]:6d}  {r[0]}")
sys.exit(1 if spill else 0)
