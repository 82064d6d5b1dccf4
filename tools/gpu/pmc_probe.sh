#!/bin/bash
# usage: pmc_probe.sh <outdir> <kernel-name-substring> <program> [args...] ; one rocprofv3 --pmc pass per counter group
set -e
OUT=$1; KERN=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- "$@" > $OUT/run$i.log 2>&1
done
python3 - $OUT $KERN <<'PY'
import sys, csv, glob, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, v in agg.items():
        line = "%-32s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v))
        print(line); fh.write(line + "\n")
PY
