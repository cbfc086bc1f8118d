#!/bin/bash
# Usage: bash tools/pmc_kernel.sh <tag> <kernel-substring> -- <program args...>
# Runs the program under rocprofv3 --pmc in a few passes and prints per-kernel averages of each counter.
tag=$1; pat=$2; shift 3
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_${tag}_$i -o p --output-format csv -- "$@" > /tmp/pmc_${tag}_$i.log 2>&1
done
python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
tag, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("%-28s %16.0f  (n=%d)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
PY
