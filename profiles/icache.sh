#!/bin/bash
# Instruction-cache and issue-side counters of the plan kernel (one rocprofv3 pass per group, --kernel-trace only).
set -u
NAME=${1:-r02_icache}
OUT=gpurun_out/prof_$NAME
export TMPDIR=/tmp
B="python3 bench.py --steps 30 --warmup 4 --swarms 512 --headline-only"
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/ic1 -- $B > $OUT/ic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/ic2 -- $B > $OUT/ic2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/ic3 -- $B > $OUT/ic3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES --output-format csv -d $OUT/ic4 -- $B > $OUT/ic4.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
c = defaultdict(list)
for sub in sorted(glob.glob(os.path.join(out, "ic*"))):
    for f in glob.glob(os.path.join(sub, "*", "*_counter_collection.csv")):
        rows = list(csv.DictReader(open(f)))
        ids = sorted({int(r["Dispatch_Id"]) for r in rows if "dlscqp_plan_kernel" in r["Kernel_Name"]})[-30:]
        for r in rows:
            if "dlscqp_plan_kernel" in r["Kernel_Name"] and int(r["Dispatch_Id"]) in ids:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(c):
    print("%-32s %.6g" % (k, sum(c[k]) / len(c[k])))
PY
