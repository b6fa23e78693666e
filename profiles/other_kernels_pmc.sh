#!/bin/bash
# PMC counters of the other kernels of the path (half-space generator on the stress batch, corridor kernel on the maze mission),
# one rocprofv3 run per counter group, the program itself after "--":   bash profiles/other_kernels_pmc.sh r04  ->  profiles/r04_other_kernels_pmc.md
set -u
NAME=${1:-r04}
export TMPDIR=/tmp
O=gpurun_out/other_pmc
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/lsc1 -- python3 tests/tools/lsc_time.py > $O/lsc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $O/lsc2 -- python3 tests/tools/lsc_time.py > $O/lsc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/sfc1 -- python3 tests/tools/fly_maze.py maze 80 > $O/sfc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $O/sfc2 -- python3 tests/tools/fly_maze.py maze 80 > $O/sfc2.log 2>&1
python3 - <<PY
import csv, glob, collections
out = ["# PMC counters of the other kernels (average per launch; rocprofv3, one run per counter group)", ""]
for tag, kern, grid in (("lsc", "lsc_kernel", "max"), ("sfc", "sfc_kernel", None)):
    acc = collections.defaultdict(list)
    dur = []
    if grid == "max":  # the launches with the most threads: the stress batch
        f = sorted(glob.glob("$O/%s1/*/*kernel_trace.csv" % tag))[-1]
        grid = str(max(int(r["Grid_Size_X"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]))
    for d in (tag + "1", tag + "2"):
        f = sorted(glob.glob("$O/%s/*/*counter_collection.csv" % d))[-1]
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        f = sorted(glob.glob("$O/%s/*/*kernel_trace.csv" % d))[-1]
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and (grid is None or r["Grid_Size_X"] == grid):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    us = sum(dur) / max(len(dur), 1)
    out.append("## dlscqp_%s (%s), %.1f us per launch under the profiler" % (kern, "2 048 workgroups, %s threads in all, stress batch" % grid if tag == "lsc" else "10 workgroups of 256 threads, maze mission", us))
    out.append("")
    out.append("| counter | per launch |")
    out.append("|---|---|")
    for k in sorted(acc):
        out.append("| %s | %.4g |" % (k, sum(acc[k]) / len(acc[k])))
    if "SQ_INSTS_VALU" in acc and us > 0:
        v = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"])
        simds = 1024 if tag == "lsc" else 40
        out.append("")
        out.append("VALU issue: SQ_INSTS_VALU x 4 cycles / (%d SIMDs x %.1f us x 2.4 GHz) = %.2f of the issue slots of the SIMDs the launch can occupy" % (simds, us, v * 4 / (simds * us * 2400)))
    if "SQ_WAIT_ANY" in acc:
        out.append("SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (sum(acc["SQ_WAIT_ANY"]) / sum(acc["SQ_WAVE_CYCLES"])))
    out.append("")
open("profiles/${NAME}_other_kernels_pmc.md", "w").write("\n".join(out))
print("\n".join(out))
PY
