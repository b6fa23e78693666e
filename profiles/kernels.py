#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 --kernel-trace run of the full bench.py (every kernel of the path: plan, half-space
generation, corridor boxes, advance): calls, average / min / max duration, registers, LDS, scratch.

usage: python profiles/kernels.py gpurun_out/prof_r02/all r02   ->  profiles/r02_kernels.md"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, name = sys.argv[1], sys.argv[2]
    tr = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)  # the newest run of the directory
    rows = list(csv.DictReader(open(tr[-1])))
    by = defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"]
        if "dlscqp" not in k:
            continue
        import re
        m = re.search(r"(dlscqp_\w+(<[^>]*>)?)", k)
        short = m.group(1) if m else k
        by[(short, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append(r)
    out = ["# rocprofv3 --kernel-trace of the full bench.py: every kernel of the path (%s)\n" % name,
           "| kernel | workgroups | calls | avg us | min us | max us | VGPR | SGPR | LDS B/WG | scratch B/lane |", "|---|---|---|---|---|---|---|---|---|---|"]
    for (k, wg), rs in sorted(by.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        if len(rs) < 3:
            continue
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rs]
        r0 = rs[0]
        out.append("| %s | %d | %d | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (k, wg, len(rs), sum(du) / len(du), min(du), max(du), r0["VGPR_Count"],
                                                                                r0["SGPR_Count"], r0["LDS_Block_Size"], r0["Scratch_Size"]))
    txt = "\n".join(out) + "\n"
    here = os.path.dirname(os.path.abspath(__file__))
    open(os.path.join(here, name + "_kernels.md"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
