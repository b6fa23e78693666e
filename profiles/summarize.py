#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of `bench.py` runs into profiles/<name>.md + hbm_traffic_latest.json.

usage: python profiles/summarize.py gpurun_out/prof_r01 r01 --steps 30 --agents 16384

Expects sub-directories trace/ (--kernel-trace --stats), fetch/ (--pmc FETCH_SIZE), write/
(--pmc WRITE_SIZE) and optional sq*/ (SQ counters), each written by a separate rocprofv3 run of the
same bench command.  Only the LAST `steps` dispatches of dlscqp_plan_kernel (the timed region of
bench.py; earlier ones are workload generation and warm-up) are averaged.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and on
gfx950 FETCH_SIZE reports half of the fetched bytes (MI355X_MICROARCH.md, "HBM").
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNEL = "dlscqp_plan_kernel"
HERE = os.path.dirname(os.path.abspath(__file__))


PLAN_KERNEL_SOURCES = ("gi_core.hpp", "qp_types.hpp", "dlscqp.hip up to the half-space generator (wave primitives, launch order, dlscqp_plan_kernel)")


def csrc_sha1():
    """Hash of the PLAN KERNEL's sources a profile was measured on (bench.py compares it with the tree it runs from): gi_core.hpp,
    qp_types.hpp and the part of dlscqp.hip in front of the half-space generator -- the wave primitives, the launch order and
    dlscqp_plan_kernel itself.  An edit to a generator kernel, the exchange or the host code behind that point does not
    invalidate the plan kernel's counters (round 4 lost its driver line's counter fields to such an edit)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(HERE), "dlsc_gc_planner_amd", "csrc")
    for f in ("gi_core.hpp", "qp_types.hpp"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    hip = open(os.path.join(d, "dlscqp.hip"), "rb").read()
    cut = hip.find(b"// SURVEY.md row N1: one workgroup")
    h.update(b"dlscqp.hip")
    h.update(hip if cut < 0 else hip[:cut])
    return h.hexdigest()


def dynamic_lds_bytes():
    """The plan kernel takes its LDS as DYNAMIC shared memory: the trace's LDS column shows the static part (0).  Ask the
    library what the S64 launch of bench.py asks for (needs the GPU: this runs on the box the profile was taken on)."""
    try:
        sys.path.insert(0, os.path.dirname(HERE))
        from dlsc_gc_planner_amd import workloads
        from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
        opt = BatchTrajOptimizer(workloads.s64_config())
        b = int(opt.lds_bytes(0, 120))
        opt.close()
        return b
    except Exception:
        return None


def read(path):
    with open(path) as fh:
        return list(csv.DictReader(fh))


def last_dispatches(rows, steps):
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if KERNEL in r["Kernel_Name"]})
    return set(ids[-steps:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prof_dir")
    ap.add_argument("name")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--bytes-per-qp", type=int, default=4900)
    ap.add_argument("--concurrent", type=int, default=1,
                    help="launches of the kernel that run side by side in the timed loop (bench.py's swarm groups, one stream each): device-level "
                         "rates are per-launch figures x this; the PMC passes serialise the kernels, their counters are per launch either way")
    args = ap.parse_args()
    out = []
    summary = {"name": args.name, "agents_per_launch": args.agents}
    tr = sorted(glob.glob(os.path.join(args.prof_dir, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    if tr:
        rows = read(tr[-1])
        keep = last_dispatches(rows, args.steps)
        sel = [r for r in rows if KERNEL in r["Kernel_Name"] and int(r["Dispatch_Id"]) in keep]
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in sel]
        avg = sum(durs) / len(durs)
        r0 = sel[0]
        summary.update(avg_kernel_us=avg, min_kernel_us=min(durs), max_kernel_us=max(durs), launches=len(durs),
                       vgpr=int(r0["VGPR_Count"]), sgpr=int(r0["SGPR_Count"]), lds_bytes=int(r0["LDS_Block_Size"]),
                       scratch=int(r0["Scratch_Size"]), grid=int(r0["Grid_Size_X"]), wg=int(r0["Workgroup_Size_X"]))
        out.append("## kernel trace (last %d dispatches of %s)\n" % (len(durs), KERNEL))
        out.append("| avg us | min us | max us | grid | workgroup | VGPR | SGPR | LDS B/WG | scratch B/lane |")
        out.append("|---|---|---|---|---|---|---|---|---|")
        out.append("| %.1f | %.1f | %.1f | %d | %d | %d | %d | %d | %d |\n" % (
            avg, min(durs), max(durs), summary["grid"], summary["wg"], summary["vgpr"], summary["sgpr"],
            summary["lds_bytes"], summary["scratch"]))
        dyn = dynamic_lds_bytes()
        if dyn:
            by_lds = (160 * 1024) // (dyn + summary["lds_bytes"])
            by_vgpr = 4 * max(1, 512 // max(2 * summary["vgpr"], 1))  # the trace's VGPR column counts register pairs
            summary.update(dynamic_lds_bytes=dyn, waves_per_cu=min(by_lds, by_vgpr, 32))
            out.append("dynamic LDS per workgroup (= per agent, one wavefront): %d B -> %d agents fit the 160 KiB of a CU; the registers allow "
                       "%d wavefronts per CU; resident: %d wavefronts per CU\n" % (dyn, by_lds, by_vgpr, summary["waves_per_cu"]))
        alg = args.agents * args.bytes_per_qp
        summary["algorithmic_bytes_per_launch"] = alg
        summary["concurrent_launches"] = args.concurrent
        summary["achieved_GBps"] = args.concurrent * alg / (avg * 1e-6) / 1e9
        out.append("algorithmic bytes per launch = %d agents x %d B = %d B; %d launch(es) side by side, %.1f us each on average -> %.2f GB/s "
                   "achieved (%.4f of 8000 GB/s)\n" % (args.agents, args.bytes_per_qp, alg, args.concurrent, avg, summary["achieved_GBps"],
                                                       summary["achieved_GBps"] / 8000))
        st = sorted(glob.glob(os.path.join(args.prof_dir, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if st:
            out.append("rocprofv3 --stats (all dispatches of the process, including workload generation):\n")
            out.append("```")
            out.extend(l.rstrip() for l in open(st[-1]).readlines()[:6])
            out.append("```\n")
    counters = defaultdict(list)
    for sub in sorted(glob.glob(os.path.join(args.prof_dir, "*"))):
        cc = sorted(glob.glob(os.path.join(sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not cc:
            continue
        rows = read(cc[-1])
        keep = last_dispatches(rows, args.steps)
        for r in rows:
            if KERNEL in r["Kernel_Name"] and int(r["Dispatch_Id"]) in keep:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if counters:
        out.append("## PMC counters (average per launch over the timed dispatches, one rocprofv3 run per group)\n")
        out.append("| counter | avg per launch |")
        out.append("|---|---|")
        for k in sorted(counters):
            v = sum(counters[k]) / len(counters[k])
            summary["pmc_" + k] = v
            out.append("| %s | %.6g |" % (k, v))
        out.append("")
    if "pmc_SQ_INSTS_VALU" in summary and "avg_kernel_us" in summary:
        # share of the VALU pipes' cycles that the kernel's instructions occupy: a wave64 FP64 instruction holds its SIMD's vector
        # pipe for 4 cycles, any other VALU instruction for 2 (MI355X_MICROARCH.md: SIMD-32, FP64 at half rate); 256 CUs x 4 SIMDs;
        # 2.4 GHz peak engine clock (a lower real clock makes the true share higher).  FP64 = the ADD / MUL / FMA / TRANS_F64 counters
        # (comparisons, v_max_f64 and moves of doubles are priced at 2: a lower bound)
        f64 = sum(summary.get("pmc_SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
        have_f64 = "pmc_SQ_INSTS_VALU_FMA_F64" in summary
        other = summary["pmc_SQ_INSTS_VALU"] - f64
        cyc = 4.0 * f64 + 2.0 * other if have_f64 else 4.0 * summary["pmc_SQ_INSTS_VALU"]
        frac = args.concurrent * cyc / (1024 * summary["avg_kernel_us"] * 1e-6 * 2.4e9)
        summary["valu_issue_frac"] = frac
        summary["valu_f64_insts_per_agent"] = f64 / args.agents if have_f64 else None
        out.append("VALU pipe: %d launch(es) x (4 x %.0f FP64 + 2 x %.0f other VALU instructions per agent) / (1024 SIMDs x %.1f us x 2.4 GHz) = %.3f of "
                   "the chip's vector-pipe cycles (%.0f VALU instructions per agent%s; the launch duration is the kernel trace's, where the "
                   "launches overlap; the counters are per launch)\n" % (
                       args.concurrent, f64 / args.agents, other / args.agents, summary["avg_kernel_us"], frac, summary["pmc_SQ_INSTS_VALU"] / args.agents,
                       "" if have_f64 else "; no FP64 counters in this run: every instruction priced at 4"))
        for k in ("SALU", "LDS", "SMEM", "VMEM_RD", "VMEM_WR"):
            if "pmc_SQ_INSTS_" + k in summary:
                out.append("%s instructions per agent: %.1f" % (k, summary["pmc_SQ_INSTS_" + k] / args.agents))
        out.append("")
    if "pmc_SQ_WAIT_ANY" in summary and "pmc_SQ_WAVE_CYCLES" in summary:
        out.append("SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f of the wave-cycles waiting\n" % (summary["pmc_SQ_WAIT_ANY"] / summary["pmc_SQ_WAVE_CYCLES"]))
    if "pmc_SQ_INSTS_MFMA" in summary:
        out.append("MFMA instructions per launch: %.0f (SQ_INSTS_MFMA; FP64: %s; SQ_VALU_MFMA_BUSY_CYCLES %s) -- the method has no GEMM-shaped "
                   "work: zero by design\n" % (summary["pmc_SQ_INSTS_MFMA"], summary.get("pmc_SQ_INSTS_VALU_MFMA_F64"), summary.get("pmc_SQ_VALU_MFMA_BUSY_CYCLES")))
    if "pmc_FETCH_SIZE" in summary and "pmc_WRITE_SIZE" in summary:
        hbm = (2 * summary["pmc_FETCH_SIZE"] + summary["pmc_WRITE_SIZE"]) * 1024
        summary["hbm_bytes_per_launch"] = hbm
        out.append("HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.4g B (algorithmic %.4g B, ratio %.2f)\n" % (
            hbm, summary.get("algorithmic_bytes_per_launch", float("nan")),
            hbm / summary.get("algorithmic_bytes_per_launch", float("nan"))))
    here = HERE
    with open(os.path.join(here, args.name + ".md"), "w") as fh:
        fh.write("# rocprofv3 summary %s\n\n" % args.name)
        fh.write("\n".join(out) + "\n")
    with open(os.path.join(here, args.name + ".json"), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    if "hbm_bytes_per_launch" in summary:
        with open(os.path.join(here, "hbm_traffic_latest.json"), "w") as fh:
            json.dump({"agents_per_launch": args.agents, "hbm_bytes_per_launch": summary["hbm_bytes_per_launch"],
                       "valu_issue_frac": summary.get("valu_issue_frac"), "kernel": KERNEL, "source": args.name,
                       "valu_insts_per_agent": summary.get("pmc_SQ_INSTS_VALU", 0.0) / args.agents if "pmc_SQ_INSTS_VALU" in summary else None,
                       "valu_f64_insts_per_agent": summary.get("valu_f64_insts_per_agent"),
                       "salu_insts_per_agent": summary.get("pmc_SQ_INSTS_SALU", 0.0) / args.agents if "pmc_SQ_INSTS_SALU" in summary else None,
                       "wait_any_frac": (summary["pmc_SQ_WAIT_ANY"] / summary["pmc_SQ_WAVE_CYCLES"]) if "pmc_SQ_WAIT_ANY" in summary and "pmc_SQ_WAVE_CYCLES" in summary else None,
                       "mfma_insts_per_launch": summary.get("pmc_SQ_INSTS_MFMA"), "mfma_busy_cycles_per_launch": summary.get("pmc_SQ_VALU_MFMA_BUSY_CYCLES"),
                       "csrc_sha1": csrc_sha1(), "csrc_hashed": list(PLAN_KERNEL_SOURCES)}, fh)
    print("\n".join(out))


if __name__ == "__main__":
    main()
