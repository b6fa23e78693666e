#!/usr/bin/env python3
"""Diagnostic: where one wavefront of dlscqp_plan_kernel spends its cycles, by phase.

Builds (if needed) and loads the -DDLSCQP_STAMPS variant of the HIP library, runs the S64 workload
and prints the share of wave-cycles per phase of plan_agent (s_memtime stamps, gi_core.hpp w.stamp).
The stamped build is slower than the product build and is never shipped; read shares, not totals.

    python profiles/stamps.py [--swarms 256]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the product's flags (__graft_entry__.HIPCC_FLAGS: the compiler's default scheduling strategy since round 6; DLSCQP_DIAG_ILP=1: rounds
# 2-5's iterative-ilp, with which this compiler's register allocator crashes on the S64 kernel once the diagnostic statements are in it)
DIAG_SCHED = ["-Wno-pass-failed"] if not os.environ.get("DLSCQP_DIAG_ILP") else ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
LIB = os.environ.get("DLSCQP_STAMPS_LIB", os.path.join(CSRC, "libdlscqp_stamps.so"))

PHASES = ["agent+goalLP", "init(x0,bounds)", "scan+argmax", "decode,G rows asked", "d1=Z'a", "rv=S d1",
          "a'Ga,dx=Ga-Z d1,a.dx", "ratio test", "primal/dual step", "add row", "givens drop", "readout"]


def ensure_lib():
    """(re)build the stamped library when a source is newer than it"""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):  # never a stale diagnostic build
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", *DIAG_SCHED, "-DDLSCQP_STAMPS", "-o", LIB, os.path.join(CSRC, "dlscqp.hip"),
                               os.path.join(CSRC, "qp_tables.cpp"), os.path.join(CSRC, "sfc_world.cpp"), os.path.join(CSRC, "grid_mapf.cpp")])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=256)
    args = ap.parse_args()
    ensure_lib()
    os.environ["DLSCQP_LIB"] = LIB
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, seed=2000)
    db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
    if not os.environ.get("DLSCQP_NO_ROW_PITCH"):
        db.use_row_pitch()  # as bench.py launches the workload: every agent's rows at a fixed pitch
    N = len(wl["agents"])
    stamps = torch.zeros((N, 12), dtype=torch.int64, device="cuda")
    db.c.debug_stamps = stamps.data_ptr()
    for _ in range(3):
        db.launch()
    res = db.results()
    st = stamps.cpu().numpy().astype(np.float64)
    it = res["iters"]
    tot = st.sum()
    print("agents %d  mean cycles/wave %.0f  (iterations mean %.2f max %d)" % (N, st.sum(1).mean(), it[:, 0].mean(), it[:, 0].max()))
    print("%-24s %8s %12s" % ("phase", "share", "cycles/wave"))
    for p, name in enumerate(PHASES):
        print("%-24s %7.1f%% %12.0f" % (name, 100 * st[:, p].sum() / tot, st[:, p].mean()))
    # per-iteration cost from agents with many iterations
    big = it[:, 0] >= 10
    if big.any():
        print("\nagents with >= 10 iterations: %d; cycles per inner iteration by phase:" % big.sum())
        iters = it[big, 0].astype(np.float64)
        for p in range(2, 11):
            print("  %-22s %10.0f" % (PHASES[p], (st[big, p] / iters).mean()))
        print("  total per iteration     %10.0f" % (st[big, 2:11].sum(1) / iters).mean())
    worst = int(np.argmax(it[:, 0]))
    print("\nslowest agent: %d iterations (%d adds, %d drops, %d active at exit), %.0f cycles; per phase:" % (
        it[worst, 0], it[worst, 1], it[worst, 2], it[worst, 3], st[worst].sum()))
    for p, name in enumerate(PHASES):
        print("  %-22s %10.0f  (%.0f per iteration)" % (name, st[worst, p], st[worst, p] / max(it[worst, 0], 1)))
    zero = it[:, 0] == 0
    print("\nagents with 0 iterations: %d (%.0f%%), mean cycles %.0f" % (zero.sum(), 100 * zero.mean(), st[zero].sum(1).mean()))
    opt.close()


if __name__ == "__main__":
    main()
