#!/usr/bin/env python3
"""Diagnostic: when and where every wavefront of one dlscqp_plan_kernel launch ran (-DDLSCQP_TIMELINE build, never shipped).

s_memrealtime (100 MHz) at the start and the end of every agent's solve plus HW_ID / XCC_ID: how full the wave slots are over
the launch, how long a slot stays empty between two wavefronts (workgroup dispatch), how long the tail is.

    python profiles/timeline.py [--swarms 512] [--grid 0]
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
LIB = os.path.join(CSRC, "libdlscqp_timeline.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=512)
    ap.add_argument("--workload", default="s64", choices=["s64", "x2048"])
    args = ap.parse_args()
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):  # never a stale diagnostic build
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               *DIAG_SCHED, "-DDLSCQP_TIMELINE", "-I" + os.path.join(ROOT, "include"), "-o", LIB]
                              + [os.path.join(CSRC, f) for f in ("dlscqp.hip", "qp_tables.cpp", "sfc_world.cpp", "grid_mapf.cpp")])
    os.environ["DLSCQP_LIB"] = LIB
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    if args.workload == "x2048":
        return x2048(np, torch)
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, seed=2000)
    db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
    if not os.environ.get("DLSCQP_NO_ROW_PITCH"):
        db.use_row_pitch()  # as bench.py launches the workload: every agent's rows at a fixed pitch
    N = len(wl["agents"])
    stamps = torch.zeros((N, 12), dtype=torch.int64, device="cuda")
    db.launch()
    it = db.results()["iters"][:, 0]
    order = torch.from_numpy(np.argsort(-it, kind="stable").astype(np.int32)).cuda()  # expensive agents first, as the bench does
    db.c.agent_order = order.data_ptr()
    db.c.debug_stamps = stamps.data_ptr()
    for _ in range(3):
        db.launch()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy()
    t0, t1, hw, xcc, slot = st[:, 0], st[:, 1], st[:, 2], st[:, 3] & 15, st[:, 4]
    tick = 1e-2  # us per tick (100 MHz)
    span = (t1.max() - t0.min()) * tick
    dur = (t1 - t0) * tick
    place = (xcc << 32) | (hw & 0xffff)  # XCC, SE, SH, CU, pipe, SIMD, wave slot
    n_slots = len(np.unique(place))
    print("agents %d, launch span %.1f us, wave slots used %d, mean solve %.2f us, sum of solves / (slots x span) = %.3f"
          % (N, span, n_slots, dur.mean(), dur.sum() / (n_slots * span)))
    res = db.results()["iters"]
    for a in np.argsort(-dur)[:6]:
        print("  agent %6d: %6.1f us from %6.1f us after launch start, %2d iterations (%d adds, %d drops, %d active at exit)" %
              (a, dur[a], (t0[a] - t0.min()) * tick, res[a, 0], res[a, 1], res[a, 2], res[a, 3]))
    for lo, hi in ((0, 0), (1, 3), (4, 9), (10, 19), (20, 99)):
        m = (res[:, 0] >= lo) & (res[:, 0] <= hi)
        if m.any():
            print("  %2d-%2d iterations: %5d agents, mean solve %6.1f us, %4.1f %% of the slot time" % (lo, hi, m.sum(), dur[m].mean(), 100 * dur[m].sum() / dur.sum()))
    gaps, first, last = [], [], []
    for p in np.unique(place):
        sel = np.where(place == p)[0]
        o = sel[np.argsort(t0[sel])]
        gaps.append((t0[o][1:] - t1[o][:-1]) * tick)
        first.append((t0[o][0] - t0.min()) * tick)
        last.append((t1.max() - t1[o][-1]) * tick)
    gaps = np.concatenate(gaps)
    print("gap between two wavefronts on one slot: p10 %.2f  p50 %.2f  p90 %.2f  mean %.2f us (n = %d); per agent: %.1f %% of the mean solve"
          % (np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90), gaps.mean(), len(gaps), 100 * gaps.mean() / dur.mean()))
    print("slot first used after launch start: p50 %.2f  max %.2f us;  slot idle before launch end (tail): p50 %.2f  p90 %.2f  max %.2f us"
          % (np.percentile(first, 50), np.max(first), np.percentile(last, 50), np.percentile(last, 90), np.max(last)))
    # concurrency over the launch
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    tt = (ev[:, 0] - t0.min()) * tick
    for frac in (0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99):
        i = np.searchsorted(tt, frac * span)
        print("  at %2.0f %% of the span: %4d wavefronts in flight" % (100 * frac, conc[min(i, len(conc) - 1)]))
    opt.close()


def x2048(np, torch):
    """The stress batch (2 048 agents = one wavefront per wave slot): the launch is as long as its slowest agent."""
    sys.argv = ["bench.py"]
    import bench
    ss = bench.StressStep(0, 0, 1, None)
    ss.time(3, False)
    gen, db, _, sel = ss.parts[0]
    stamps = torch.zeros((ss.N, 12), dtype=torch.int64, device="cuda")
    db.c.debug_stamps = stamps.data_ptr()
    for _ in range(3):
        db.launch(True, True)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy()
    it = db.results()["iters"]
    cnt = gen.row_counts.cpu().numpy() if getattr(gen, "row_counts", None) is not None else np.diff(gen.row_offsets)
    dur = (st[:, 1] - st[:, 0]) * 1e-2
    span = (st[:, 1].max() - st[:, 0].min()) * 1e-2
    print("agents %d, launch span %.1f us; solve us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" %
          (ss.N, span, dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
    print("rows per agent: mean %.0f max %d; iterations: mean %.2f max %d" % (cnt.mean(), cnt.max(), it[:, 0].mean(), it[:, 0].max()))
    order = np.argsort(-dur)[:8]
    for a in order:
        print("  agent %4d: %6.1f us, %2d iterations (%d adds, %d drops), %4d rows -> %.1f us per pricing pass" %
              (a, dur[a], it[a, 0], it[a, 1], it[a, 2], cnt[a], dur[a] / (it[a, 0] + 1)))
    z = it[:, 0] == 0
    if z.any():
        print("agents with 0 iterations: %d, mean %.1f us (rows %.0f)" % (z.sum(), dur[z].mean(), cnt[z].mean()))
    # us per pass against rows
    x, y = cnt.astype(float), dur / (it[:, 0] + 1)
    A = np.stack([np.ones_like(x), x], 1)
    c = np.linalg.lstsq(A, y, rcond=None)[0]
    print("least squares: us per pricing pass = %.2f + %.4f x rows" % (c[0], c[1]))
    ss.close()


if __name__ == "__main__":
    main()
