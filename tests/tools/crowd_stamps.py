#!/usr/bin/env python3
"""Diagnostic: profiles/stamps.py's table (cycles per phase of plan_agent, -DDLSCQP_STAMPS build) for the flown 2 048-agent crowd
of bench.py (moving_crowd): several steps of the flight, summed.    python tests/tools/crowd_stamps.py [--warm 25] [--steps 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import stamps as S  # noqa: E402  (LIB, PHASES, ensure_lib)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm", type=int, default=25)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    S.ensure_lib()
    os.environ["DLSCQP_LIB"] = S.LIB
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.harness import ClosedLoop
    cfg = workloads.x2048_config()
    N = 2048
    start, goal = workloads.swapping_pairs(cfg, N)
    sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5,
                     neighbours="device", nbr_pitch=256, corridor_margin=0.5, prune=True, order_hint=True)
    for _ in range(a.warm):
        sim.step()
    st_all, it_all = [], []
    stamps = torch.zeros((N, 12), dtype=torch.int64, device="cuda")
    sim.batch.c.debug_stamps = stamps.data_ptr()
    for _ in range(a.steps):
        stamps.zero_()
        sim.step()
        torch.cuda.synchronize()
        st_all.append(stamps.cpu().numpy().astype(np.float64))
        it_all.append(sim.batch.results()["iters"].copy())
    st, it = np.concatenate(st_all), np.concatenate(it_all)
    tot = st.sum()
    print("crowd: %d agent-steps, mean cycles/wave %.0f (iterations mean %.2f max %d), rows kept %.0f" % (
        len(st), st.sum(1).mean(), it[:, 0].mean(), it[:, 0].max(), float(sim.row_counts.float().mean())))
    print("%-24s %8s %12s %14s" % ("phase", "share", "cycles/wave", "per iteration"))
    iters = max(it[:, 0].sum(), 1)
    for p, name in enumerate(S.PHASES):
        print("%-24s %7.1f%% %12.0f %14.0f" % (name, 100 * st[:, p].sum() / tot, st[:, p].mean(), st[:, p].sum() / iters if 2 <= p <= 10 else 0))
    zero = it[:, 0] == 0
    print("agents with 0 iterations: %.0f%%, mean cycles %.0f; slowest agent-step %.0f cycles at %d iterations" % (
        100 * zero.mean(), st[zero].sum(1).mean() if zero.any() else 0, st.sum(1).max(), it[int(np.argmax(st.sum(1))), 0]))
    sim.close()
