#!/usr/bin/env python3
"""Diagnostic (-DDLSCQP_STAMPS build): the flown 2 048-agent crowd's slowest agents, step by step -- how long the slowest agent-step is
against the mean, and how well the previous step's iteration counts predict who it will be (what a launch of the predicted-heavy
agents in the build with helper wavefronts would need).    python tests/tools/crowd_tail.py [--warm 20] [--steps 60]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import stamps as S  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm", type=int, default=20)
    ap.add_argument("--steps", type=int, default=60)
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
    stamps = torch.zeros((N, 12), dtype=torch.int64, device="cuda")
    sim.batch.c.debug_stamps = stamps.data_ptr()
    cyc, its = [], []
    for _ in range(a.steps):
        stamps.zero_()
        sim.step()
        torch.cuda.synchronize()
        cyc.append(stamps.cpu().numpy().astype(np.float64).sum(1))
        its.append(sim.batch.results()["iters"][:, 0].copy())
    cyc, its = np.array(cyc), np.array(its)
    print("per step: slowest agent %.0f k cycles (median over steps; max %.0f k), mean agent %.0f k, sum / 1024 slots %.0f k; iterations of the slowest: median %d"
          % (np.median(cyc.max(1)) / 1e3, cyc.max() / 1e3, cyc.mean() / 1e3, np.median(cyc.sum(1)) / 1024e3, np.median(its[np.arange(len(its)), cyc.argmax(1)])))
    for k in (32, 64, 128, 256, 512):
        rest = []
        for t in range(1, len(cyc)):
            heavy = np.argsort(-its[t - 1], kind="stable")[:k]
            m = np.ones(N, bool)
            m[heavy] = False
            rest.append(cyc[t][m].max() / cyc[t].max())
        print("the %3d agents with the most iterations in the previous step taken out: slowest of the rest / slowest of all, median %.2f, quartiles %.2f %.2f"
              % (k, np.median(rest), np.percentile(rest, 25), np.percentile(rest, 75)))
    # cycles against iterations: a line
    A = np.stack([its.ravel(), np.ones(its.size)], 1)
    coef = np.linalg.lstsq(A, cyc.ravel(), rcond=None)[0]
    print("cycles ~ %.0f per iteration + %.0f" % (coef[0], coef[1]))
    sim.close()
