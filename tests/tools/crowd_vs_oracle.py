#!/usr/bin/env python3
"""Manual tool: the flown 2 048-agent crowd (bench.py: x2048_moving_crowd) at a late step, EVERY agent re-planned by the CPU oracle
from the step's own inputs read back from HBM (rows after the generator's presolve, corridor boxes, agent records).
    python tests/tools/crowd_vs_oracle.py [--steps 40]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import numpy as np
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.harness import ClosedLoop
    import oracle_lib as O
    cfg = workloads.x2048_config()
    N = 2048
    start, goal = workloads.swapping_pairs(cfg, N)
    sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5, neighbours="device", nbr_pitch=256,
                     corridor_margin=0.5, prune=True, order_hint=True)
    sim.keep_inputs = True
    for k in range(a.steps):
        sim.step()
    res = sim.batch.results()
    recs, rows_off, rows, boxes, _ = sim.step_inputs()
    t0 = time.time()
    sm = gm = 0
    worst = 0.0
    n_ok = 0
    for i in range(N):
        r = rows[rows_off[i]:rows_off[i + 1]]
        ref = O.plan_batch(cfg, recs[i:i + 1], np.array([0, len(r)], np.int64), r, boxes[i:i + 1], 0, True)
        sm += int(ref["status"][0] != res["status"][i] or ref["goal_status"][0] != res["goal_status"][i])
        gm += int((ref["goal"][0] != res["goal"][i]).any() and ref["goal_status"][0] != abi.INFEASIBLE)
        if ref["status"][0] == abi.OK and res["status"][i] == abi.OK:
            n_ok += 1
            worst = max(worst, float(np.abs(ref["ctrl"][0].astype(np.float64) - res["ctrl"][i]).max()))
    print("step %d of the flown crowd: %d agents, %d solved, rows per agent %.0f, mean iterations %.2f (max %d): verdict mismatches %d, sub-goal "
          "mismatches %d, max |ctrl - oracle| %.2e (oracle %.0f s)" % (a.steps, N, n_ok, np.diff(rows_off).mean(), res["iters"][:, 0].mean(),
                                                                      res["iters"][:, 0].max(), sm, gm, worst, time.time() - t0))
    sim.close()


if __name__ == "__main__":
    main()
