#!/usr/bin/env python3
"""Manual tool: the inputs of a late step of the flown 2 048-agent crowd (bench.py: x2048_moving_crowd) for the agents with the most
active-set iterations and as many taken at random, read back from HBM into gpurun_out/crowd_step.npz -- to study the plan kernel's
pricing passes on the CPU (tests/tools/crowd_skip_study.py).
    python tests/tools/crowd_dump.py [--steps 40] [--agents 192]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--agents", type=int, default=192)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "crowd_step.npz"))
    a = ap.parse_args()
    import numpy as np
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.harness import ClosedLoop
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
    it = res["iters"][:, 0]
    order = np.argsort(-it, kind="stable")
    rng = np.random.default_rng(7)
    pick = np.concatenate([order[:a.agents], rng.choice(order[a.agents:], a.agents, replace=False)])
    off = [0]
    parts = []
    for i in pick:
        parts.append(rows[rows_off[i]:rows_off[i + 1]])
        off.append(off[-1] + len(parts[-1]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, recs=recs[pick], rows_off=np.array(off, np.int64), rows=np.concatenate(parts), boxes=boxes[pick],
                        iters=res["iters"][pick], status=res["status"][pick], goal_status=res["goal_status"][pick], ctrl=res["ctrl"][pick],
                        goal=res["goal"][pick], agent=pick, all_iters=res["iters"])
    print("step %d: iterations mean %.2f max %d; wrote %d agents, %d rows to %s" % (a.steps, it.mean(), it.max(), len(pick), off[-1], a.out))
    sim.close()


if __name__ == "__main__":
    main()
