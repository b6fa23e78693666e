#!/usr/bin/env python3
"""Developer tool (CPU only): host time of the waypoint planner (csrc/grid_mapf.cpp, dlscqp_mapf_step) per replanning step in the
forest10_spin4_100 world, agents walking to their waypoints at top speed, the four obstacles spinning.
    python tests/tools/mapf_time.py [--steps 80]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dlsc_gc_planner_amd import abi  # noqa: E402
from dlsc_gc_planner_amd.harness import MapfWaypoints, SpinObstacles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--world", default="world_forest1.json")
    a = ap.parse_args()
    m = json.load(open(os.path.join(ROOT, "tests", "golden", a.world)))
    m = dict(dict(dim=2, M=10, dt=0.2, communication_range=3.0, downwash=2.0), **m)  # (the maze file: the launch file's values)
    cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], communication_range=m["communication_range"], world_use_octomap=True,
                          world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    mp = MapfWaypoints(cfg, m["boxes"], m["world_resolution"], m["grid_resolution"], start, goal, m["radius"], m["downwash"],
                       m.get("uncertainty_horizon", 1.0))
    spin = SpinObstacles(m["obstacles"]) if m.get("obstacles") else None
    pos, cur, wp = start.copy(), start.copy(), start.copy()
    N = len(start)
    ts = []
    for k in range(a.steps):
        plans = np.repeat(np.repeat(pos[:, None, None, :], cfg.M, 1), cfg.n + 1, 2) if k else None
        dyn = None
        if spin is not None:
            p, v = spin.state(k * cfg.dt)
            dyn = dict(position=p, velocity=v, radius=spin.radius, max_acc=spin.max_acc)
        t0 = time.perf_counter()
        mp.step(pos, cur, wp, plans, dyn=dyn, alert=np.zeros((N, len(spin.radius)), np.uint8) if spin is not None and k else None)
        ts.append((time.perf_counter() - t0) * 1e3)
        cur = wp.copy()
        d = wp - pos
        dist = np.linalg.norm(d, axis=1, keepdims=True)
        pos = (pos + d * np.minimum(1.0, 0.2 / np.maximum(dist, 1e-9))).astype(np.float32)
    ts = np.array(ts)
    print("%s: %d agents, %d steps: dlscqp_mapf_step p50 %.3f ms, mean %.3f, p90 %.3f, max %.3f (first %.3f)" %
          (a.world, N, a.steps, np.median(ts), ts.mean(), np.percentile(ts, 90), ts.max(), ts[0]))
    print("agents at goal:", int((np.linalg.norm(pos - goal, axis=1) < 0.3).sum()))
    mp.close()


if __name__ == "__main__":
    main()
