#!/usr/bin/env python3
"""Developer tool: p50 of a replanning step of the small missions, eager, with and without the fused step (no stage timers: they
keep the stages apart).    python tests/tools/fused_time.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dlsc_gc_planner_amd import abi  # noqa: E402
from dlsc_gc_planner_amd.harness import ClosedLoop  # noqa: E402
import test_gpu_closed_loop as T  # noqa: E402


def missions():
    cfg, m = T._mission("empty10_1")
    yield "empty10", cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), dict(
        radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"])
    cfg, m = T._mission("empty60_1")
    yield "empty60", cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), dict(
        radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"])
    m, cfg, world = T._maze()
    yield "maze10", cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), dict(
        radius=m["radius"], downwash=2.0, max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0, waypoint_step=m["grid_resolution"],
        world=dict(world, pibt="reference"))


for name, cfg, start, goal, kw in missions():
    for fused, graph in (("0", False), ("1", False), ("0", True), ("1", True)):
        os.environ["DLSCQP_FUSED_STEP"] = fused
        sim = ClosedLoop(cfg, start, goal, **kw)
        ts = []
        for k in range(120):
            if graph and k == 3:
                sim.enable_graph()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%-8s fused %s graph %-5s: p50 %.4f ms  p10 %.4f  (fused steps %d)" % (name, fused, graph, float(np.percentile(ts[10:], 50)), float(np.percentile(ts[10:], 10)), sim.fused_steps_done), flush=True)
        sim.close()
