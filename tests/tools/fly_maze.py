"""Fly the maze10 mission (or forest with 'forest') on the device closed loop and print the p50 step time; for A/B runs of a
library variant (DLSCQP_LIB=...) under rocprofv3 --kernel-trace --stats.   python tests/tools/fly_maze.py [maze|forest] [steps]"""
import json
import os
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.harness import ClosedLoop, SpinObstacles

which = sys.argv[1] if len(sys.argv) > 1 else "maze"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
g = os.path.join("tests", "golden")
if which == "maze":
    w = json.load(open(os.path.join(g, "world_maze1.json")))
    cfg = abi.make_config(dim=2, M=10, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=3.0,
                          world_use_octomap=True, world_min=tuple(w["world"][:3]), world_max=tuple(w["world"][3:]), world_z_2d=1.0)
    world = dict(boxes=w["boxes"], resolution=w["world_resolution"], grid_resolution=w["grid_resolution"], pibt="reference")
    sim = ClosedLoop(cfg, np.array(w["start"], np.float32), np.array(w["goal"], np.float32), radius=w["radius"], downwash=2.0,
                     max_vel=w["max_vel"], max_acc=w["max_acc"], nominal_velocity=1.0, waypoint_step=w["grid_resolution"], world=world)
else:
    f = json.load(open(os.path.join(g, "world_forest1.json")))
    cfg = abi.make_config(dim=f["dim"], M=f["M"], dt=f["dt"], control_input_weight=0.01, terminal_weight=1.0,
                          slack_collision_weight=f["slack_collision_weight"], communication_range=f["communication_range"],
                          world_use_octomap=True, world_min=tuple(f["world"][:3]), world_max=tuple(f["world"][3:]))
    spin = SpinObstacles(f["obstacles"])
    world = dict(boxes=f["boxes"], resolution=f["world_resolution"], grid_resolution=f["grid_resolution"], pibt="reference",
                 uncertainty_horizon=f["uncertainty_horizon"])
    sim = ClosedLoop(cfg, np.array(f["start"], np.float32), np.array(f["goal"], np.float32), radius=f["radius"], downwash=f["downwash"],
                     max_vel=f["max_vel"], max_acc=f["max_acc"], nominal_velocity=1.0, waypoint_step=f["grid_resolution"],
                     world=world, dyn=spin.dyn(horizon=f["uncertainty_horizon"], duration=f["obs_duration"]))
ts = []
for k in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("%s: p50 step %.3f ms  p90 %.3f ms  (%d steps)  final positions checksum %.6f" %
      (which, np.percentile(ts[5:], 50), np.percentile(ts[5:], 90), steps, float(sim.pos.sum())))
sim.close()
