#!/usr/bin/env python3
"""Developer tool: the stationary stress batch and the empty60 mission under a given DLSCQP_WPA / DLSCQP_LIB."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dlsc_gc_planner_amd import abi  # noqa: E402
from dlsc_gc_planner_amd.harness import ClosedLoop  # noqa: E402

ss = bench.StressStep(0, 0, 1, None)
ss.time(30, False)
x = ss.time(60, False)
ss.close()
m = json.load(open(os.path.join(ROOT, "tests", "golden", "missions_empty.json")))["empty60_1"]
w = m["world"]
cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                      world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]))
sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"], downwash=m["downwash"],
                 max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"])
for _ in range(10):
    sim.step()
sim.enable_stage_timers()
ts = []
for _ in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
st = sim.stage_times()
print("WPA=%s lib=%s: x2048 stationary %.4f ms/step; empty60 p50 %.4f ms, traj_optimization %.4f ms" % (
    os.environ.get("DLSCQP_WPA", "auto"), os.path.basename(os.environ.get("DLSCQP_LIB", "libdlscqp.so")), x, np.percentile(ts, 50),
    np.percentile(st["traj_optimization_time"], 50)))
sim.close()
