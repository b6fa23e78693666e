"""Fly empty10 / empty60 (BASELINE.json configs[0], [2]) on the device closed loop and print the p50 step time.
    python tests/tools/fly_empty.py [empty10_1|empty60_1] [steps]"""
import json
import os
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.harness import ClosedLoop

name = sys.argv[1] if len(sys.argv) > 1 else "empty60_1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 110
m = json.load(open(os.path.join("tests", "golden", "missions_empty.json")))[name]
w = m["world"]
cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                      world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]))
sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"], downwash=m["downwash"],
                 max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"])
ts = []
for k in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("%s: p50 step %.4f ms  p90 %.4f ms  (%d steps)  checksum %.6f" % (name, np.percentile(ts[5:], 50), np.percentile(ts[5:], 90), steps, float(sim.pos.sum())))
sim.close()
