import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from dlsc_gc_planner_amd import abi, harness
from dlsc_gc_planner_amd.harness import ClosedLoop
g = os.path.join("tests", "golden")
w = json.load(open(os.path.join(g, "world_maze1.json")))
cfg = abi.make_config(dim=2, M=10, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=3.0,
                      world_use_octomap=True, world_min=tuple(w["world"][:3]), world_max=tuple(w["world"][3:]), world_z_2d=1.0)
world = dict(boxes=w["boxes"], resolution=w["world_resolution"], grid_resolution=w["grid_resolution"], pibt="reference")
sim = ClosedLoop(cfg, np.array(w["start"], np.float32), np.array(w["goal"], np.float32), radius=w["radius"], downwash=2.0,
                 max_vel=w["max_vel"], max_acc=w["max_acc"], nominal_velocity=1.0, waypoint_step=w["grid_resolution"], world=world)
acc = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[key] = acc.get(key, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(sim, "_waypoints_host", "waypoints_host(total)")
wrap(sim.grid, "step", "  mapf_step(C++)")
wrap(sim, "_set_comm_neighbours", "  set_comm_neighbours")
wrap(sim.gen, "set_neighbours", "    gen.set_neighbours")
wrap(sim, "_body", "body(launches)")
n = 150
for k in range(n):
    if k == 10: acc.clear(); t0 = time.perf_counter()
    sim.step(); torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per step ms: total %.3f" % (tot / (n - 10) * 1e3))
for k, v in acc.items(): print("  %-28s %.3f" % (k, v / (n - 10) * 1e3))
sim.close()
