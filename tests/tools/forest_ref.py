import json, os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.harness import ClosedLoop, SpinObstacles
m = json.load(open("tests/golden/world_forest1.json"))
for mode in ("reference", True):
    cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], control_input_weight=0.01, terminal_weight=1.0,
                          communication_range=m["communication_range"], slack_collision_weight=m["slack_collision_weight"],
                          world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"], pibt=mode, uncertainty_horizon=m["uncertainty_horizon"])
    spin = SpinObstacles(m["obstacles"])
    sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0, waypoint_step=m["grid_resolution"], world=world, dyn=spin.dyn(size_prediction=1, horizon=m["uncertainty_horizon"], duration=m["obs_duration"]))
    worst, wobs, fails, done, rel, mapf_fail = 1e9, 1e9, 0, None, 0, 0
    for k in range(500):
        status,_ = sim.step()
        fails += int((status != abi.OK).sum())
        worst = min(worst, sim.min_separation())
        if sim.sim_time <= m["obs_duration"]: wobs = min(wobs, sim.min_obstacle_separation())
        rel += int(sim.released.sum())
        mapf_fail += int(getattr(sim, "mapf_rc", 0) != 0)
        p = sim.pos.cpu().numpy()
        if done is None and (np.linalg.norm(p - np.array(m["goal"]), axis=1) < m["goal_threshold"]).all():
            done = (k+1)*cfg.dt; break
    sim.close()
    print("mode", mode, "done", done, "agent ratio", worst, "obstacle ratio", wobs, "fail-safes", fails, "trap releases", rel, "mapf failures", mapf_fail)
