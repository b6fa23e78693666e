"""Runtime reader of the reference's input formats and a mission runner ("load missions/*.json").

    python -m dlsc_gc_planner_amd.mission missions/maze10_dense/maze10_1.json --world world/maze_icra2023/dense/maze1.csv \
        --launch simulation --out result.csv

* read_mission: `Mission::readMissionFile` (/root/reference/src/mission.cpp:94-260): world box, quadrotor types (only element 0 of
  max_vel / max_acc is used, :120-124), agents (start / goal read with GetFloat, z := world_z_2d in 2-D runs), spin obstacles.
* read_world_csv: `MapManager::updateOctreeFromCSV` (src/map_manager.cpp:264-316): rows centre x,y,z, size x,y,z, stored as
  float32 points before they are rasterised.
* LAUNCH: the parameter sets of the reference's launch files.
* fly: the reference's `MultiSyncSimulator::run` loop on the device-resident closed loop (harness.ClosedLoop) with the reference's
  waypoint planner (row N4) -- until `isFinished` (multi_sync_simulator.cpp:563-585) or max_planner_iteration; writes the result
  file in the reference's layout and returns the numbers of its summary file (:852-900).
The hot path itself needs none of this: it is the harness around it, for users who come with the reference's files.
"""
import csv
import json

import numpy as np

from . import abi

# launch/*.launch: what differs between the files; everything else is common
LAUNCH = {
    "simulation": dict(dim=2, M=10, use_octomap=True, communication_range=3.0, slack_collision_weight=1.0, goal_threshold=0.1,
                       uncertainty_horizon=1.0, obs_duration=30.0, max_planner_iteration=600),
    "testall_DLSCGC_empty": dict(dim=3, M=5, use_octomap=False, communication_range=-1.0, slack_collision_weight=1.0, goal_threshold=0.1,
                                 uncertainty_horizon=1.0, obs_duration=25.0, max_planner_iteration=600),
    "testall_DLSCGC_2D": dict(dim=2, M=10, use_octomap=True, communication_range=3.0, slack_collision_weight=100.0, goal_threshold=0.1,
                              uncertainty_horizon=1.0, obs_duration=25.0, max_planner_iteration=600),
    "testall_DLSCGC_3D": dict(dim=3, M=10, use_octomap=True, communication_range=3.0, slack_collision_weight=100.0, goal_threshold=0.6,
                              uncertainty_horizon=0.7, obs_duration=60.0, max_planner_iteration=600),
}
COMMON = dict(dt=0.2, n=5, phi=3, phi_n=1, control_input_weight=0.01, terminal_weight=1.0, world_z_2d=1.0, world_resolution=0.1,
              grid_resolution=0.5, slack_threshold=0.1, reset_threshold=0.5)


def read_mission(path, world_dimension=3, world_z_2d=1.0):
    """Mission::readMissionFile.  Returns dict(world_min, world_max [3] float32; start, goal [N,3] float32; max_vel, max_acc,
    radius, downwash, nominal_velocity [N]; obstacles: the JSON's list, downwash 0 -> 1 as mission.cpp:237)."""
    doc = json.load(open(path))
    worlds = doc["world"]
    if len(worlds) != 1:
        raise ValueError("[Mission] World must have one element")
    d = worlds[0]["dimension"]
    quads = {}
    for name, q in doc["quadrotors"].items():
        quads[name] = dict(max_vel=float(q["max_vel"][0]), max_acc=float(q["max_acc"][0]), radius=float(q["radius"]),
                           downwash=float(q["downwash"]), nominal_velocity=float(q["nominal_velocity"]))
    agents = doc["agents"]
    N = len(agents)
    out = dict(world_min=np.float32(d[:3]), world_max=np.float32(d[3:]), start=np.zeros((N, 3), np.float32), goal=np.zeros((N, 3), np.float32))
    for k in ("max_vel", "max_acc", "radius", "downwash", "nominal_velocity"):
        out[k] = np.zeros(N)
    for qi, a in enumerate(agents):
        if "type" not in a or "start" not in a or "goal" not in a:
            raise ValueError("[Mission] Agent must have type, start and goal elements")
        q = dict(quads[a["type"]])
        for k in ("downwash", "nominal_velocity"):
            if k in a:
                q[k] = float(a[k])
        for key in ("start", "goal"):
            p = np.float32(a[key][:3])
            if world_dimension == 2:
                p[2] = np.float32(world_z_2d)
            out[key][qi] = p
        for k in q:
            out[k][qi] = q[k]
    obs = []
    for o in doc.get("obstacles", []):
        o = dict(o)
        if o.get("downwash", 0) == 0:
            o["downwash"] = 1
        obs.append(o)
    out["obstacles"] = obs
    return out


def read_world_csv(path):
    """Rows of a world CSV as [K, 6] float64 (centre x,y,z, size x,y,z), through float32 like the reference's point3d."""
    rows = []
    with open(path) as fh:
        for row in csv.reader(fh):
            if len(row) < 6:
                break
            rows.append([float(np.float32(float(t))) for t in row[:6]])
    return np.array(rows, np.float64).reshape(-1, 6)


def make_config(mission, launch="simulation", device=0, **overrides):
    p = dict(COMMON)
    p.update(LAUNCH[launch] if isinstance(launch, str) else launch)
    p.update(overrides)
    cfg = abi.make_config(dim=p["dim"], M=p["M"], n=p["n"], phi=p["phi"], phi_n=p["phi_n"], dt=p["dt"],
                          control_input_weight=p["control_input_weight"], terminal_weight=p["terminal_weight"],
                          slack_collision_weight=p["slack_collision_weight"], communication_range=p["communication_range"],
                          world_use_octomap=p["use_octomap"], world_min=tuple(float(v) for v in mission["world_min"]),
                          world_max=tuple(float(v) for v in mission["world_max"]), world_z_2d=p["world_z_2d"], device=device)
    return cfg, p


def fly(mission_path, world_path=None, launch="simulation", result_csv=None, device=0, max_steps=None, **overrides):
    """Fly a mission of the reference until every agent is within goal_threshold of its goal.  Returns the reference's summary
    numbers: total_flight_time, total_flight_distance, safety_ratio_agent, safety_ratio_obs, qp_failsafes, steps."""
    from .harness import ClosedLoop, SpinObstacles
    prm = dict(COMMON)
    prm.update(LAUNCH[launch] if isinstance(launch, str) else launch)
    prm.update(overrides)
    m = read_mission(mission_path, prm["dim"], prm["world_z_2d"])
    cfg, prm = make_config(m, launch, device, **overrides)
    world = None
    if prm["use_octomap"]:
        if world_path is None:
            raise ValueError("this launch uses an obstacle world: give the world CSV")
        world = dict(boxes=read_world_csv(world_path), resolution=prm["world_resolution"], grid_resolution=prm["grid_resolution"],
                     pibt="reference", uncertainty_horizon=prm["uncertainty_horizon"], slack_threshold=prm["slack_threshold"])
    dyn = None
    if m["obstacles"]:
        if any(o.get("type") != "spin" for o in m["obstacles"]):
            raise ValueError("only spin obstacles are restated (obstacle.hpp:96-150)")
        dyn = SpinObstacles(m["obstacles"]).dyn(size_prediction=1, horizon=prm["uncertainty_horizon"], duration=prm["obs_duration"])
    # the mirror takes one vehicle type per run, like the reference's grid planner (agent 0's radius and downwash, grid_based_planner.cpp:31)
    sim = ClosedLoop(cfg, m["start"], m["goal"], radius=float(m["radius"][0]), downwash=float(m["downwash"][0]), max_vel=float(m["max_vel"][0]),
                     max_acc=float(m["max_acc"][0]), nominal_velocity=float(m["nominal_velocity"][0]), waypoint_step=prm["grid_resolution"],
                     device=device, dyn=dyn, result_csv=result_csv, world=world)
    steps = prm["max_planner_iteration"] if max_steps is None else max_steps
    dist = np.zeros(len(m["start"]))
    prev = m["start"].astype(np.float64)
    worst_a, worst_o, fails, done = 1e9, 1e9, 0, None
    try:
        for k in range(steps):
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            p = sim.pos.cpu().numpy().astype(np.float64)
            dist += np.linalg.norm(p - prev, axis=1)
            prev = p
            if len(p) > 1:
                worst_a = min(worst_a, sim.min_separation())
            if dyn is not None and sim.sim_time <= prm["obs_duration"]:
                worst_o = min(worst_o, sim.min_obstacle_separation())
            if (np.linalg.norm(p - m["goal"], axis=1) <= prm["goal_threshold"]).all():  # isFinished
                done = (k + 1) * cfg.dt
                break
    finally:
        sim.close()
    return dict(total_flight_time=done, total_flight_distance=float(dist.sum()), safety_ratio_agent=worst_a,
                safety_ratio_obs=worst_o if dyn is not None else 1e9, qp_failsafes=fails, steps=k + 1, agents=len(m["start"]))


def main():
    import argparse
    ap = argparse.ArgumentParser(description="fly a dlsc_gc_planner mission file on the MI355X closed loop")
    ap.add_argument("mission")
    ap.add_argument("--world", default=None, help="world CSV (obstacle boxes); required by launches that use the octomap")
    ap.add_argument("--launch", default="simulation", choices=sorted(LAUNCH))
    ap.add_argument("--out", default=None, help="result file in the reference's log/result_*.csv layout")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(fly(a.mission, a.world, a.launch, a.out, a.device)))


if __name__ == "__main__":
    main()
