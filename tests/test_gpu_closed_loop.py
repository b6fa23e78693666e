"""-m gpu, row N3 (first slice): the reference's empty-world missions flown in closed loop entirely on the device
(half-space generation -> goal LP -> QP -> state update).  System-level properties the reference's own batch
runs check (log/summary_*.csv: safety_ratio_agent >= 1, goals reached)."""
import json
import os

import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.harness import ClosedLoop

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _mission(name):
    m = json.load(open(os.path.join(HERE, "golden", "missions_empty.json")))[name]
    w = m["world"]
    # launch/testall_DLSCGC_empty.launch: 3-D, M = 5, dt = 0.2, no octomap, unlimited communication range
    cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                          world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]))
    return cfg, m


@pytest.mark.parametrize("name,steps", [("empty10_1", 120), ("empty60_1", 100)])
def test_empty_mission_closed_loop_is_collision_free(gpu_required, name, steps):
    cfg, m = _mission(name)
    sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"],
                     downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"])
    try:
        d0 = float(np.linalg.norm(np.array(m["goal"]) - np.array(m["start"]), axis=1).mean())
        worst, fails = 1e9, 0
        for _ in range(steps):
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            worst = min(worst, sim.min_separation())
        dist = (sim.pos - sim.desired_goal).norm(dim=1).cpu().numpy()
    finally:
        sim.close()
    print(name, "min safety ratio", worst, "QP fail-safes", fails, "mean dist to goal", dist.mean(), "of", d0)
    assert worst >= 0.999  # safety_ratio_agent of the reference's summary (1.00058 in its committed run)
    assert fails <= 0.02 * steps * len(dist)
    if name == "empty10_1":
        # goal_threshold 0.1 in the launch file.  Without the grid planner (row N4) two agents of this mission block
        # each other head-on and wait -- the deadlock DLSC-GC resolves with PIBT waypoints -- the rest must arrive
        assert (dist < 0.15).mean() >= 0.7
    else:
        assert dist.mean() < 0.7 * d0  # 60 agents in 3 x 3 x 2 m without a waypoint planner: progress, not arrival


def test_result_file_reproduces_the_first_rows_of_the_reference_log(gpu_required, tmp_path):
    """The reference's committed run (maze10_dense mission, 2-D, M = 10): its result file's rows at t = 0.1 and
    t = 0.2 are the first plan sampled at 0.1 s and the state after the first step.  The harness, started from the same
    states with the same first waypoints (one grid cell ahead), writes the same numbers into the same columns."""
    import fixtures
    k = fixtures.kat()
    c = k["config"]
    cfg = abi.make_config(dim=c["dim"], M=c["M"], n=c["n"], dt=c["dt"], control_input_weight=c["control_input_weight"],
                          terminal_weight=c["terminal_weight"], communication_range=c["communication_range"],
                          world_min=tuple(c["world_min"]), world_max=tuple(c["world_max"]), world_z_2d=c["world_z_2d"],
                          world_use_octomap=False)
    start = np.array([a["start"] for a in k["agents"]], np.float32)
    goal = start.copy()
    goal[:, 0] = np.where(start[:, 0] < 2.0, 5.0, -1.0)  # the two columns of agents swap sides
    path = tmp_path / "result.csv"
    sim = ClosedLoop(cfg, start, goal, radius=c["radius"], downwash=2.0, max_vel=c["max_vel"], max_acc=c["max_acc"],
                     nominal_velocity=c["nominal_velocity"], waypoint_step=c["grid_resolution"], result_csv=str(path))
    try:
        for _ in range(2):
            status, _ = sim.step()
            assert int((status != abi.OK).sum()) == 0
    finally:
        sim.close()
    lines = open(path).read().strip().split("\n")
    assert lines[0].count("planning_time") == len(start)
    rows = [np.array(l.split(","), float).reshape(len(start), 12) for l in lines[1:]]
    assert [round(r[0, 1], 6) for r in rows] == [0.0, 0.1, 0.2, 0.3]
    for a in k["agents"]:
        q = a["id"]
        for row, key in ((rows[1], "t0.1"), (rows[2], "t0.2")):
            exp = a[key]
            assert row[q, 0] == q
            # both files print 6 significant digits
            assert np.allclose(row[q, 2:5], exp["pos"], rtol=3e-6, atol=2e-6), (q, key, row[q, 2:5], exp["pos"])
            assert np.allclose(row[q, 5:8], exp["vel"], rtol=3e-5, atol=2e-6), (q, key, row[q, 5:8], exp["vel"])
            assert np.allclose(row[q, 8:11], exp["acc"], rtol=2e-4, atol=2e-5), (q, key, row[q, 8:11], exp["acc"])


def test_closed_loop_with_dynamic_obstacles(gpu_required):
    """empty10 with two obstacles crossing the arena at constant velocity: rows from the device generator's dynamic
    branch (slack variables), agents keep clear of the obstacles and of each other."""
    cfg, m = _mission("empty10_1")
    cfg.slack_collision_weight = 100.0
    dyn = dict(position=np.float32([[-3.0, 0.3, 1.0], [0.4, 3.0, 1.2]]), velocity=np.float32([[0.3, 0.0, 0.0], [0.0, -0.25, 0.0]]),
               radius=np.array([0.3, 0.25]), max_acc=np.array([0.0, 0.0]), downwash=np.array([1.0, 1.0]), size_prediction=1, horizon=1.0)
    sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"],
                     downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"], dyn=dyn)
    try:
        worst_agents, worst_obs, fails = 1e9, 1e9, 0
        for _ in range(80):
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            worst_agents = min(worst_agents, sim.min_separation())
            d = (sim.pos[:, None, :] - sim.dyn_pos[None, :, :]).norm(dim=2).cpu().numpy()
            worst_obs = min(worst_obs, float((d / (m["radius"] + dyn["radius"])[None, :]).min()))
    finally:
        sim.close()
    print("dynamic obstacles: min agent ratio", worst_agents, "min obstacle ratio", worst_obs, "fail-safes", fails)
    assert worst_agents >= 0.999
    assert worst_obs >= 0.98  # soft constraints (slack variables): clearance up to the slack the QP bought
    assert fails <= 0.05 * 80 * sim.N


def _maze():
    m = json.load(open(os.path.join(HERE, "golden", "world_maze1.json")))
    # launch/simulation.launch: 2-D, M = 10, communication range 3, octomap on
    cfg = abi.make_config(dim=2, M=10, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=3.0,
                          world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]), world_z_2d=1.0)
    world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"])
    return m, cfg, world


def _wall_clearance(pos, boxes):
    b = np.array(boxes)
    d = np.maximum(np.abs(pos[:, None, :2] - b[None, :, :2]) - 0.5 * b[None, :, 3:5], 0)
    return float(np.linalg.norm(d, axis=2).min())


@pytest.mark.parametrize("agents,steps", [([0], 250), ([0, 1, 2, 3, 4], 200)])
def test_maze_mission_with_device_corridors(gpu_required, agents, steps):
    """The reference's default mission (maze10_dense, simulation.launch) with static obstacles: corridor boxes from the
    device generator (row N2), half-spaces (N1), goal LP + QP, all resident in HBM; waypoints from a host-side grid
    search (the reference's planner minus PIBT, so only agents flying in the same direction are used).  Checked: never
    closer to a pillar than the agent radius, never closer to each other than the collision distance, no QP fail-safe,
    the single agent arrives, the group of five crosses the maze."""
    m, cfg, world = _maze()
    start, goal = np.array(m["start"], np.float32)[agents], np.array(m["goal"], np.float32)[agents]
    sim = ClosedLoop(cfg, start, goal, radius=m["radius"], downwash=2.0, max_vel=m["max_vel"], max_acc=m["max_acc"],
                     nominal_velocity=1.0, waypoint_step=m["grid_resolution"], world=world)
    try:
        d0 = float(np.linalg.norm(goal - start, axis=1).mean())
        worst, wall, fails = 1e9, 1e9, 0
        for _ in range(steps):
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            if len(agents) > 1:
                worst = min(worst, sim.min_separation())
            wall = min(wall, _wall_clearance(sim.pos.cpu().numpy(), m["boxes"]))
        dist = (sim.pos - sim.desired_goal).norm(dim=1).cpu().numpy()
    finally:
        sim.close()
    print("maze", agents, "wall clearance", wall, "agent ratio", worst, "fail-safes", fails, "dist", dist, "of", d0)
    assert wall >= m["radius"] - 1e-4
    assert worst >= 0.999
    assert fails == 0
    if len(agents) == 1:
        assert dist[0] < 0.15  # arrived (goal_threshold 0.1 of the launch file + float slack)
    else:
        assert dist.mean() < 0.4 * d0  # through the maze; the rest is the goal-column swap that needs PIBT


def test_full_maze_mission_with_pibt_waypoints(gpu_required, tmp_path):
    """All ten agents of the reference's default mission (two groups crossing the maze head-on through single-lane
    corridors): waypoints from the host-side PIBT stand-in (harness.PibtWaypoints), everything else on the device.  The
    reference's committed run of this mission reports safety_ratio_agent 1.00058 and every agent at its goal
    (log/summary_DLSCGC_10agents.csv); here: every agent arrives, the safety ratio stays above 1, no agent comes closer
    to a pillar than its radius, no QP fail-safe, and the run is written in the reference's result layout."""
    m, cfg, world = _maze()
    world = dict(world, pibt=True)
    path = tmp_path / "result_maze10.csv"
    sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"], downwash=2.0,
                     max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0, waypoint_step=m["grid_resolution"],
                     world=world, result_csv=str(path))
    try:
        worst, wall, fails, done = 1e9, 1e9, 0, None
        for k in range(450):
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            worst = min(worst, sim.min_separation())
            p = sim.pos.cpu().numpy()
            wall = min(wall, _wall_clearance(p, m["boxes"]))
            if done is None and (np.linalg.norm(p - np.array(m["goal"]), axis=1) < 0.1).all():
                done = (k + 1) * cfg.dt
                break
    finally:
        sim.close()
    print("maze10 with PIBT: all agents at their goals after", done, "s; safety ratio", worst, "wall clearance", wall, "fail-safes", fails)
    assert done is not None and done < 90.0
    assert worst >= 1.0 and wall >= m["radius"] - 1e-4 and fails == 0
    lines = open(path).read().strip().split("\n")
    assert lines[0].count("planning_time") == 10 and len(lines) == 1 + 2 * int(round(done / cfg.dt))


def _voxel_clearance(pos, boxes, res):
    """Distance in the plane to the obstacle cells the reference's map holds for `boxes` (MapManager::updateOctreeFromCSV,
    map_manager.cpp:264-308, marks the cells [round(lo / res), round(hi / res)) -- a box that is not aligned with the
    map resolution is shifted by up to half a cell)."""
    b = np.array(boxes)
    lo = np.round((b[:, :2] - 0.5 * b[:, 3:5]) / res) * res
    hi = np.round((b[:, :2] + 0.5 * b[:, 3:5]) / res) * res
    d = np.maximum(np.maximum(lo[None] - pos[:, None, :2], pos[:, None, :2] - hi[None]), 0)
    return float(np.linalg.norm(d, axis=2).min())


def test_forest_mission_with_spinning_obstacles(gpu_required):
    """BASELINE.json configs[3]: the reference's forest10_spin4_100 mission in its own launch shape
    (testall_DLSCGC_3D.launch: 3-D, M = 10, communication range 3, octomap forest, four obstacles circling the arena at the
    agents' top speed with slack variables: 220 variables, 124 reduced dimensions, a working set beyond the LDS of a CU --
    the kernel variant with its working set in HBM).  Corridors, half-spaces (agents + dynamic obstacles), waypoint trap,
    goal LP and QP on the device; waypoints from the reference's own planner restated on the host (row N4: warning nodes,
    obstacles of interest from the QP's collision alerts, escape goals, PIBT).  Checked: at sampled steps the device's plans
    equal the CPU oracle's on the very inputs of that step (verdicts identical, control points <= 1e-4 relative); agents
    never closer to each other than the collision distance, to a tree cell than their radius, or to a spinning obstacle
    than the sum of the radii (the obstacles move at the agents' top speed); fail-safes stay rare; every agent arrives.
    (With round 1's simplified waypoint stand-in the same mission took 65 s and had one obstacle contact.)"""
    import oracle_lib
    from dlsc_gc_planner_amd.harness import SpinObstacles
    m = json.load(open(os.path.join(HERE, "golden", "world_forest1.json")))
    cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], control_input_weight=0.01, terminal_weight=1.0,
                          communication_range=m["communication_range"], slack_collision_weight=m["slack_collision_weight"],
                          world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"], pibt="reference",
                 uncertainty_horizon=m["uncertainty_horizon"])
    spin = SpinObstacles(m["obstacles"])
    sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"],
                     downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                     waypoint_step=m["grid_resolution"], world=world,
                     dyn=spin.dyn(size_prediction=1, horizon=m["uncertainty_horizon"], duration=m["obs_duration"]))
    try:
        assert sim.opt.lds_bytes(4, 2000) <= 160 * 1024  # 124 active rows at most: S (62 KB) + Gram table (29 KB) fit the LDS of a CU
        sim.keep_inputs = True
        worst, wobs, wall, fails, done, checked, worst_err = 1e9, 1e9, 1e9, 0, None, 0, 0.0
        sfc_oracle = oracle_lib.SfcWorld(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], margin=m["radius"],
                                         grid_resolution=m["grid_resolution"], max_vel=m["max_vel"], dt=cfg.dt)
        boxes_checked = 0
        for k in range(500):
            sample = k % 3 == 1 and k < 320
            if sample:  # what the corridor generator is about to read
                before = [t.cpu().numpy().copy() for t in (sim.init_buf, sim.cur_goal, sim.sfc.sfcs)]
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            if sample:
                agents, ro, rows, sfc, D = sim.step_inputs()  # rows as the LP + QP saw them, i.e. after the waypoint trap
                # row N2 in 3-D: the step's corridor boxes against the oracle's restatement, bit for bit
                for a in range(sim.N):
                    _, box = sfc_oracle.construct(before[0][a], before[1][a], agents["next_waypoint"][a], before[2][a])
                    assert np.array_equal(box, sfc[a]), (k, a)
                    boxes_checked += cfg.M
                got = sim.batch.results()
                ref = oracle_lib.plan_batch(cfg, agents, ro, rows, sfc, D, True)
                assert np.array_equal(got["goal_status"], ref["goal_status"]) and np.array_equal(got["status"], ref["status"]), k
                ok = ref["status"] == abi.OK
                if ok.any():
                    err = float(np.abs(got["ctrl"][ok].astype(np.float64) - ref["ctrl"][ok]).max())
                    assert err <= 1e-4 * max(1.0, float(np.abs(ref["ctrl"][ok]).max())), (k, err)
                    worst_err = max(worst_err, err)
                checked += int(ok.sum())
            worst = min(worst, sim.min_separation())
            if sim.sim_time <= m["obs_duration"]:
                wobs = min(wobs, sim.min_obstacle_separation())
            p = sim.pos.cpu().numpy()
            wall = min(wall, _voxel_clearance(p, m["boxes"], m["world_resolution"]))
            if (np.linalg.norm(p - np.array(m["goal"]), axis=1) < 0.1).all():
                done = (k + 1) * cfg.dt
                break
    finally:
        sim.close()
        sfc_oracle.close()
    print("forest10_spin4_100 (3-D, M = 10): all agents at their goals after", done, "s; agent safety ratio", worst,
          "tree clearance", wall, "obstacle safety ratio", wobs, "fail-safes", fails, "| plans checked against the oracle:",
          checked, "worst control-point difference", worst_err, "| corridor boxes equal to the oracle's:", boxes_checked)
    assert checked >= 100 and boxes_checked >= 2000
    assert worst >= 1.0 and wall >= m["radius"] - 1e-4
    assert wobs >= 1.0  # no contact with the spinning obstacles
    assert fails <= 5
    assert done is not None and done <= 30.0


@pytest.mark.parametrize("with_world", [False, True, "reference", "forest", "crowd"])
def test_hip_graph_step_equals_eager_step(gpu_required, with_world):
    """ClosedLoop.enable_graph(): the steady-state step captured once in a HIP graph and replayed gives the same plans,
    states and verdicts, bit for bit, as the eager sequence of launches -- in an empty world (waypoints inside the graph),
    in the maze (corridor generator inside the graph, waypoints from the host outside), in the maze as the reference's
    mission runs it (communication-range neighbour lists rebuilt on the device every step: the one-launch search) and in the
    forest (dynamic obstacles, waypoint trap)."""
    import torch
    if with_world == "crowd":
        # BASELINE.json configs[4] as bench.py flies it: 2 048 agents, neighbour lists through cells (a memset-free search: round 4's
        # captured step faulted on its second replay), presolve, launch order from class masks (three graphs, one per phase)
        from dlsc_gc_planner_amd import workloads
        cfg = workloads.x2048_config(device=0)
        start, goal = workloads.swapping_pairs(cfg, 2048)
        kw = dict(radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5, device=0, neighbours="device", nbr_pitch=256,
                  corridor_margin=0.5, prune=True, order_hint=True)
    elif with_world == "forest":
        from dlsc_gc_planner_amd.harness import SpinObstacles
        m = json.load(open(os.path.join(HERE, "golden", "world_forest1.json")))
        cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], control_input_weight=0.01, terminal_weight=1.0,
                              communication_range=m["communication_range"], slack_collision_weight=m["slack_collision_weight"],
                              world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
        world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"], pibt="reference",
                     uncertainty_horizon=m["uncertainty_horizon"])
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=world)
        kw_dyn = lambda: SpinObstacles(m["obstacles"]).dyn(size_prediction=1, horizon=m["uncertainty_horizon"], duration=m["obs_duration"])  # noqa: E731
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    elif with_world:
        m, cfg, world = _maze()
        kw = dict(radius=m["radius"], downwash=2.0, max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=dict(world, pibt=True if with_world is True else "reference"))
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    else:
        cfg, m = _mission("empty10_1")
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"],
                  nominal_velocity=m["nominal_velocity"])
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    runs = []
    for use_graph in (False, True):
        sim = ClosedLoop(cfg, start, goal, **(dict(kw, dyn=kw_dyn()) if with_world == "forest" else kw))
        try:
            hist = []
            for k in range(40 if with_world != "crowd" else 14):
                if use_graph and k == 3:
                    sim.enable_graph()
                status, iters = sim.step()
                hist.append((sim.plans.cpu().numpy().copy(), sim.pos.cpu().numpy().copy(), sim.cur_goal.cpu().numpy().copy(),
                             status.cpu().numpy().copy()))
            assert (sim.graph is not None) == use_graph
        finally:
            sim.close()
        runs.append(hist)
    for k, (e, g) in enumerate(zip(*runs)):
        for x, y in zip(e, g):
            assert np.array_equal(x, y), k
    moved = np.linalg.norm(runs[1][-1][1] - start, axis=1)
    assert moved.mean() > (1.0 if with_world != "crowd" else 0.5)  # the agents did fly (the crowd: 14 steps)


@pytest.mark.parametrize("with_world", [False, "reference", "forest"])
def test_fused_step_equals_the_chain_of_launches(gpu_required, with_world, monkeypatch):
    """dlscqp_fused_step_device (round 6): the steady-state step of a small mission as ONE kernel -- neighbour lists, corridor boxes,
    half-spaces, waypoint trap, goal LP + QP, state update as stages of a workgroup per agent, one grid-wide barrier in front of the
    epilogue -- gives the same plans, states, sub-goals and verdicts, bit for bit, as the four or five launches it replaces: the
    empty world, the maze as the reference's mission runs it (lists rebuilt every step, corridors), the forest (dynamic obstacles,
    slack variables, waypoint trap); eager and replayed as a HIP graph."""
    import torch
    if with_world == "forest":
        from dlsc_gc_planner_amd.harness import SpinObstacles
        m = json.load(open(os.path.join(HERE, "golden", "world_forest1.json")))
        cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], control_input_weight=0.01, terminal_weight=1.0,
                              communication_range=m["communication_range"], slack_collision_weight=m["slack_collision_weight"],
                              world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
        world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"], pibt="reference",
                     uncertainty_horizon=m["uncertainty_horizon"])
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=world)
        kw_dyn = lambda: SpinObstacles(m["obstacles"]).dyn(size_prediction=1, horizon=m["uncertainty_horizon"], duration=m["obs_duration"])  # noqa: E731
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    elif with_world:
        m, cfg, world = _maze()
        kw = dict(radius=m["radius"], downwash=2.0, max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=dict(world, pibt="reference"))
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    else:
        cfg, m = _mission("empty10_1")
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"],
                  nominal_velocity=m["nominal_velocity"])
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    runs = []
    for fused, use_graph in (("0", False), ("1", False), ("1", True)):
        monkeypatch.setenv("DLSCQP_FUSED_STEP", fused)
        sim = ClosedLoop(cfg, start, goal, **(dict(kw, dyn=kw_dyn()) if with_world == "forest" else kw))
        try:
            hist = []
            for k in range(40):
                if use_graph and k == 3:
                    sim.enable_graph()
                status, iters = sim.step()
                hist.append((sim.plans.cpu().numpy().copy(), sim.pos.cpu().numpy().copy(), sim.cur_goal.cpu().numpy().copy(),
                             status.cpu().numpy().copy()))
            # every steady-state step was ONE launch (or none was)
            assert sim.fused_steps_done == (0 if fused == "0" else (39 if not use_graph else sim.fused_steps_done)) and (fused == "0" or sim.fused_steps_done >= 3)
            assert sim.fused_step == (fused == "1")
        finally:
            sim.close()
        runs.append(hist)
    for other in runs[1:]:
        for k, (e, g) in enumerate(zip(runs[0], other)):
            for x, y in zip(e, g):
                assert np.array_equal(x, y), k
    assert np.linalg.norm(runs[1][-1][1] - start, axis=1).mean() > 1.0  # the agents did fly


@pytest.mark.parametrize("mission", ["empty", "maze", "forest", "crowd"])
def test_bookkeeping_in_the_plan_kernels_epilogue_equals_the_advance_launch(gpu_required, mission, monkeypatch):
    """dlscqp_device_batch.advance_after: the wavefront that planned an agent also applies dlscqp_advance to it (fail-safe, sub-goal,
    state at dt, next records, initial trajectory, boxes, the planner's inputs in pinned memory) -- the same flight, bit for bit, as
    with dlscqp_advance_device in a launch of its own (DLSCQP_FUSED_ADVANCE=0), eager and as a HIP graph."""
    kw_dyn = None
    if mission == "crowd":
        from dlsc_gc_planner_amd import workloads
        cfg = workloads.x2048_config(device=0)
        start, goal = workloads.swapping_pairs(cfg, 2048)
        kw = dict(radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5, device=0, neighbours="device", nbr_pitch=256,
                  corridor_margin=0.5, prune=True, order_hint=True)
    elif mission == "forest":
        from dlsc_gc_planner_amd.harness import SpinObstacles
        m = json.load(open(os.path.join(HERE, "golden", "world_forest1.json")))
        cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], control_input_weight=0.01, terminal_weight=1.0,
                              communication_range=m["communication_range"], slack_collision_weight=m["slack_collision_weight"],
                              world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
        world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"], pibt="reference",
                     uncertainty_horizon=m["uncertainty_horizon"])
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=world)
        kw_dyn = lambda: SpinObstacles(m["obstacles"]).dyn(size_prediction=1, horizon=m["uncertainty_horizon"], duration=m["obs_duration"])  # noqa: E731
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    elif mission == "maze":
        m, cfg, world = _maze()
        kw = dict(radius=m["radius"], downwash=2.0, max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=dict(world, pibt="reference"))
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    else:
        cfg, m = _mission("empty60_1")
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"])
        start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    steps = 14 if mission == "crowd" else 80
    runs = []
    for mode in ("0", "1", "graph"):
        monkeypatch.setenv("DLSCQP_FUSED_ADVANCE", "0" if mode == "0" else "1")
        sim = ClosedLoop(cfg, start, goal, **(dict(kw, dyn=kw_dyn()) if kw_dyn else kw))
        try:
            assert sim.fused_advance == (mode != "0")
            hist = []
            for k in range(steps):
                if mode == "graph" and k == 3:
                    sim.enable_graph()
                status, iters = sim.step()
                hist.append((sim.plans.cpu().numpy().copy(), sim.batch.t["agents"].cpu().numpy().copy(), sim.cur_goal.cpu().numpy().copy(),
                             sim.init_buf.cpu().numpy().copy(), status.cpu().numpy().copy()))
        finally:
            sim.close()
        runs.append(hist)
    for other in runs[1:]:
        for k, (e, g) in enumerate(zip(runs[0], other)):
            for x, y in zip(e, g):
                assert np.array_equal(x, y), k


@pytest.mark.parametrize("mission", ["maze", "forest"])
def test_planner_inputs_written_by_the_advance_kernel_equal_the_packed_copy(gpu_required, mission, monkeypatch):
    """dlscqp_advance.out_host_pack: positions, sub-goals, plans and collision alerts for the host-side waypoint planner, written
    into pinned host memory by the step's last kernel, give the same flight, bit for bit, as packing them on the device and copying
    them (DLSCQP_HOST_PACK=0: what rounds 3-5 did) -- the maze as the reference's mission runs it, and the forest with its spinning
    obstacles (obstacle states mirrored on the host, alerts from the slack costs), also replayed as a HIP graph."""
    if mission == "forest":
        from dlsc_gc_planner_amd.harness import SpinObstacles
        m = json.load(open(os.path.join(HERE, "golden", "world_forest1.json")))
        cfg = abi.make_config(dim=m["dim"], M=m["M"], dt=m["dt"], control_input_weight=0.01, terminal_weight=1.0,
                              communication_range=m["communication_range"], slack_collision_weight=m["slack_collision_weight"],
                              world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
        world = dict(boxes=m["boxes"], resolution=m["world_resolution"], grid_resolution=m["grid_resolution"], pibt="reference",
                     uncertainty_horizon=m["uncertainty_horizon"])
        kw = dict(radius=m["radius"], downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=world)
        kw_dyn = lambda: SpinObstacles(m["obstacles"]).dyn(size_prediction=1, horizon=m["uncertainty_horizon"], duration=m["obs_duration"])  # noqa: E731
    else:
        m, cfg, world = _maze()
        kw = dict(radius=m["radius"], downwash=2.0, max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=1.0,
                  waypoint_step=m["grid_resolution"], world=dict(world, pibt="reference"))
    start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    runs = []
    for mode in ("0", "1", "graph"):
        monkeypatch.setenv("DLSCQP_HOST_PACK", "0" if mode == "0" else "1")
        sim = ClosedLoop(cfg, start, goal, **(dict(kw, dyn=kw_dyn()) if mission == "forest" else kw))
        try:
            assert sim.pack_in_advance == (mode != "0")
            hist = []
            for k in range(160 if mission == "forest" else 60):
                if mode == "graph" and k == 3:
                    sim.enable_graph()
                status, iters = sim.step()
                hist.append((sim.plans.cpu().numpy().copy(), sim.pos.cpu().numpy().copy(), sim.cur_goal.cpu().numpy().copy(),
                             sim.wp.cpu().numpy().copy(), status.cpu().numpy().copy()))
                if mode != "0":  # what the kernel wrote for the planner is this step's state
                    N, E = sim.N, cfg.M * (cfg.n + 1) * 3
                    h = sim._adv_pack.numpy()
                    assert np.array_equal(h[:3 * N].reshape(N, 3), hist[-1][1]) and np.array_equal(h[3 * N:6 * N].reshape(N, 3), hist[-1][2])
                    assert np.array_equal(h[6 * N:6 * N + N * E].reshape(hist[-1][0].shape), hist[-1][0])
                    if sim.D:
                        assert np.array_equal(h[6 * N + N * E:].reshape(N, sim.D) > 0.5, sim.batch.out_slack.cpu().numpy()[:, :sim.D] > sim.slack_threshold)
        finally:
            sim.close()
        runs.append(hist)
    for other in runs[1:]:
        for k, (e, g) in enumerate(zip(runs[0], other)):
            for x, y in zip(e, g):
                assert np.array_equal(x, y), k
    assert np.linalg.norm(runs[1][-1][1] - start, axis=1).mean() > 1.0  # the agents did fly


@pytest.mark.parametrize("M", [2, 5, 10])
def test_advance_kernel_equals_its_numpy_restatement(gpu_required, M):
    """dlscqp_advance_device (fail-safe, sub-goal, state := plan(dt), next records, initial trajectory, straight-line
    waypoint) against the same arithmetic in numpy float32: bit-exact (the waypoint to float32 rounding of a division)."""
    import torch
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    rng = np.random.default_rng(7 + M)
    cfg = abi.make_config(dim=3, M=M, dt=0.2)
    n, N = cfg.n, 37
    opt = BatchTrajOptimizer(cfg)
    try:
        agents = np.zeros(N, abi.AGENT_DTYPE)
        agents["id"] = np.arange(N)
        agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = 1.0, 2.0, 0.15, 1.0
        db = opt.make_device_batch(agents, np.zeros(N + 1, np.int64), np.zeros(0, abi.ROW_DTYPE), None, 0)
        f = lambda *shape: rng.uniform(-3, 3, shape).astype(np.float32)  # noqa: E731
        prev, ctrl, goal, cur, want = f(N, M, n + 1, 3), f(N, M, n + 1, 3), f(N, 3), f(N, 3), f(N, 3)
        status = rng.choice([abi.OK, abi.INFEASIBLE], N, p=[0.7, 0.3]).astype(np.int32)
        # desired goal nearer than one waypoint step to where the agent will be (segment 1 of the new or of the old plan)
        want[:5] = np.where((status[:5] == abi.OK)[:, None], ctrl[:5, 1, 0], prev[:5, min(2, M - 1), 0 if M > 2 else n]) + np.float32(0.01)
        gstat = rng.choice([abi.OK, abi.INFEASIBLE, abi.GOAL_FALLBACK], N).astype(np.int32)
        up = lambda x: torch.from_numpy(x.copy()).cuda()  # noqa: E731
        db.out_status.copy_(up(status)); db.out_goal_status.copy_(up(gstat)); db.out_ctrl.copy_(up(ctrl)); db.out_goal.copy_(up(goal))
        plans, cg, init, dg = up(prev), up(cur), torch.zeros((N, M, n + 1, 3), device="cuda"), up(want)
        opt.make_advance(db, plans, cg, init, desired_goal=dg, waypoint_step=0.5)()
        torch.cuda.synchronize()
        rec = db.t["agents"].cpu().numpy().view(abi.AGENT_DTYPE).reshape(-1)
        # ---- restatement ----
        shift = lambda p: np.concatenate([p[:, 1:], np.repeat(p[:, -1:, -1:], n + 1, axis=2)], axis=1)  # noqa: E731
        ok = status == abi.OK
        new = np.where(ok[:, None, None, None], ctrl, shift(prev))
        assert np.array_equal(plans.cpu().numpy(), new)
        assert np.array_equal(init.cpu().numpy(), shift(new))
        cg_new = np.where((gstat != abi.INFEASIBLE)[:, None], goal, cur)
        assert np.array_equal(cg.cpu().numpy(), cg_new) and np.array_equal(rec["current_goal_point"], cg_new)
        s1, s2 = np.float32(n / cfg.dt), np.float32((n - 1) / cfg.dt)
        c = new[:, 1, 0:3]
        v0, v1 = (c[:, 1] - c[:, 0]) * s1, (c[:, 2] - c[:, 1]) * s1
        pos = c[:, 0]
        assert np.array_equal(rec["position"], pos)
        assert np.array_equal(rec["velocity"], v0)
        assert np.array_equal(rec["acceleration"], (v1 - v0) * s2)
        d = want - pos
        dist = np.sqrt((d * d).sum(1, dtype=np.float32).astype(np.float64))
        wp = np.where((dist <= 0.5)[:, None], want, pos + d / np.maximum(dist, 1e-9).astype(np.float32)[:, None] * np.float32(0.5))
        assert (dist[:5] <= 0.5).all() and (dist[5:] > 0.5).any()
        assert np.allclose(rec["next_waypoint"], wp, rtol=0, atol=1e-6)
        assert np.array_equal(rec["id"], np.arange(N)) and np.all(rec["max_acc"] == 2.0)  # the rest of the record is untouched
    finally:
        opt.close()
