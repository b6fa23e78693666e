"""CPU tier, row N2 (first step: the oracle's restatement only).  Safe-flight-corridor boxes in the reference's own
maze world (world/maze_icra2023/dense/maze1.csv, numbers in tests/golden/world_maze1.json).  The reference has no
tests or golden vectors for this row and its octomap / DynamicEDT3D dependencies are absent (parity unpinned, see
oracle/dlsc_sfc_oracle.c), so these are the properties its own code relies on: a corridor box contains the points
it was grown from, keeps the agent radius away from every occupied cell, stays inside the world, and cannot grow
further."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
RES, R_AGENT = 0.1, 0.15


@pytest.fixture(scope="module")
def maze():
    m = json.load(open(os.path.join(HERE, "golden", "world_maze1.json")))
    w = O.SfcWorld(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], margin=m["radius"],
                   grid_resolution=m["grid_resolution"], max_vel=m["max_vel"], dt=0.2)
    occ = np.zeros((0, 3))
    cells = []
    for c in m["boxes"]:
        lo = [_c_round((c[k] - 0.5 * c[3 + k]) / RES) for k in range(3)]
        hi = [_c_round((c[k] + 0.5 * c[3 + k]) / RES) for k in range(3)]
        g = np.stack(np.meshgrid(*[np.arange(lo[k], hi[k]) for k in range(3)], indexing="ij"), -1).reshape(-1, 3)
        cells.append(g)
    occ = np.unique(np.concatenate(cells), axis=0)
    yield m, w, occ
    w.close()


def _c_round(v):
    """C round(): halves away from zero (map_manager.cpp uses it to pick the cells of a box)."""
    import math
    return int(math.copysign(math.floor(abs(v) + 0.5), v))


def _clearance(box, occ):
    """Smallest L-infinity distance between the box and an occupied cell cube (independent of the oracle)."""
    lo, hi = occ * RES, (occ + 1) * RES
    gap = np.maximum(np.maximum(lo - box[3:], box[:3] - hi), 0.0)
    return gap.max(axis=1).min()


def test_world_grid_matches_the_box_list(maze):
    m, w, occ = maze
    assert len(occ) > 20000  # 50 pillars of 0.5 x 0.5 x 2.5 m, many of them adjacent
    # a cell inside a pillar is blocked, the middle of a free corridor is not (agent radius margin)
    assert w.obstacle_in_box([0.0, 0.0, 1.0, 0.0, 0.0, 1.0])
    assert not w.obstacle_in_box([-1.0, 3.0, 1.0, -1.0, 3.0, 1.0])


def test_boxes_grown_from_the_mission_starts(maze):
    m, w, occ = maze
    wmin, wmax = np.array(m["world"][:3]), np.array(m["world"][3:])
    for p in m["start"] + m["goal"]:
        ok, sfcs = w.initialize(p, 10)
        assert ok
        box = sfcs[0].astype(np.float64)
        assert (sfcs == sfcs[0]).all()
        assert (box[:3] <= np.array(p) + 1e-6).all() and (box[3:] >= np.array(p) - 1e-6).all()
        assert (box[:3] >= wmin - 1e-5).all() and (box[3:] <= wmax + 1e-5).all()
        # margin: after the compensation step the box keeps floor(margin / res) * res = 0.1 m to every occupied cube
        assert _clearance(box, occ) >= 0.1 - 1e-5
        # z is free in this world: the box reaches the floor and stops at the iteration cap above the agent
        # (11 slabs of 0.1 m + the margin compensation, collision_constraints.cpp:1031, 1077-1086)
        assert box[2] <= wmin[2] + 1e-5 and abs(box[5] - (p[2] + 1.1 + 0.05)) < 1e-5


def test_expansion_is_maximal_up_to_the_iteration_cap(maze):
    m, w, occ = maze
    rng = np.random.default_rng(0)
    cap = int(round(max(2 * m["grid_resolution"], m["max_vel"] * 0.2) / RES)) + 1
    tried = 0
    for _ in range(200):
        p = rng.uniform([-1.5, 0.0, 0.3], [5.5, 4.0, 2.0])
        init = np.concatenate([np.floor(p / RES) * RES, np.ceil(p / RES) * RES]).astype(np.float32)
        ok, box = w.expand(init)
        if not ok:
            assert w.obstacle_in_box(init)
            continue
        tried += 1
        delta = R_AGENT - int(R_AGENT / RES) * RES
        core = box.astype(np.float64).copy()  # undo the margin compensation
        for k in range(3):
            if core[k] > m["world"][k] + 1e-5 + delta:
                core[k] += delta
            if core[3 + k] < m["world"][3 + k] - 1e-5 - delta:
                core[3 + k] -= delta
        assert not w.obstacle_in_box(core)
        for axis in range(6):  # one more slab in any direction is blocked, outside the world, or beyond the cap
            k = axis % 3
            grown = core.copy()
            slab = core.copy()
            if axis < 3:
                grown[k] -= RES
                slab[3 + k], slab[k] = core[k], grown[k]
                steps = int(round((init[k] - core[k]) / RES))
                outside = grown[k] < m["world"][k] - 1e-5
            else:
                grown[3 + k] += RES
                slab[k], slab[3 + k] = core[3 + k], grown[3 + k]
                steps = int(round((core[3 + k] - init[3 + k]) / RES))
                outside = grown[3 + k] > m["world"][3 + k] + 1e-5
            assert outside or steps >= cap or w.obstacle_in_box(slab) or w.obstacle_in_box(grown), (p, axis, core)
    assert tried > 60  # the rest of the random points lie inside or too close to a pillar


def test_corridor_follows_a_flight_through_the_maze(maze):
    """constructSFCFromInitialTraj over a straight flight along a free corridor: boxes shift with the plan, the last box
    contains the plan's end point, the sub-goal and (greedy case) the waypoint."""
    m, w, occ = maze
    M, n, dt = 10, 5, 0.2
    pos = np.array([-1.0, 0.5, 1.0], np.float32)  # corridor y = 0.5 between the pillar rows at y = 0 and y = 1
    ok, sfcs = w.initialize(pos, M)
    assert ok
    codes = []
    for step in range(15):
        end = pos + np.float32([0.1 * (step + 1), 0, 0])
        traj = np.linspace(pos + np.float32([0.1 * step, 0, 0]), end, M * (n + 1)).reshape(M, n + 1, 3).astype(np.float32)
        goal = end + np.float32([0.2, 0, 0])
        wp = end + np.float32([0.5, 0, 0])
        code, sfcs = w.construct(traj, goal, wp, sfcs)
        codes.append(code)
        last = sfcs[-1].astype(np.float64)
        for q in (traj[-1, -1], goal) + ((wp,) if code == 0 else ()):
            assert (last[:3] <= q + 1e-5).all() and (last[3:] >= q - 1e-5).all()
        for b in sfcs:
            assert _clearance(b.astype(np.float64), occ) >= 0.1 - 1e-5
    assert 0 in codes


# ---- the device source (sfc_core.hpp + sfc_world.cpp, host build) against the oracle ------------------------------
def _flights(m, n_agents=10, steps=8, M=10, n=5, seed=3):
    """Agents starting at the mission's start cells and creeping towards a neighbouring free cell: per step the
    shifted plan, sub-goal and waypoint that constructSFCFromInitialTraj sees."""
    rng = np.random.default_rng(seed)
    start = np.array(m["start"], np.float32)[:n_agents]
    goal = np.array(m["goal"], np.float32)[:n_agents]
    frames = []
    pos = start.copy()
    for s in range(steps):
        dirn = goal - pos
        dirn[:, 2] = 0
        dirn /= np.maximum(np.linalg.norm(dirn, axis=1, keepdims=True), 1e-6)
        nxt = (pos + 0.12 * dirn + rng.normal(0, 0.01, pos.shape)).astype(np.float32)
        traj = np.ascontiguousarray(np.linspace(pos, nxt, M * (n + 1)).transpose(1, 0, 2).reshape(n_agents, M, n + 1, 3), dtype=np.float32)
        ahead = np.where(np.arange(n_agents) % 2 == 0, 0.2, 0.5)[:, None]  # near and far waypoints: greedy and fallback boxes
        frames.append((traj, (nxt + 0.15 * dirn).astype(np.float32), (nxt + ahead * dirn).astype(np.float32)))
        pos = nxt
    return start, frames


@pytest.mark.parametrize("variant", ["turns", "in_sequence", "records"])
def test_device_source_matches_oracle_bit_for_bit(maze, variant, monkeypatch):
    """variant: how sfc_core.hpp gets through the trial expansions -- "turns": 64 trials per pass with the summed-volume table
    (the default); "in_sequence": one trial after the other, every box point by point through the lattice-point table
    (worlds without the table, or a limit of moves beyond the chain); "records": the same on the 32-byte records of round 3."""
    if variant == "in_sequence":
        monkeypatch.setenv("EMUL_SFC_SVT", "0")
    if variant == "records":
        monkeypatch.setenv("EMUL_SFC_RECORDS", "1")
    import emul_lib as E
    from dlsc_gc_planner_amd import abi
    m, w, occ = maze
    cfg = abi.make_config(dim=2, M=10, dt=0.2, world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    world = dict(boxes=m["boxes"], resolution=m["world_resolution"], world_min=m["world"][:3], world_max=m["world"][3:])
    start, frames = _flights(m)
    N = len(start)
    radius, vmax = np.full(N, m["radius"]), np.full(N, m["max_vel"])
    dev, codes = E.generate_sfc(cfg, world, radius, vmax, m["grid_resolution"], position=start)
    ref = np.stack([w.initialize(p, cfg.M)[1] for p in start])
    assert (codes == 0).all() and np.array_equal(dev, ref)
    seen = set()
    for traj, goals, wps in frames:
        dev, codes = E.generate_sfc(cfg, world, radius, vmax, m["grid_resolution"], sfcs=dev, initial_traj=traj, goals=goals, waypoints=wps)
        for a in range(N):
            code, ref[a] = w.construct(traj[a], goals[a], wps[a], ref[a])
            assert code == codes[a], (a, code, codes[a])
            seen.add(code)
        assert np.array_equal(dev, ref)
    assert 0 in seen and len(seen) >= 2  # greedy boxes and at least one fallback
    # a blocked start cell is reported, not thrown
    _, codes = E.generate_sfc(cfg, world, radius[:1], vmax[:1], m["grid_resolution"], position=np.float32([[0.0, 0.0, 1.0]]))
    assert codes[0] == -1


CLASSES = [(0.15, 1.0), (0.26, 1.0), (0.15, 25.0), (0.15, 40.0)]  # (radius, max_vel)


def mixed_classes_check(m, generate):
    """Agents of several radius / speed classes in one launch against one oracle world per class.  Radius 0.26 asks for another
    summed-volume table than 0.15 (sfc_gap_max 2 and 1); max_vel 25 lifts the limit of moves per direction to 51 (one pass of 64
    trials no longer covers a round of six directions), 40 to 81 -- beyond the chain of face positions, so that agent grows its
    boxes one trial after the other.  generate(radius, vmax, **kw) -> (sfcs, codes) as emul_lib.generate_sfc."""
    start, frames = _flights(m, n_agents=8, steps=4)
    N = len(start)
    cls = [CLASSES[a % len(CLASSES)] for a in range(N)]
    radius, vmax = np.array([c[0] for c in cls]), np.array([c[1] for c in cls])
    worlds = {c: O.SfcWorld(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], margin=c[0],
                            grid_resolution=m["grid_resolution"], max_vel=c[1], dt=0.2) for c in CLASSES}
    try:
        dev, codes = generate(radius, vmax, position=start)
        ref = np.zeros_like(dev)
        for a in range(N):
            ok, ref[a] = worlds[cls[a]].initialize(start[a], dev.shape[1])
            assert (codes[a] == 0) == bool(ok)
        live = codes == 0
        assert live.sum() >= N - 2 and np.array_equal(dev[live], ref[live])
        for traj, goals, wps in frames:
            dev, codes = generate(radius, vmax, sfcs=dev, initial_traj=traj, goals=goals, waypoints=wps)
            for a in np.nonzero(live)[0]:
                code, ref[a] = worlds[cls[a]].construct(traj[a], goals[a], wps[a], ref[a])
                assert code == codes[a], (a, code, codes[a])
            assert np.array_equal(dev[live], ref[live])
    finally:
        for w in worlds.values():
            w.close()


def test_device_source_mixed_radius_and_speed_classes(maze):
    import emul_lib as E
    from dlsc_gc_planner_amd import abi
    m = maze[0]
    cfg = abi.make_config(dim=2, M=10, dt=0.2, world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    world = dict(boxes=m["boxes"], resolution=m["world_resolution"], world_min=m["world"][:3], world_max=m["world"][3:])
    mixed_classes_check(m, lambda radius, vmax, **kw: E.generate_sfc(cfg, world, radius, vmax, m["grid_resolution"], **kw))


@pytest.mark.parametrize("variant", ["turns", "in_sequence"])
def test_device_source_open_space(maze, variant, monkeypatch):
    """No obstacle within reach: every trial is accepted until the limit of moves or the boundary -- rounds of more than 64
    trials (a pass that ends without a lane that fails, then continues), 6 x 51 moves.  in_sequence: the same boxes grown one
    trial after the other, point by point -- boxes of more than 64 lattice points along an axis (beyond the per-axis tables)."""
    if variant == "in_sequence":
        monkeypatch.setenv("EMUL_SFC_SVT", "0")
    import emul_lib as E
    from dlsc_gc_planner_amd import abi
    m = dict(maze[0])
    m["boxes"] = [[-1.8, -0.1, 0.25, 0.2, 0.2, 0.5]]  # one small box in a corner of the world
    m["world"] = [-2.0, -0.3, 0.0, 6.0, 4.3, 6.0]
    cfg = abi.make_config(dim=3, M=5, dt=0.2, world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    world = dict(boxes=m["boxes"], resolution=m["world_resolution"], world_min=m["world"][:3], world_max=m["world"][3:])
    pos = np.float32([[2.0, 2.0, 3.0], [0.5, 1.0, 1.0], [5.5, 4.0, 5.5]])
    for vmax in (1.0, 25.0):
        w = O.SfcWorld(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], margin=m["radius"],
                       grid_resolution=m["grid_resolution"], max_vel=vmax, dt=0.2)
        try:
            dev, codes = E.generate_sfc(cfg, world, np.full(3, m["radius"]), np.full(3, vmax), m["grid_resolution"], position=pos)
            ref = np.stack([w.initialize(p, cfg.M)[1] for p in pos])
            assert (codes == 0).all() and np.array_equal(dev, ref)
            assert vmax == 1.0 or (ref[0, 0, 3:] - ref[0, 0, :3]).min() > 3.0  # grown far beyond one pass of trials
        finally:
            w.close()
