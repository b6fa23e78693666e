"""Synthetic agent-QP workloads of BASELINE.json / SURVEY.md section 8(d).

Host-side numpy only.  A workload is produced by a short closed-loop run of the planner
pipeline that surrounds the hot path in the reference (previous-solution shift,
DLSC-GC half-space generation, goal LP, trajectory QP): the `solver` callback that plans a
batch is supplied by the caller -- bench.py passes the HIP library, the fixture generator in
tests/golden passes the CPU oracle.  This module itself never touches either.

The half-space construction follows TrajPlanner::generateDLSCGC
(/root/reference/src/traj_planner.cpp:603-666) closely enough to give the QPs the reference's
structure (27 rows per neighbour, identical last-segment rows, downwash-scaled normals); it is
a workload generator, not a parity-checked restatement of that function.
"""
import itertools

import numpy as np

from . import abi

DOWNWASH = 2.0  # every shipped mission: "downwash": 2.0


def s64_config(device=0):
    """Config 2 of BASELINE.json: 3-D, 5 segments x degree 5, no octomap, no comm range."""
    return abi.make_config(dim=3, M=5, n=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0,
                           communication_range=-1.0, world_use_octomap=False,
                           world_min=(-5.0, -5.0, 0.0), world_max=(5.0, 5.0, 2.5), device=device)


def x2048_config(device=0):
    """Config 5 of BASELINE.json: 2-D, 10 segments, SFC boxes, communication range 3."""
    return abi.make_config(dim=2, M=10, n=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0,
                           communication_range=3.0, world_use_octomap=True,
                           world_min=(-20.0, -20.0, 0.0), world_max=(20.0, 20.0, 2.5), device=device)


# ---------------------------------------------------------------------------------------------
_EDGES = np.array(list(itertools.combinations(range(6), 2)))
_TRIS = np.array(list(itertools.combinations(range(6), 3)))


def hull_closest_point(pts):
    """Closest point to the origin of conv(pts[b, 0..5, :]) for a batch; exact enumeration of
    vertices, edges and triangles (the relative control-point hull has 6 points)."""
    B = pts.shape[0]
    best = pts[:, 0, :].copy()
    best_d = np.einsum("bk,bk->b", best, best)

    def consider(c):
        nonlocal best, best_d
        d = np.einsum("bk,bk->b", c, c)
        better = d < best_d
        best = np.where(better[:, None], c, best)
        best_d = np.where(better, d, best_d)

    for v in range(1, 6):
        consider(pts[:, v, :])
    a = pts[:, _EDGES[:, 0], :]
    b = pts[:, _EDGES[:, 1], :]
    ab = b - a
    den = np.einsum("bek,bek->be", ab, ab)
    t = np.where(den > 0, -np.einsum("bek,bek->be", a, ab) / np.where(den > 0, den, 1), 0.0)
    t = np.clip(t, 0.0, 1.0)
    ce = a + t[..., None] * ab
    for e in range(ce.shape[1]):
        consider(ce[:, e, :])
    a = pts[:, _TRIS[:, 0], :]
    b = pts[:, _TRIS[:, 1], :]
    c = pts[:, _TRIS[:, 2], :]
    ab, ac = b - a, c - a
    # solve [ab.ab ab.ac; ab.ac ac.ac][u v]' = -[a.ab a.ac]'
    g11 = np.einsum("btk,btk->bt", ab, ab)
    g12 = np.einsum("btk,btk->bt", ab, ac)
    g22 = np.einsum("btk,btk->bt", ac, ac)
    r1 = -np.einsum("btk,btk->bt", a, ab)
    r2 = -np.einsum("btk,btk->bt", a, ac)
    det = g11 * g22 - g12 * g12
    ok = det > 1e-14 * np.maximum(g11 * g22, 1e-300)
    dets = np.where(ok, det, 1.0)
    u = (r1 * g22 - r2 * g12) / dets
    v = (g11 * r2 - g12 * r1) / dets
    inside = ok & (u >= 0) & (v >= 0) & (u + v <= 1)
    ct = a + u[..., None] * ab + v[..., None] * ac
    far = np.full((B, 3), 1e9)
    for tix in range(ct.shape[1]):
        consider(np.where(inside[:, tix, None], ct[:, tix, :], far))
    return best


def segment_closest_points(p1, q1, p2, q2):
    """Closest points between segments p1->q1 and p2->q2 (batch), after Ericson's
    ClosestPtSegmentSegment; plays the role of closestPointsBetweenLineSegments
    (geometry.hpp:184-274)."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a = np.einsum("bk,bk->b", d1, d1)
    e = np.einsum("bk,bk->b", d2, d2)
    f = np.einsum("bk,bk->b", d2, r)
    c = np.einsum("bk,bk->b", d1, r)
    b = np.einsum("bk,bk->b", d1, d2)
    eps = 1e-12
    denom = a * e - b * b
    s = np.where(denom > eps, np.clip((b * f - c * e) / np.where(denom > eps, denom, 1), 0, 1), 0.0)
    s = np.where(a <= eps, 0.0, s)
    t = np.where(e > eps, (b * s + f) / np.where(e > eps, e, 1), 0.0)
    s2 = np.where(t < 0, np.clip(-c / np.where(a > eps, a, 1), 0, 1), np.where(t > 1, np.clip((b - c) / np.where(a > eps, a, 1), 0, 1), s))
    s = np.where(a > eps, s2, 0.0)
    t = np.clip(t, 0, 1)
    return p1 + s[:, None] * d1, p2 + t[:, None] * d2


def _shift_plan(plan):
    """initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol, traj_planner.cpp:409-421, 304-314."""
    out = np.empty_like(plan)
    out[:, :-1] = plan[:, 1:]
    out[:, -1] = plan[:, -1, -1][:, None, :]
    return out


def build_rows(cfg, pos, init, goals, radius, neigh, seg_sel, downwash=DOWNWASH):
    """DLSC-GC half-spaces of every agent against its selected neighbours.

    init: [N, M, 6, 3] shifted previous plans (own initial trajectory and the others' prediction)
    goals: [N, 3] current goal points, neigh: [N, J] neighbour ids, seg_sel: list (len J) of the
    segment lists kept for the j-th neighbour.  Returns (row_offsets, rows)."""
    N, M, n1, _ = init.shape
    dim = cfg.world_dimension
    dw = downwash if dim == 3 else 1.0
    scale = np.array([1.0, 1.0, 1.0 / dw])
    rows_per_agent = []
    J = neigh.shape[1]
    coll = 2 * radius
    all_rows = []
    for j in range(J):
        other = neigh[:, j]
        mine_t = init.astype(np.float64) * scale
        oth = init[other].astype(np.float64)
        oth_t = oth * scale
        segs = seg_sel[j]
        for m in segs:
            if m < M - 1:
                rel = mine_t[:, m] - oth_t[:, m]  # [N, 6, 3]
                cp = hull_closest_point(rel)
                nrm = np.linalg.norm(cp, axis=1)
                bad = nrm < 1e-9
                n_t = cp / np.where(bad, 1.0, nrm)[:, None]
                n_t[bad] = np.array([1.0, 0.0, 0.0])
                normal = n_t * scale
                dvals = 0.5 * (coll + np.einsum("nik,nk->ni", rel, n_t))  # [N, 6]
                for i in range(n1):
                    if m == 0 and i < cfg.phi:
                        continue
                    r = np.zeros(N, abi.ROW_DTYPE)
                    r["normal"] = normal.astype(np.float32)
                    r["point"] = oth[:, m, i].astype(np.float32)
                    r["d"] = dvals[:, i]
                    r["m"], r["i"], r["obs"], r["slack"], r["flags"] = m, i, j, -1, 0
                    all_rows.append(r)
            else:
                p1 = oth_t[:, -1, -1]
                q1 = goals[other].astype(np.float64) * scale
                p2 = mine_t[:, -1, -1]
                q2 = goals.astype(np.float64) * scale
                c1, c2 = segment_closest_points(p1, q1, p2, q2)
                diff = c2 - c1
                dist = np.linalg.norm(diff, axis=1)
                bad = dist < 1e-9
                n_t = diff / np.where(bad, 1.0, dist)[:, None]
                n_t[bad] = np.array([1.0, 0.0, 0.0])
                normal = n_t * scale
                point = c1 / scale
                dval = 0.5 * (coll + dist)
                for i in range(n1):
                    r = np.zeros(N, abi.ROW_DTYPE)
                    r["normal"] = normal.astype(np.float32)
                    r["point"] = point.astype(np.float32)
                    r["d"] = dval
                    r["m"], r["i"], r["obs"], r["slack"], r["flags"] = m, i, j, -1, 0
                    all_rows.append(r)
    K = len(all_rows)
    rows = np.stack(all_rows, axis=1).reshape(-1) if K else np.zeros(0, abi.ROW_DTYPE)  # [N*K], agent-major
    row_offsets = (np.arange(N + 1) * K).astype(np.int64)
    return row_offsets, np.ascontiguousarray(rows)


def _neighbours(pos, J, group):
    """J nearest neighbours of every agent inside its own swarm of `group` agents."""
    N = pos.shape[0]
    out = np.zeros((N, J), np.int64)
    for g0 in range(0, N, group):
        p = pos[g0:g0 + group].astype(np.float64)
        d = np.linalg.norm(p[:, None, :] - p[None, :, :], axis=2)
        np.fill_diagonal(d, np.inf)
        out[g0:g0 + group] = np.argsort(d, axis=1)[:, :J] + g0
    return out


def generate(cfg, solver, n_swarms=1, swarm_size=64, seed=2000, steps=6, infeasible_frac=0.05,
             waypoint_ahead=5.0, keep=1, spread=1.0):
    """Closed-loop synthetic workload: `n_swarms` independent swarms of `swarm_size` agents.

    solver(agents, row_offsets, rows, sfc) -> dict(goal [N,3] f32, ctrl [N,M,6,3] f32,
    status [N] i32, goal_status [N] i32) plans one batch (goal LP + trajectory QP).
    Returns the inputs of replanning step `steps` (0-based) as a dict with agents, row_offsets,
    rows, sfc (None), K (rows per agent) and the bookkeeping needed to continue the loop.
    keep > 1: out["history"] holds the inputs of the last `keep` consecutive replanning steps (oldest first, the returned
    step last) -- the same agents 0.2 s apart, so that a launch order derived from one step's iteration counts is the hint
    a closed loop really has for the next step.  spread < 1 packs every swarm into that fraction of the arena (denser
    traffic: more active half-spaces per agent)."""
    rng = np.random.default_rng(seed)
    dim, M = cfg.world_dimension, cfg.M
    N = n_swarms * swarm_size
    wmin = np.array(cfg.world_min[:]) + 1.0
    wmax = np.array(cfg.world_max[:]) - 1.0
    wmin[2], wmax[2] = cfg.world_min[2] + 0.5, cfg.world_max[2] - 0.5
    side = int(np.ceil(np.sqrt(swarm_size)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    cell = np.stack([gx.ravel(), gy.ravel()], 1)[:swarm_size]
    pitch = (wmax[:2] - wmin[:2]) / side * spread
    pos = np.zeros((N, 3))
    goal_final = np.zeros((N, 3))
    for s in range(n_swarms):
        sl = slice(s * swarm_size, (s + 1) * swarm_size)
        jit = rng.uniform(0.2, 0.8, (swarm_size, 2))
        pos[sl, :2] = wmin[:2] + (cell + jit) * pitch
        pos[sl, 2] = rng.uniform(wmin[2], wmax[2], swarm_size) if dim == 3 else cfg.world_z_2d
        perm = rng.permutation(swarm_size)
        gj = rng.uniform(0.2, 0.8, (swarm_size, 2))
        goal_final[sl, :2] = wmin[:2] + (cell[perm] + gj) * pitch
        goal_final[sl, 2] = rng.uniform(wmin[2], wmax[2], swarm_size) if dim == 3 else cfg.world_z_2d
    pos = pos.astype(np.float32)
    goal_final = goal_final.astype(np.float32)
    vel = np.zeros((N, 3), np.float32)
    acc = np.zeros((N, 3), np.float32)
    radius = 0.15
    cur_goal = pos.copy()
    # constant-position "previous plan" for the first step (planConstVelTraj with zero velocity)
    plan = np.repeat(np.repeat(pos[:, None, None, :], M, 1), cfg.n + 1, 2).astype(np.float32)
    J = 5 if swarm_size > 5 else max(swarm_size - 1, 1)
    seg_sel = [list(range(M))] * min(J, 4) + ([[M - 2, M - 1]] if J == 5 else [])
    out = None
    history = []
    lp_failed = np.zeros(N, bool)
    for step in range(steps + 1):
        init = plan if step == 0 else _shift_plan(plan)
        neigh = _neighbours(pos, J, swarm_size)
        to_goal = goal_final.astype(np.float64) - pos
        dist = np.linalg.norm(to_goal, axis=1)
        ahead = np.minimum(dist, waypoint_ahead)
        wp = (pos + to_goal / np.maximum(dist, 1e-9)[:, None] * ahead[:, None]).astype(np.float32)
        # an agent whose goal LP was infeasible keeps its sub-goal for one step (in the reference the
        # grid planner hands out conflict-free waypoints, multi_sync_simulator.cpp:308-466)
        wp = np.where(lp_failed[:, None], cur_goal, wp).astype(np.float32)
        row_offsets, rows = build_rows(cfg, pos, init, cur_goal, radius, neigh, seg_sel)
        agents = np.zeros(N, abi.AGENT_DTYPE)
        agents["position"], agents["velocity"], agents["acceleration"] = pos, vel, acc
        agents["current_goal_point"], agents["next_waypoint"] = cur_goal, wp
        agents["id"] = np.arange(N)
        agents["n_obs"], agents["n_slack_obs"] = J, 0
        agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = 1.0, 2.0, radius, 1.0
        out = dict(agents=agents, row_offsets=row_offsets, rows=rows, sfc=None,
                   K=int(row_offsets[1] - row_offsets[0]) if N else 0, step=step)
        if step > steps - keep:
            history.append(out)
        if step == steps:
            break
        res = solver(agents, row_offsets, rows, None)
        ok = (res["status"] == abi.OK)
        gok = (res["goal_status"] != abi.INFEASIBLE)
        lp_failed = ~gok
        cur_goal = np.where(gok[:, None], res["goal"], cur_goal).astype(np.float32)
        new_plan = np.where(ok[:, None, None, None], res["ctrl"], init)  # failsafe: traj_planner.cpp:776
        plan = new_plan.astype(np.float32)
        # agent_manager.cpp:51-57 + trajectory.cpp:110-199: state = getStateAt(dt), which lands on
        # the start of segment 1 (t_normalized = 0), float32 control-point differences
        n = cfg.n
        s1 = np.float32(n / cfg.dt)
        s2 = np.float32((n - 1) / cfg.dt)
        c0, c1, c2 = plan[:, 1, 0], plan[:, 1, 1], plan[:, 1, 2]
        pos = c0.copy()
        v0 = (c1 - c0) * s1
        v1 = (c2 - c1) * s1
        vel = v0
        acc = (v1 - v0) * s2
    # deliberately infeasible subset: two opposing half-spaces on one control point (the same agents in every kept step)
    n_bad = int(round(infeasible_frac * N))
    bad = np.zeros(N, bool)
    if n_bad > 0 and out["K"] >= 2:
        bad_ids = rng.choice(N, n_bad, replace=False)
        bad[bad_ids] = True
        K = out["K"]
        for o in history:
            rows = o["rows"].reshape(N, K)
            for a in bad_ids:
                r0, r1 = rows[a, K - 1].copy(), rows[a, K - 2].copy()
                nrm = np.array([0.6, 0.0, 0.8], np.float32)
                r1["normal"], r0["normal"] = nrm, -nrm
                r1["point"] = r0["point"] = o["agents"]["position"][a]
                r1["d"] = r0["d"] = 0.25
                r1["m"] = r0["m"] = M - 2
                r1["i"] = r0["i"] = 2
                rows[a, K - 1], rows[a, K - 2] = r0, r1
            o["rows"] = np.ascontiguousarray(rows.reshape(-1))
    for o in history:
        o["infeasible_mask"] = bad
    out["history"] = history
    return out


def swarm_missions(cfg, n_swarms=1, swarm_size=64, seed=2000, spread=1.0):
    """Start and goal points of generate()'s swarms (the same random draws): [N, 3] float32 each.  For the device-resident closed
    loop of the same workload (harness.ClosedLoop(neighbours="knn", knn=(swarm_size, 5), tail=(4, 2), prune=True))."""
    rng = np.random.default_rng(seed)
    dim = cfg.world_dimension
    N = n_swarms * swarm_size
    wmin = np.array(cfg.world_min[:]) + 1.0
    wmax = np.array(cfg.world_max[:]) - 1.0
    wmin[2], wmax[2] = cfg.world_min[2] + 0.5, cfg.world_max[2] - 0.5
    side = int(np.ceil(np.sqrt(swarm_size)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    cell = np.stack([gx.ravel(), gy.ravel()], 1)[:swarm_size]
    pitch = (wmax[:2] - wmin[:2]) / side * spread
    pos = np.zeros((N, 3))
    goal_final = np.zeros((N, 3))
    for s in range(n_swarms):
        sl = slice(s * swarm_size, (s + 1) * swarm_size)
        jit = rng.uniform(0.2, 0.8, (swarm_size, 2))
        pos[sl, :2] = wmin[:2] + (cell + jit) * pitch
        pos[sl, 2] = rng.uniform(wmin[2], wmax[2], swarm_size) if dim == 3 else cfg.world_z_2d
        perm = rng.permutation(swarm_size)
        gj = rng.uniform(0.2, 0.8, (swarm_size, 2))
        goal_final[sl, :2] = wmin[:2] + (cell[perm] + gj) * pitch
        goal_final[sl, 2] = rng.uniform(wmin[2], wmax[2], swarm_size) if dim == 3 else cfg.world_z_2d
    return pos.astype(np.float32), goal_final.astype(np.float32)


def permute_batch(agents, row_offsets, rows, perm):
    """The same batch with agent `perm[s]` stored at index `s` (agent records and their CSR row blocks).
    Planners that keep their local agents sorted by expected cost (previous step's iteration count) get the
    most-expensive-first launch order without an indirection, and contiguous output ranges per cost class."""
    perm = np.asarray(perm, dtype=np.int64)
    ro = np.asarray(row_offsets, dtype=np.int64)
    cnt = (ro[1:] - ro[:-1])[perm]
    new_ro = np.zeros(len(perm) + 1, dtype=np.asarray(row_offsets).dtype)
    np.cumsum(cnt, out=new_ro[1:])
    # gather index of every row: start of its source block + position inside the block
    src = np.repeat(ro[:-1][perm], cnt) + (np.arange(int(cnt.sum())) - np.repeat(new_ro[:-1].astype(np.int64), cnt))
    return agents[perm].copy(), new_ro, rows[src].copy()


def x2048(cfg, n_agents=2048, seed=5000, spacing=0.85, jitter=0.15, sense=3.0):
    """BASELINE.json configs[4] ("stress: 2048 synthetic agents, maze10_dense-style constraint density"): agents on a
    jittered grid in 2-D, each with a smooth previous plan, every agent within L-infinity `sense` metres is a neighbour
    (about 50 neighbours -> ~2 800 DLSC-GC rows per agent once generated), one safe-flight-corridor box per segment
    around the shifted previous plan (0.25-0.75 m margins, the corridor widths of maze10_dense), communication range
    from `cfg`.  Returns the inputs of the device generator (plans, goals, neighbour lists) and of the optimizer
    (agent records consistent with the shifted plans, SFC boxes).  The shifted previous plan satisfies every box,
    velocity/acceleration and communication row, so the QPs are feasible unless the neighbours squeeze an agent."""
    rng = np.random.default_rng(seed)
    M, n = cfg.M, cfg.n
    side = int(np.ceil(np.sqrt(n_agents)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    cell = np.stack([gx.ravel(), gy.ravel()], 1)[:n_agents].astype(np.float64)
    z = float(cfg.world_z_2d)
    origin = -0.5 * spacing * (side - 1)
    p0 = np.zeros((n_agents, 3))
    p0[:, :2] = origin + cell * spacing + rng.uniform(-jitter, jitter, (n_agents, 2))
    p0[:, 2] = z
    # previous plans: rest-to-rest quintic moves p0 + D s(t / T), s(u) = 10u^3 - 15u^4 + 6u^5, like real plans of the
    # reference (which end with a full stop, traj_optimizer.cpp:515-524): the shifted plan is then itself a feasible
    # trajectory, the premise of the DLSC feasibility guarantee.  |D| <= 0.12 keeps every pair beyond the collision
    # distance (grid spacing - 2 jitter - 2 |D| > 0.3).  Control points: exact Bernstein form of each segment.
    D = np.zeros((n_agents, 3))
    D[:, :2] = rng.uniform(-0.12, 0.12, (n_agents, 2))
    T = M * cfg.dt
    tau = np.arange(n + 1) / n
    from math import comb
    colloc = np.array([[comb(n, i) * t ** i * (1 - t) ** (n - i) for i in range(n + 1)] for t in tau])
    plans = np.zeros((n_agents, M, n + 1, 3), np.float32)
    for m in range(M):
        u = (m + tau) * cfg.dt / T
        sv = 10 * u ** 3 - 15 * u ** 4 + 6 * u ** 5          # samples of the quintic at 6 nodes of the segment
        ctrl = np.linalg.solve(colloc, sv)                    # its 6 Bernstein control points (exact: degree 5)
        plans[:, m] = (p0[:, None, :] + D[:, None, :] * ctrl[None, :, None]).astype(np.float32)
    plans[:, 1:, 0] = plans[:, :-1, n]
    goals = plans[:, -1, -1] + 0.0
    goals[:, :2] += rng.uniform(-0.2, 0.2, (n_agents, 2)).astype(np.float32)  # inside the last corridor box (margin >= 0.25)
    # neighbours within the sensing box (multi_sync_simulator.cpp:489-491 applies the communication range the same way)
    order = np.argsort(p0[:, 0], kind="stable")
    xs = p0[order, 0]
    offs, ids = [0], []
    for a in range(n_agents):
        lo, hi = np.searchsorted(xs, p0[a, 0] - sense), np.searchsorted(xs, p0[a, 0] + sense, side="right")
        cand = order[lo:hi]
        cand = cand[(np.abs(p0[cand, 1] - p0[a, 1]) <= sense) & (cand != a)]
        ids.append(np.sort(cand))
        offs.append(offs[-1] + len(cand))
    nbr_ids = np.concatenate(ids).astype(np.int32)
    # shifted plans = what every agent starts from (traj_planner.cpp:412-421)
    init = np.concatenate([plans[:, 1:], np.repeat(plans[:, -1:, -1:], n + 1, axis=2)], axis=1)
    agents = np.zeros(n_agents, abi.AGENT_DTYPE)
    agents["id"] = np.arange(n_agents)
    agents["position"] = init[:, 0, 0]
    agents["velocity"] = (init[:, 0, 1] - init[:, 0, 0]) * np.float32(n / cfg.dt)
    agents["acceleration"] = ((init[:, 0, 2] - init[:, 0, 1]) - (init[:, 0, 1] - init[:, 0, 0])) * np.float32(n * (n - 1) / cfg.dt ** 2)
    agents["current_goal_point"] = goals
    # the grid planner's waypoint lies in the last corridor box too (goal LP: the sub-goal slides from the waypoint
    # towards the current goal inside that box, goal_optimizer.cpp:138-198)
    wp = goals.copy()
    wp[:, :2] += rng.uniform(-0.04, 0.04, (n_agents, 2))
    agents["next_waypoint"] = wp.astype(np.float32)
    agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = 1.0, 2.0, 0.15, 1.0
    agents["n_obs"] = np.diff(offs)
    margin = rng.uniform(0.25, 0.75, (n_agents, M, 1, 3)).astype(np.float32)
    sfc = np.concatenate([init.min(axis=2, keepdims=True) - margin, init.max(axis=2, keepdims=True) + margin], axis=3)
    sfc = sfc.reshape(n_agents, M, 6).astype(np.float32)
    return dict(plans=plans, init=init, goals=goals.astype(np.float32), radius=np.full(n_agents, 0.15),
                downwash=np.full(n_agents, DOWNWASH), nbr_offsets=np.array(offs, np.int32), nbr_ids=nbr_ids,
                agents=agents, sfc=np.ascontiguousarray(sfc))


def swapping_pairs(cfg, n_agents, seed=5000, spacing=0.85, jitter=0.15, hop=4):
    """BASELINE.json configs[4] as a MOVING crowd: the start points of x2048 (jittered grid); every agent swaps places with the agent `hop` grid cells further
    along x (3.4 m): counter-flowing pairs everywhere, and the density of the crowd stays what it is."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n_agents)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    cell = np.stack([gx.ravel(), gy.ravel()], 1)[:n_agents]
    origin = -0.5 * spacing * (side - 1)
    start = np.zeros((n_agents, 3), np.float32)
    start[:, :2] = origin + cell * spacing + rng.uniform(-jitter, jitter, (n_agents, 2))
    start[:, 2] = cfg.world_z_2d
    index = {(int(a), int(b)): k for k, (a, b) in enumerate(cell)}
    partner = np.arange(n_agents)
    for k, (a, b) in enumerate(cell):
        q = (int(a) + hop, int(b)) if (int(a) // hop) % 2 == 0 else (int(a) - hop, int(b))
        partner[k] = index.get(q, k)
    partner = np.where(partner[partner] == np.arange(n_agents), partner, np.arange(n_agents))  # pairs only
    return start, start[partner].copy()
