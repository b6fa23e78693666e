"""Row N1 (DLSC-GC half-space generation), CPU tier: the oracle's restatement against the reference's own
openGJK (oracle/_ref, built from /root/reference where that exists), and the device source (lsc_core.hpp,
host build) against the oracle."""
import ctypes as C

import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import emul_lib as E
import lsc_cases
import oracle_lib as O

# float32 geometry with a different but equivalent operation order in the 3x3 inverse: parity is to a few ulp
NRM_TOL = 3e-6
D_TOL = 3e-6


def _cmp_rows(a, b, nrm_tol=NRM_TOL, d_tol=D_TOL):
    assert len(a) == len(b)
    for k in ("m", "i", "slack", "flags"):
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(a["normal"] - b["normal"]).max() <= nrm_tol
    assert np.abs(a["point"] - b["point"]).max() <= 1e-5
    assert np.abs(a["d"] - b["d"]).max() <= d_tol


def _oracle_rows(cfg, sw, first=False, dyn=None):
    if first:
        init = np.stack([O.const_vel_plan(sw["position"][a], sw["velocity"][a], cfg.M, cfg.n, cfg.dt) for a in range(len(sw["goals"]))])
    else:
        init = np.stack([O.shift_plan(p) for p in sw["plans"]])
    return O.generate_dlscgc(cfg, init, sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], dyn=dyn)


def test_hull_enumeration_equals_reference_opengjk():
    r = O.ref_gjk()
    if r is None:
        pytest.skip("oracle/_ref/libref_gjk.so not built (needs /root/reference)")
    rng = np.random.default_rng(1)
    inside = 0
    for t in range(3000):
        c = rng.normal(size=3) * rng.uniform(0.2, 3)
        pts = c + rng.normal(size=(6, 3)) * rng.uniform(0.05, 1.0)
        if t % 5 == 0:
            pts[:, 2] = 0  # 2-D runs / downwash-ignored case: coplanar hull
        d, v = O.hull_closest_point(pts)
        v2 = np.zeros(3)
        p = np.ascontiguousarray(pts)
        d2 = r.ref_gjk_point_hull(p.ctypes.data_as(C.POINTER(C.c_double)), 6, v2.ctypes.data_as(C.POINTER(C.c_double)))
        inside += d2 == 0
        assert np.abs(v - v2).max() < 1e-9 and abs(d - d2) < 1e-9
    assert inside > 50  # origin inside the hull (colliding plans): both return the zero vector


@pytest.mark.parametrize("M,dim", [(5, 3), (10, 3), (10, 2)])
def test_oracle_rows_identical_with_reference_gjk(M, dim):
    if O.ref_gjk() is None:
        pytest.skip("oracle/_ref/libref_gjk.so not built (needs /root/reference)")
    cfg = abi.make_config(dim=dim, M=M)
    sw = lsc_cases.random_swarm(cfg, 12, seed=M + dim)
    own = _oracle_rows(cfg, sw)
    O.use_reference_gjk(True)
    try:
        ref = _oracle_rows(cfg, sw)
    finally:
        O.use_reference_gjk(False)
    assert np.array_equal(own[0], ref[0])
    _cmp_rows(own[1], ref[1], nrm_tol=1.5e-7, d_tol=1e-9)  # float32 rounding of a 1e-13 difference at most


@pytest.mark.parametrize("M,dim,first", [(5, 3, False), (10, 3, False), (10, 2, False), (5, 3, True)])
def test_device_source_matches_oracle(M, dim, first):
    cfg = abi.make_config(dim=dim, M=M)
    sw = lsc_cases.random_swarm(cfg, 16, seed=100 + M + dim)
    ro, rows = _oracle_rows(cfg, sw, first)
    if first:
        eo, erows = E.generate_lsc(cfg, None, sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"],
                                   position=sw["position"], velocity=sw["velocity"])
    else:
        eo, erows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"])
    assert np.array_equal(ro, eo)
    assert len(rows) > 200
    _cmp_rows(erows, rows)


@pytest.mark.parametrize("M,dim", [(5, 3), (10, 2), (10, 3)])
def test_device_source_matches_oracle_in_a_crowd(M, dim):
    """Vehicles on top of each other: the relative hull contains the origin or grazes it, so the generator's
    separating-plane shortcut does not apply and its tetrahedron search decides (lsc_core.hpp hull6_closest_to_origin);
    the oracle always runs the full enumeration."""
    cfg = abi.make_config(dim=dim, M=M)
    sw = lsc_cases.crowd(cfg, 12, seed=900 + M + dim)
    ro, rows = _oracle_rows(cfg, sw)
    eo, erows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"])
    assert np.array_equal(ro, eo)
    zero = (rows["normal"] == 0).all(axis=1)
    assert zero.sum() >= 30, "the crowd holds pairs whose hull contains the origin (zero normal: the row is skipped, traj_optimizer.cpp:421)"
    _cmp_rows(erows, rows)


def test_shifted_plans_satisfy_their_own_half_spaces():
    """DLSC guarantee (the reason the QP stays feasible): when the previous plans keep the collision distance,
    every agent's shifted plan satisfies every generated row."""
    cfg = abi.make_config(dim=3, M=5)
    rng = np.random.default_rng(3)
    N = 10
    sw = lsc_cases.random_swarm(cfg, N, seed=3, spread=0.1)
    # well separated agents moving in parallel, far-apart goals
    for a in range(N):
        base = np.array([2.0 * a, 0.0, 1.0])
        for m in range(cfg.M):
            for i in range(cfg.n + 1):
                sw["plans"][a, m, i] = base + np.array([0.0, 0.1 * (m + i / cfg.n), 0.0])
        sw["goals"][a] = sw["plans"][a, -1, -1] + np.float32([0, 1.0, 0])
    offs, rows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"])
    init = np.stack([O.shift_plan(p) for p in sw["plans"]])
    for a in range(N):
        r = rows[offs[a]:offs[a + 1]]
        if len(r) == 0:
            continue
        cp = init[a][r["m"], r["i"]].astype(np.float64)
        lhs = np.einsum("kd,kd->k", r["normal"].astype(np.float64), cp - r["point"].astype(np.float64)) - r["d"]
        assert lhs.min() > -1e-6


# ---- dynamic (non-agent) obstacles -------------------------------------------------------------------------------
def _binom(a, b):
    from math import comb
    return comb(a, b)


def test_power_to_bernstein_closed_form_is_b_inverse():
    """The size prediction multiplies power-basis coefficients by B_inv (traj_planner.cpp:352-358); oracle and device
    use B_inv(k, i) = C(i,k) / C(n,k).  B as in polynomial.hpp:280-293."""
    n = 5
    B = np.zeros((n + 1, n + 1))
    for i in range(n + 1):
        for j in range(i, n + 1):
            B[i, j] = _binom(n, i) * _binom(n - i, n - j) * (-1) ** (j - i)
    Binv = np.linalg.inv(B)
    closed = np.array([[(_binom(i, k) / _binom(n, k)) if i >= k else 0.0 for i in range(n + 1)] for k in range(n + 1)])
    assert np.abs(Binv - closed).max() < 1e-12


@pytest.mark.parametrize("M,dim,first", [(5, 3, False), (10, 3, False), (10, 2, False), (5, 3, True)])
def test_device_source_matches_oracle_with_dynamic_obstacles(M, dim, first):
    cfg = abi.make_config(dim=dim, M=M)
    sw = lsc_cases.random_swarm(cfg, 12, seed=300 + M + dim)
    # obstacle 0 moves exactly like agent 0's path in the first-plan case (coinciding paths -> the reference's heuristic)
    dyn = lsc_cases.random_dynamic_obstacles(4, seed=9 + M, first_at=(sw["position"][0], sw["velocity"][0]) if first else None)
    ro, rows = _oracle_rows(cfg, sw, first, dyn=dyn)
    kw = dict(position=sw["position"], velocity=sw["velocity"]) if first else {}
    eo, erows = E.generate_lsc(cfg, None if first else sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"],
                               sw["nbr_ids"], dyn=dyn, **kw)
    assert np.array_equal(ro, eo)
    per = cfg.M * (cfg.n + 1) - cfg.phi
    assert len(rows) == per * (int(sw["nbr_offsets"][-1]) + 12 * 4)
    _cmp_rows(erows, rows)
    dynrows = rows[rows["flags"] == abi.ROW_DYNAMIC]
    assert len(dynrows) == 12 * 4 * per and set(np.unique(dynrows["slack"])) == {0, 1, 2, 3}
    assert np.isfinite(dynrows["normal"]).all() and (np.linalg.norm(dynrows["normal"], axis=1) > 0).all()


def test_dynamic_obstacle_size_prediction_profile():
    """d = predicted size + agent radius: radius + 0.5 a t^2 sampled as Bernstein control points up to the uncertainty
    horizon, constant afterwards (traj_planner.cpp:338-368)."""
    cfg = abi.make_config(dim=3, M=5)
    sw = lsc_cases.random_swarm(cfg, 2, seed=1)
    sw["nbr_offsets"], sw["nbr_ids"] = np.zeros(3, np.int32), np.zeros(0, np.int32)
    dyn = dict(position=np.float32([[0, 0, 1]]), velocity=np.float32([[0.5, 0, 0]]), radius=np.array([0.3]),
               max_acc=np.array([2.0]), downwash=np.array([1.0]), size_prediction=1, horizon=0.6)
    _, rows = _oracle_rows(cfg, sw, dyn=dyn)
    r0 = rows[:cfg.M * (cfg.n + 1) - cfg.phi]
    ra = sw["radius"][0]
    t_end = {m: (m + 1) * cfg.dt for m in range(cfg.M)}
    for m in range(cfg.M):
        last = r0[(r0["m"] == m) & (r0["i"] == cfg.n)]["d"][0]  # last control point = value at the end of the segment
        t = min(t_end[m], 0.6)
        assert abs(last - (0.3 + 0.5 * 2.0 * t * t + ra)) < 1e-12
    off = dict(dyn, size_prediction=0)
    _, rows2 = _oracle_rows(cfg, sw, dyn=off)
    assert np.allclose(rows2[:len(r0)]["d"], 0.3 + ra)


def test_disturbed_neighbour_is_predicted_to_stand_still():
    """TrajPlanner::checkObstacleDisturbance (traj_planner.cpp:329-336): neighbours observed farther than reset_threshold from
    where their shifted plan starts are predicted at rest at the observed position -- device source against the oracle chain."""
    cfg = abi.make_config(dim=3, M=5)
    sw = lsc_cases.random_swarm(cfg, 10, seed=77)
    obs = sw["plans"][:, 1, 0].copy()          # where the shifted plans start
    obs[[2, 5, 6]] += np.float32([0.4, -0.3, 0.2])  # three agents were pushed away (0.54 m)
    obs[8] += np.float32([0.05, 0, 0])          # below the threshold
    thr = 0.5
    init = np.stack([O.shift_plan(p) for p in sw["plans"]])
    pred, n_reset = init.copy(), 0
    for j in range(10):
        pred[j], r = O.obstacle_disturbance(init[j], obs[j], thr)
        n_reset += r
    assert n_reset == 3 and np.array_equal(pred[8], init[8]) and (pred[5] == obs[5]).all()
    # oracle rows: own trajectory from `init`, neighbours from `pred`
    offs, out = [0], []
    for a in range(10):
        nb = sw["nbr_ids"][sw["nbr_offsets"][a]:sw["nbr_offsets"][a + 1]]
        mixed = pred.copy()
        mixed[a] = init[a]
        _, r = O.generate_dlscgc(cfg, mixed, sw["goals"], sw["radius"], sw["downwash"], [0, len(nb)], nb, local_ids=[a])
        out.append(r)
        offs.append(offs[-1] + len(r))
    rows = np.concatenate(out)
    eo, erows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"],
                               observed=obs, reset_threshold=thr)
    assert np.array_equal(eo, np.array(offs))
    _cmp_rows(erows, rows)
    _, plain = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"])
    assert np.abs(plain["normal"] - erows["normal"]).max() > 1e-3  # the reset does change rows


@pytest.mark.parametrize("M,dim,comm", [(5, 3, 3.0), (10, 2, 3.0), (5, 3, -1.0)])
def test_waypoint_trap_releases_the_same_obstacles_as_the_oracle(M, dim, comm):
    """TrajPlanner::checkWaypointTrap (traj_planner.cpp:708-735): device helpers (host build) against the oracle's
    restatement, row for row; the cases cover trapped / free waypoints and obstacles within / out of reach."""
    cfg = abi.make_config(dim=dim, M=M, communication_range=comm, world_use_octomap=True)
    N = 18
    sw = lsc_cases.random_swarm(cfg, N, seed=500 + M)
    dyn = lsc_cases.random_dynamic_obstacles(3, seed=40 + M)
    ag, sfc = lsc_cases.trap_agents(cfg, sw, dyn, seed=5)
    trap = dict(agents=ag, sfc=sfc)
    eo, erows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], dyn=dyn,
                               trap=trap)
    _, before = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], dyn=dyn)
    released = trap["released"]
    per = cfg.M * (cfg.n + 1) - cfg.phi
    exp_rel = np.zeros(N, np.int32)
    for a in range(N):
        n_nbr = int(sw["nbr_offsets"][a + 1] - sw["nbr_offsets"][a])
        exp_rows, exp_rel[a] = O.waypoint_trap(cfg, before[eo[a]:eo[a + 1]], n_nbr, ag[a], sfc[a, -1], dyn)
        got = erows[eo[a]:eo[a + 1]]
        for k in ("normal", "point", "d", "m", "i", "slack", "flags"):
            assert np.array_equal(got[k], exp_rows[k]), (a, k)
        zeroed = (np.abs(got["normal"]).sum(axis=1) == 0) & (got["flags"] == abi.ROW_DYNAMIC)
        assert zeroed.sum() == exp_rel[a] * per
    assert np.array_equal(released, exp_rel)
    if comm > 0:
        assert (released > 0).sum() >= (2 if dim == 3 else 1) and (released == 0).sum() >= 2, released
    else:
        # unlimited range: the communication box is the default Box (the origin), every agent counts as trapped, so every
        # obstacle within reach of the waypoint is released (harmless in the reference: such launches have no obstacles)
        assert (released > 0).sum() >= 2


def test_waypoint_trap_on_the_first_plan_uses_the_default_communication_box():
    """On the first plan generateSFC only initialises the corridor (traj_planner.cpp:692-705): constructCommunicationRange
    has not run yet, the communication box is the default Box at the origin, every agent counts as trapped and every
    dynamic obstacle within reach of its waypoint is released -- device helpers and oracle agree, and differ from a later
    step's answer on the same geometry."""
    cfg = abi.make_config(dim=3, M=5, communication_range=3.0, world_use_octomap=True)
    N = 18
    sw = lsc_cases.random_swarm(cfg, N, seed=505)
    dyn = lsc_cases.random_dynamic_obstacles(3, seed=45)
    ag, sfc = lsc_cases.trap_agents(cfg, sw, dyn, seed=5)
    pos, vel = sw["plans"][:, 0, 0].copy(), np.zeros((N, 3), np.float32)
    trap = dict(agents=ag, sfc=sfc)
    eo, erows = E.generate_lsc(cfg, None, sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], position=pos,
                               velocity=vel, dyn=dyn, trap=trap)
    _, before = E.generate_lsc(cfg, None, sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], position=pos,
                               velocity=vel, dyn=dyn)
    first, later = np.zeros(N, np.int32), np.zeros(N, np.int32)
    for a in range(N):
        n_nbr = int(sw["nbr_offsets"][a + 1] - sw["nbr_offsets"][a])
        exp_rows, first[a] = O.waypoint_trap(cfg, before[eo[a]:eo[a + 1]], n_nbr, ag[a], sfc[a, -1], dyn, first_plan=True)
        _, later[a] = O.waypoint_trap(cfg, before[eo[a]:eo[a + 1]], n_nbr, ag[a], sfc[a, -1], dyn, first_plan=False)
        for k in ("normal", "point", "d"):
            assert np.array_equal(erows[eo[a]:eo[a + 1]][k], exp_rows[k]), (a, k)
    assert np.array_equal(trap["released"], first)
    assert (first >= later).all() and first.sum() > 0  # the default box traps everybody: never fewer releases than later


@pytest.mark.parametrize("M,dim,n_dyn", [(10, 2, 0), (10, 3, 2), (5, 3, 0)])
def test_pruned_rows_are_implied_rows_and_the_plans_do_not_change(M, dim, n_dyn):
    """The generator's presolve (dlscqp_lsc_gen.prune): every row it drops is implied by the agent's corridor / world box with
    a margin (checked corner by corner in numpy), the kept rows are the others in their original order, and goal LP + QP give
    the same verdicts and solutions with and without the dropped rows."""
    cfg = abi.make_config(dim=dim, M=M, communication_range=3.0, world_use_octomap=True, slack_collision_weight=100.0,
                          world_min=(-6.0, -6.0, 0.0), world_max=(6.0, 6.0, 2.5))
    N = 14
    sw = lsc_cases.random_swarm(cfg, N, seed=700 + M + dim, spread=1.6)
    if dim == 2:
        sw["plans"][..., 2] = np.float32(cfg.world_z_2d)
        sw["goals"][:, 2] = np.float32(cfg.world_z_2d)
    dyn = lsc_cases.random_dynamic_obstacles(n_dyn, seed=3) if n_dyn else None
    init = np.stack([O.shift_plan(p) for p in sw["plans"]])
    rng = np.random.default_rng(1)
    sfc = np.zeros((N, M, 6), np.float32)
    for a in range(N):
        for m in range(M):
            lo, hi = init[a, m].min(axis=0), init[a, m].max(axis=0)
            mg = rng.uniform(0.2, 0.6, 3).astype(np.float32)
            sfc[a, m, :3], sfc[a, m, 3:] = lo - mg, hi + mg
    full_ro, full = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], dyn=dyn)
    pr = dict(sfc=sfc)
    ro, rows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], dyn=dyn, prune=pr)
    assert pr["counts"].sum() == len(rows) < len(full) and (pr["counts"] > 0).any()
    wmin, wmax = np.array(cfg.world_min[:]), np.array(cfg.world_max[:])
    dropped = 0
    for a in range(N):
        f = full[full_ro[a]:full_ro[a + 1]]
        k = rows[ro[a]:ro[a + 1]]
        # kept rows: a subsequence of the full rows
        kept_idx, j = [], 0
        for i in range(len(f)):
            if j < len(k) and f[i].tobytes() == k[j].tobytes():
                kept_idx.append(i)
                j += 1
        assert j == len(k), a
        for i in sorted(set(range(len(f))) - set(kept_idx)):
            r = f[i]
            lo = np.maximum(sfc[a, r["m"], :3].astype(np.float64), wmin)[:dim]
            hi = np.minimum(sfc[a, r["m"], 3:].astype(np.float64), wmax)[:dim]
            if r["m"] == M - 1 and r["i"] == cfg.n and not (r["flags"] & 1):  # also a row of the goal LP, which has no variable bounds
                lo, hi = sfc[a, r["m"], :3].astype(np.float64)[:dim], sfc[a, r["m"], 3:].astype(np.float64)[:dim]
            nv = r["normal"].astype(np.float64)[:dim]
            mn = np.where(nv >= 0, nv * lo, nv * hi).sum()
            assert mn - (r["d"] + (nv * r["point"].astype(np.float64)[:dim]).sum()) >= 1e-4 - 1e-12
            dropped += 1
    assert dropped > 50
    ag = np.zeros(N, abi.AGENT_DTYPE)
    ag["position"], ag["current_goal_point"], ag["next_waypoint"] = init[:, 0, 0], sw["goals"], sw["goals"]
    ag["velocity"] = (init[:, 0, 1] - init[:, 0, 0]) * np.float32(cfg.n / cfg.dt)
    ag["max_vel"], ag["max_acc"], ag["radius"], ag["nominal_velocity"] = 1.0, 2.0, sw["radius"], 1.0
    ag["n_slack_obs"] = n_dyn
    a1 = E.plan_batch(cfg, ag, full_ro, full, sfc, n_dyn, True)
    a2 = E.plan_batch(cfg, ag, ro, rows, sfc, n_dyn, True)
    assert np.array_equal(a1["status"], a2["status"]) and np.array_equal(a1["goal_status"], a2["goal_status"])
    assert np.array_equal(a1["goal"], a2["goal"])
    ok = a1["status"] == abi.OK
    if ok.any():
        assert np.abs(a1["x"][ok] - a2["x"][ok]).max() < 1e-8


def test_pruning_by_the_world_box_alone_keeps_the_goal_lp_rows():
    """Without corridor boxes (world_use_octomap off) rows are implied by the variable bounds only -- which the goal LP does
    not have: the rows it shares with the QP (last point of the last segment, agent obstacles) all stay."""
    cfg = abi.make_config(dim=3, M=5, world_min=(-0.5, -0.5, 0.9), world_max=(0.5, 0.5, 1.4))  # most neighbours are far outside
    sw = lsc_cases.random_swarm(cfg, 14, seed=41, spread=2.5)
    args = (cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"])
    full_ro, full = E.generate_lsc(*args)
    pr = dict(sfc=None)
    ro, rows = E.generate_lsc(*args, prune=pr)
    assert 0 < len(rows) < len(full)
    lp = lambda r: int(((r["m"] == cfg.M - 1) & (r["i"] == cfg.n)).sum())  # noqa: E731
    assert lp(rows) == lp(full) > 20


def test_pruning_leaves_the_plans_of_the_stress_workload_unchanged():
    """The same on feasible problems: 96 agents of the stress workload (BASELINE.json configs[4]: 2-D, M = 10, corridor boxes,
    every agent within 3 m a neighbour).  Most rows go; verdicts identical, sub-goals bit-equal, solutions within 1e-8."""
    from dlsc_gc_planner_amd import workloads
    cfg = workloads.x2048_config()
    w = workloads.x2048(cfg, n_agents=96, seed=5002)
    args = (cfg, w["plans"], w["goals"], w["radius"], w["downwash"], w["nbr_offsets"], w["nbr_ids"])
    full_ro, full = E.generate_lsc(*args)
    pr = dict(sfc=w["sfc"])
    ro, rows = E.generate_lsc(*args, prune=pr)
    assert len(rows) < 0.5 * len(full), (len(rows), len(full))
    a1 = E.plan_batch(cfg, w["agents"], full_ro, full, w["sfc"], 0, True)
    a2 = E.plan_batch(cfg, w["agents"], ro, rows, w["sfc"], 0, True)
    assert np.array_equal(a1["status"], a2["status"]) and np.array_equal(a1["goal_status"], a2["goal_status"])
    assert np.array_equal(a1["goal"], a2["goal"])
    ok = a1["status"] == abi.OK
    assert ok.mean() > 0.8 and np.abs(a1["x"][ok] - a2["x"][ok]).max() < 1e-8
    print("stress workload: %d of %d rows kept by the presolve" % (len(rows), len(full)))
