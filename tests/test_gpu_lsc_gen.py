"""-m gpu, row N1: DLSC-GC half-spaces generated on the device against the oracle, and fed straight into the
QP kernel (no host round trip of the rows)."""
import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import lsc_cases
import oracle_lib as O
from test_lsc_gen import _cmp_rows, _oracle_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,dim,first,crowd", [(5, 3, False, False), (10, 3, False, False), (10, 2, False, False), (5, 3, True, False),
                                                (5, 3, False, True), (10, 2, False, True)])
def test_device_rows_match_oracle(gpu_required, M, dim, first, crowd):
    """crowd: vehicles on top of each other -- relative hulls that contain the origin (the tetrahedron search behind the
    generator's separating-plane test), graze it, or are flat."""
    cfg = abi.make_config(dim=dim, M=M)
    sw = lsc_cases.crowd(cfg, 48, seed=500 + M + dim) if crowd else lsc_cases.random_swarm(cfg, 48, seed=500 + M + dim)
    ro, rows = _oracle_rows(cfg, sw, first)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(48, sw["nbr_offsets"], sw["nbr_ids"], sw["radius"], sw["downwash"])
        goals = torch.from_numpy(sw["goals"]).cuda()
        if first:
            gen.generate(goals, position=torch.from_numpy(sw["position"]).cuda(), velocity=torch.from_numpy(sw["velocity"]).cuda())
        else:
            gen.generate(goals, plans=torch.from_numpy(sw["plans"]).cuda())
        go, grows = gen.rows_host()
    finally:
        opt.close()
    assert np.array_equal(go, ro)
    assert len(rows) > 1000
    _cmp_rows(grows, rows)


def test_generated_rows_feed_the_qp_kernel_directly(gpu_required):
    """Closed chain on the device: previous plans -> half-spaces -> goal LP + QP, compared with the oracle
    chain (oracle rows -> oracle LP/QP) on the same inputs."""
    cfg = abi.make_config(dim=3, M=5)
    N = 40
    sw = lsc_cases.random_swarm(cfg, N, seed=77, spread=3.0)
    rng = np.random.default_rng(5)
    agents = np.zeros(N, abi.AGENT_DTYPE)
    init = np.stack([O.shift_plan(p) for p in sw["plans"]])
    agents["position"] = init[:, 0, 0]
    n = cfg.n
    agents["velocity"] = (init[:, 0, 1] - init[:, 0, 0]) * np.float32(n / cfg.dt)
    agents["acceleration"] = ((init[:, 0, 2] - init[:, 0, 1]) - (init[:, 0, 1] - init[:, 0, 0])) * np.float32(n * (n - 1) / cfg.dt ** 2)
    agents["current_goal_point"] = sw["goals"]
    agents["next_waypoint"] = sw["goals"] + rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = 2.0, 20.0, 0.15, 1.0
    agents["n_obs"] = np.diff(sw["nbr_offsets"])
    ro, rows = _oracle_rows(cfg, sw)
    ref = O.plan_batch(cfg, agents, ro, rows, None, 0, True)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(N, sw["nbr_offsets"], sw["nbr_ids"], sw["radius"], sw["downwash"])
        gen.generate(torch.from_numpy(sw["goals"]).cuda(), plans=torch.from_numpy(sw["plans"]).cuda())
        db = opt.make_device_batch(agents, ro, rows[:0], None, 0, want_f64=True)  # no rows uploaded
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):  # point the batch at the generated rows
            setattr(db.c, k, gen.soa[k].data_ptr())
        ro32 = torch.from_numpy(ro.astype(np.int32)).cuda()
        db.c.row_offsets = ro32.data_ptr()
        db.c.max_rows_per_agent = int(np.diff(ro).max())
        db.launch()
        got = db.results()
    finally:
        opt.close()
    assert np.array_equal(got["goal_status"], ref["goal_status"])
    # the generated rows differ from the oracle's by float32 rounding of the last-segment geometry, so verdicts
    # are compared where the oracle's problem is not borderline and solutions to 1e-4 (north-star tolerance)
    same = got["status"] == ref["status"]
    assert same.mean() > 0.95
    ok = same & (ref["status"] == abi.OK)
    assert ok.sum() > 10
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() < 1e-4


@pytest.mark.parametrize("M,dim,first", [(5, 3, False), (10, 2, False), (5, 3, True)])
def test_device_rows_match_oracle_with_dynamic_obstacles(gpu_required, M, dim, first):
    cfg = abi.make_config(dim=dim, M=M)
    sw = lsc_cases.random_swarm(cfg, 32, seed=700 + M + dim)
    dyn = lsc_cases.random_dynamic_obstacles(5, seed=70 + M, first_at=(sw["position"][0], sw["velocity"][0]) if first else None)
    ro, rows = _oracle_rows(cfg, sw, first, dyn=dyn)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(32, sw["nbr_offsets"], sw["nbr_ids"], sw["radius"], sw["downwash"], dyn=dyn)
        goals = torch.from_numpy(sw["goals"]).cuda()
        dp, dv = torch.from_numpy(dyn["position"]).cuda(), torch.from_numpy(dyn["velocity"]).cuda()
        if first:
            gen.generate(goals, position=torch.from_numpy(sw["position"]).cuda(), velocity=torch.from_numpy(sw["velocity"]).cuda(),
                         dyn_position=dp, dyn_velocity=dv)
        else:
            gen.generate(goals, plans=torch.from_numpy(sw["plans"]).cuda(), dyn_position=dp, dyn_velocity=dv)
        go, grows = gen.rows_host()
    finally:
        opt.close()
    assert np.array_equal(go, ro)
    assert (grows["flags"] == abi.ROW_DYNAMIC).sum() == 32 * 5 * (M * (cfg.n + 1) - cfg.phi)
    _cmp_rows(grows, rows)


def test_generated_rows_with_dynamic_obstacles_feed_the_qp_kernel(gpu_required):
    """The same chain with two dynamic obstacles: their rows carry slack variables (soft constraints), the batch is
    planned by the slack-capable instantiation, and the slack cost per obstacle matches the oracle chain."""
    cfg = abi.make_config(dim=3, M=5, slack_collision_weight=100.0)
    N, D = 24, 2
    sw = lsc_cases.random_swarm(cfg, N, seed=78, spread=3.0)
    dyn = lsc_cases.random_dynamic_obstacles(D, seed=8, spread=3.0)
    rng = np.random.default_rng(6)
    agents = np.zeros(N, abi.AGENT_DTYPE)
    init = np.stack([O.shift_plan(p) for p in sw["plans"]])
    n = cfg.n
    agents["position"] = init[:, 0, 0]
    agents["velocity"] = (init[:, 0, 1] - init[:, 0, 0]) * np.float32(n / cfg.dt)
    agents["acceleration"] = ((init[:, 0, 2] - init[:, 0, 1]) - (init[:, 0, 1] - init[:, 0, 0])) * np.float32(n * (n - 1) / cfg.dt ** 2)
    agents["current_goal_point"] = sw["goals"]
    agents["next_waypoint"] = sw["goals"] + rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = 2.0, 20.0, 0.15, 1.0
    agents["n_obs"] = np.diff(sw["nbr_offsets"]) + D
    agents["n_slack_obs"] = D
    ro, rows = _oracle_rows(cfg, sw, dyn=dyn)
    ref = O.plan_batch(cfg, agents, ro, rows, None, D, True)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(N, sw["nbr_offsets"], sw["nbr_ids"], sw["radius"], sw["downwash"], dyn=dyn)
        gen.generate(torch.from_numpy(sw["goals"]).cuda(), plans=torch.from_numpy(sw["plans"]).cuda(),
                     dyn_position=torch.from_numpy(dyn["position"]).cuda(), dyn_velocity=torch.from_numpy(dyn["velocity"]).cuda())
        db = opt.make_device_batch(agents, ro, rows[:0], None, D, want_f64=True)
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(db.c, k, gen.soa[k].data_ptr())
        ro32 = torch.from_numpy(ro.astype(np.int32)).cuda()
        db.c.row_offsets = ro32.data_ptr()
        db.c.max_rows_per_agent = int(np.diff(ro).max())
        db.launch()
        got = db.results()
    finally:
        opt.close()
    assert np.array_equal(got["goal_status"], ref["goal_status"])
    same = got["status"] == ref["status"]
    assert same.mean() > 0.9
    ok = same & (ref["status"] == abi.OK)
    assert ok.sum() > 8
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() < 1e-4
    assert np.abs(got["slack_cost"][ok] - ref["slack_cost"][ok]).max() < 1e-4


def test_waypoint_trap_and_disturbance_reset_on_the_device(gpu_required):
    """TrajPlanner::checkObstacleDisturbance (:329-336) inside the half-space kernel and TrajPlanner::checkWaypointTrap
    (:708-735) as its own launch, against the oracle's restatements."""
    import lsc_cases
    import emul_lib as E
    cfg = abi.make_config(dim=3, M=5, communication_range=3.0, world_use_octomap=True)
    N = 18
    sw = lsc_cases.random_swarm(cfg, N, seed=505)
    dyn = lsc_cases.random_dynamic_obstacles(3, seed=45)
    ag, sfc = lsc_cases.trap_agents(cfg, sw, dyn, seed=5)
    obs = sw["plans"][:, 1, 0].copy()
    obs[[2, 5]] += np.float32([0.4, -0.3, 0.2])
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(N, sw["nbr_offsets"], sw["nbr_ids"], sw["radius"], sw["downwash"], dyn=dyn)
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        gen.generate(cu(sw["goals"]), plans=cu(sw["plans"]), dyn_position=cu(dyn["position"]), dyn_velocity=cu(dyn["velocity"]),
                     observed_position=cu(obs), reset_threshold=0.5)
        ro, before = gen.rows_host()
        agents_dev = torch.from_numpy(ag.view(np.uint8).reshape(-1)).cuda()
        released = torch.zeros(N, dtype=torch.int32, device="cuda")
        gen.waypoint_trap(agents_dev, cu(sfc), released)
        _, after = gen.rows_host()
        released = released.cpu().numpy()
    finally:
        opt.close()
    # the same chain through the host build of the device source (which the CPU tier pins to the oracle)
    trap = dict(agents=ag, sfc=sfc)
    eo, erows = E.generate_lsc(cfg, sw["plans"], sw["goals"], sw["radius"], sw["downwash"], sw["nbr_offsets"], sw["nbr_ids"], dyn=dyn,
                               observed=obs, reset_threshold=0.5, trap=trap)
    assert np.array_equal(ro, eo) and np.array_equal(released, trap["released"]) and (released > 0).sum() >= 2
    for k in ("m", "i", "slack", "flags"):
        assert np.array_equal(after[k], erows[k])
    assert np.abs(after["normal"] - erows["normal"]).max() <= 3e-6 and np.abs(after["d"] - erows["d"]).max() <= 3e-6
    # and directly against the oracle's trap on the device's own rows
    for a in range(N):
        n_nbr = int(sw["nbr_offsets"][a + 1] - sw["nbr_offsets"][a])
        exp, rel = O.waypoint_trap(cfg, before[ro[a]:ro[a + 1]], n_nbr, ag[a], sfc[a, -1], dyn)
        assert rel == released[a]
        for k in ("normal", "point", "d"):
            assert np.array_equal(after[ro[a]:ro[a + 1]][k], exp[k]), (a, k)


def test_device_rows_against_the_oracle_running_the_references_own_gjk(gpu_required):
    """oracle/_ref/libref_gjk.so is the reference's openGJK compiled from its own sources (oracle/Makefile); it travels to the
    GPU box.  With it plugged into the oracle as the point-to-hull routine, the device's half-spaces (exact hull
    enumeration instead of iterative GJK) still agree to float32 rounding."""
    if O.ref_gjk() is None:
        pytest.skip("oracle/_ref/libref_gjk.so not built (needs /root/reference at build time)")
    cfg = abi.make_config(dim=3, M=10)
    sw = lsc_cases.random_swarm(cfg, 24, seed=909)
    O.use_reference_gjk(True)
    try:
        ro, rows = _oracle_rows(cfg, sw, False)
    finally:
        O.use_reference_gjk(False)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(24, sw["nbr_offsets"], sw["nbr_ids"], sw["radius"], sw["downwash"])
        gen.generate(torch.from_numpy(sw["goals"]).cuda(), plans=torch.from_numpy(sw["plans"]).cuda())
        go, grows = gen.rows_host()
    finally:
        opt.close()
    assert np.array_equal(ro, go) and len(rows) > 1000
    _cmp_rows(grows, rows)
