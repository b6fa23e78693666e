"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import oracle_lib

pytestmark = pytest.mark.gpu

# north star: control points within 1e-4 relative, identical feasibility verdicts
CTRL_RTOL = 1e-4


def _compare(cfg, got, ref, x_tol=1e-6):
    assert np.array_equal(got["goal_status"], ref["goal_status"])
    gok = ref["goal_status"] != abi.INFEASIBLE
    assert np.array_equal(got["goal"][gok], ref["goal"][gok])  # float32, bit-exact
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["status"] == abi.OK
    if ok.any():
        scale = np.maximum(np.abs(ref["ctrl"][ok]).max(), 1.0)
        err = np.abs(got["ctrl"][ok].astype(np.float64) - ref["ctrl"][ok]).max()
        assert err <= CTRL_RTOL * scale, err
        assert err <= 1e-5, err  # in practice float32 rounding only
        # the objective at two points 1e-9 m apart differs by gradient x 1e-9, and the gradient of the jerk term grows with
        # control_input_weight: 1e-7 relative at the launch files' 0.01, 1e-6 at the code default 1 (src/param.cpp:71-80)
        assert np.allclose(got["cost"][ok], ref["cost"][ok], rtol=1e-7 if cfg.control_input_weight <= 0.1 else 1e-6, atol=1e-9)
    return int(ok.sum())


@pytest.fixture(scope="module")
def s64():
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    yield cfg, opt
    opt.close()


def test_s64_closed_loop_batch_matches_oracle(gpu_required, s64):
    cfg, opt = s64

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    wl = workloads.generate(cfg, solver, n_swarms=2, swarm_size=64, seed=2001)
    got = opt.plan(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, True, True)
    ref = oracle_lib.plan_batch(cfg, wl["agents"], wl["row_offsets"], wl["rows"], None, 0, True)
    n_ok = _compare(cfg, got, ref)
    assert n_ok > 64
    assert (ref["status"] != abi.OK).sum() >= int(wl["infeasible_mask"].sum())


def test_device_resident_path_matches_host_path(gpu_required, s64):
    cfg, opt = s64

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    wl = workloads.generate(cfg, solver, n_swarms=1, swarm_size=64, seed=2002)
    host = opt.plan(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, True, True)
    db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_f64=True, want_iters=True)
    db.launch()
    dev = db.results()
    ok = host["status"] == abi.OK
    assert np.array_equal(dev["status"], host["status"])
    assert np.array_equal(dev["ctrl"][ok], host["ctrl"][ok])
    assert np.array_equal(dev["goal"], host["goal"])
    ref = oracle_lib.plan_batch(cfg, wl["agents"], wl["row_offsets"], wl["rows"], None, 0, True)
    assert np.abs(dev["x"][ok] - ref["x"][ok]).max() < 1e-7


def test_chunked_launches_on_two_streams_equal_one_launch(gpu_required, s64):
    """launch_offset/launch_count (used to overlap the all-gather of a finished chunk with the stragglers):
    a batch permuted most-expensive-first and planned as two chunks on two streams gives the same answers
    as the plain launch of the original batch."""
    import torch
    cfg, opt = s64

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    wl = workloads.generate(cfg, solver, n_swarms=2, swarm_size=64, seed=2003)
    db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
    db.launch()
    ref = db.results()
    order = np.argsort(-ref["iters"][:, 0], kind="stable")
    pa, pro, prows = workloads.permute_batch(wl["agents"], wl["row_offsets"], wl["rows"], order)
    db2 = opt.make_device_batch(pa, pro, prows, None, 0, want_iters=True)
    n, h = len(pa), 40
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        db2.launch(offset=0, count=h)
    with torch.cuda.stream(s2):
        db2.launch(offset=h, count=n - h)
    got = db2.results()
    for k in ("status", "goal", "goal_status", "ctrl", "cost"):
        assert np.array_equal(got[k], ref[k][order]), k
    # out-of-range chunk is an argument error, not a launch
    with pytest.raises(Exception):
        db2.launch(offset=n - 1, count=2)


# ---- golden vectors (independent solver, KKT-certified) and every row family ------------------
import fixtures  # noqa: E402
import randqp  # noqa: E402


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_hip_matches_golden(gpu_required, name):
    cfg, b, ans = fixtures.load(name)
    opt = BatchTrajOptimizer(cfg)
    try:
        db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"],
                                   want_f64=True, want_iters=True)
        db.launch()
        got = db.results()
        assert fixtures.check_against_golden(cfg, b, ans, got) >= 8
        # the host-buffer entry gives the same bits
        host = opt.plan(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True, True)
        ok = got["status"] == abi.OK
        assert np.array_equal(host["status"], got["status"])
        assert np.array_equal(host["ctrl"][ok], got["ctrl"][ok])
        assert np.array_equal(host["cost"][ok], got["cost"][ok])
    finally:
        opt.close()


@pytest.mark.parametrize("name", list(randqp.CONFIGS))
def test_hip_matches_oracle_on_fresh_batches(gpu_required, name):
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    b = randqp.random_batch(cfg, 96, seed=31337, **gk)
    opt = BatchTrajOptimizer(cfg)
    try:
        db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_f64=True)
        db.launch()
        got = db.results()
    finally:
        opt.close()
    ref = oracle_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    _compare(cfg, got, ref)
    ok = ref["status"] == abi.OK
    assert ok.sum() >= 20
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() < 1e-7
    if b["max_slack_obs"]:
        assert np.abs(got["slack_cost"][ok] - ref["slack_cost"][ok]).max() < 1e-7


def test_known_answers_from_reference_log_on_gpu(gpu_required):
    """The reference's own logged run (CPLEX) pins the first replanning step; here through the HIP path."""
    from test_oracle import _kat_problem
    k = fixtures.kat()
    cfg, _ = _kat_problem(k, k["agents"][0])
    opt = BatchTrajOptimizer(cfg)
    try:
        for agent in k["agents"]:
            _, ag = _kat_problem(k, agent)
            ctrl, cost, _ = opt.solve(ag, np.zeros(0, abi.ROW_DTYPE))
            for t, key in ((0.1, "t0.1"), (0.2, "t0.2")):
                s = oracle_lib.state_at(ctrl, cfg.dt, t)
                exp = agent[key]
                assert np.allclose(s[0], exp["pos"], rtol=2e-6, atol=1e-6)
                assert np.allclose(s[1], exp["vel"], rtol=2e-5, atol=1e-6)
                assert np.allclose(s[2], exp["acc"], rtol=1e-4, atol=1e-5)
    finally:
        opt.close()


def test_single_agent_interface_and_error_convention(gpu_required, s64):
    cfg, opt = s64
    from dlsc_gc_planner_amd.planner import QPFailed
    cfg2, b, ans = fixtures.load("s64")
    bad = int(np.where(ans["qp_status"] == abi.INFEASIBLE)[0][0])
    good = int(np.where(ans["qp_status"] == abi.OK)[0][0])
    ro = b["row_offsets"]
    ag = b["agents"][good:good + 1].copy()
    goal = opt.solve_goal(ag, b["rows"][ro[good]:ro[good + 1]])
    ag["current_goal_point"][0] = goal
    ctrl, cost, slack = opt.solve(ag, b["rows"][ro[good]:ro[good + 1]])
    assert ctrl.shape == (cfg.M, cfg.n + 1, 3) and cost > 0 and slack.shape == (0,)
    ag = b["agents"][bad:bad + 1].copy()
    with pytest.raises(QPFailed):  # PlanningReport::QPFAILED, traj_optimizer.cpp:152
        opt.solve(ag, b["rows"][ro[bad]:ro[bad + 1]])


def test_empty_ragged_and_zero_agent_batches(gpu_required, s64):
    cfg, opt = s64
    b = randqp.random_batch(cfg, 9, seed=5, max_rows=40)
    keep, offs = [], [0]
    for a in range(9):
        r = b["rows"][b["row_offsets"][a]:b["row_offsets"][a + 1]].copy()
        if a == 2:
            r = r[:0]
        if a == 4:
            r["normal"] = 0
        keep.append(r)
        offs.append(offs[-1] + len(r))
    rows = np.concatenate(keep)
    ref = oracle_lib.plan_batch(cfg, b["agents"], np.array(offs), rows, None, 0, True)
    got = opt.plan(b["agents"], np.array(offs), rows, None, 0, True, True)
    _compare(cfg, got, ref)
    z = opt.plan(b["agents"][:0], np.array([0]), rows[:0], None, 0, True, True)
    assert z["status"].shape == (0,)
    # a batch with no rows at all
    none = opt.plan(b["agents"], np.zeros(10, np.int64), rows[:0], None, 0, True, True)
    ref0 = oracle_lib.plan_batch(cfg, b["agents"], np.zeros(10, np.int64), rows[:0], None, 0, True)
    _compare(cfg, none, ref0)


def test_update_param_rebuilds_the_tables(gpu_required):
    cfg, b, _ = fixtures.load("empty3d_M5")
    opt = BatchTrajOptimizer(cfg)
    try:
        cfg2 = abi.make_config(dim=3, M=5, control_input_weight=0.05, terminal_weight=2.0)
        opt.update_param(cfg2)
        got = opt.plan(b["agents"], b["row_offsets"], b["rows"], None, 0, True, True)
        ref = oracle_lib.plan_batch(cfg2, b["agents"], b["row_offsets"], b["rows"], None, 0, True)
        _compare(cfg2, got, ref)
    finally:
        opt.close()


def test_invalid_arguments_are_reported_not_thrown(gpu_required):
    from dlsc_gc_planner_amd.planner import DlscqpError
    with pytest.raises(DlscqpError, match="only n=5, phi=3"):  # traj_optimizer.cpp:209
        BatchTrajOptimizer(abi.make_config(n=4))
    with pytest.raises(DlscqpError):
        BatchTrajOptimizer(abi.make_config(dim=4))  # traj_optimizer.cpp:248


def test_full_size_batch_properties(gpu_required, s64):
    """BASELINE-size batch (256 swarms x 64 agents): size-independent properties instead of the oracle:
    feasibility of every solved trajectory w.r.t. the model rows, exact initial state, idempotence of the
    launch, and agreement of the fused and the two-call entry points."""
    cfg, opt = s64

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    wl = workloads.generate(cfg, solver, n_swarms=256, swarm_size=64, seed=2000)
    N, K = len(wl["agents"]), wl["K"]
    db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_f64=True)
    db.launch()
    r1 = db.results()
    db.launch()
    r2 = db.results()
    for k in ("status", "ctrl", "cost", "goal", "x"):
        assert np.array_equal(r1[k], r2[k]), k  # deterministic, no state carried between launches
    ok = r1["status"] == abi.OK
    assert 0.8 < ok.mean() < 1.0
    assert (r1["status"][wl["infeasible_mask"]] == abi.INFEASIBLE).all()
    dim, M, n = cfg.world_dimension, cfg.M, cfg.n
    P = M * (n + 1)
    x = r1["x"][ok].reshape(-1, dim, M, n + 1)
    ag = wl["agents"][ok]
    # initial state rows (traj_optimizer.cpp:334-352)
    assert np.abs(x[:, :, 0, 0] - ag["position"]).max() < 1e-9
    assert np.abs((x[:, :, 0, 1] - x[:, :, 0, 0]) * n / cfg.dt - ag["velocity"]).max() < 1e-7
    # C0..C2 continuity and the terminal stop (traj_optimizer.cpp:339-381, 515-524)
    assert np.abs(x[:, :, 1:, 0] - x[:, :, :-1, n]).max() < 1e-9
    assert np.abs((x[:, :, 1:, 1] - x[:, :, 1:, 0]) - (x[:, :, :-1, n] - x[:, :, :-1, n - 1])).max() < 1e-9
    assert np.abs(x[:, :, -1, n] - x[:, :, -1, n - 2]).max() < 1e-9
    # dynamic limits (traj_optimizer.cpp:453-487), skipping the rows the reference skips
    v = (x[:, :, :, 1:] - x[:, :, :, :-1]) * n / cfg.dt
    v[:, :, 0, :2] = 0
    assert np.abs(v).max() <= 1.0 + 1e-6
    acc = (x[:, :, :, 2:] - 2 * x[:, :, :, 1:-1] + x[:, :, :, :-2]) * n * (n - 1) / cfg.dt ** 2
    acc[:, :, 0, :1] = 0
    assert np.abs(acc).max() <= 2.0 + 1e-5
    # world box (traj_optimizer.cpp:251-265)
    xs = x.copy()
    xs[:, :, 0, :3] = 0
    assert (xs >= np.array(cfg.world_min[:])[None, :, None, None] - 1e-8).all()
    assert (xs <= np.array(cfg.world_max[:])[None, :, None, None] + 1e-8).all()
    # every LSC row (traj_optimizer.cpp:413-450)
    rows = wl["rows"].reshape(N, K)[ok]
    pidx = rows["m"].astype(np.int64) * (n + 1) + rows["i"]
    xf = r1["x"][ok].reshape(-1, dim, P)
    cp = np.take_along_axis(xf, pidx[:, None, :].repeat(dim, 1), axis=2)  # [n_ok, dim, K]
    nrm = rows["normal"].astype(np.float64)
    lhs = np.einsum("akd,adk->ak", nrm, cp)
    rhs = rows["d"] + np.einsum("akd,akd->ak", nrm, rows["point"].astype(np.float64))
    skip = (rows["m"] == 0) & (rows["i"] < cfg.phi)
    assert np.where(skip, 0.0, rhs - lhs).max() < 1e-7
    # float32 read-out is the rounded f64 solution (traj_optimizer.cpp:69-83)
    assert np.array_equal(r1["ctrl"][ok], x.transpose(0, 2, 3, 1).astype(np.float32))
    # oracle parity on a bounded sample of the big batch
    sel = np.arange(0, N, 97)
    ro = wl["row_offsets"]
    sub_rows = np.concatenate([wl["rows"][ro[a]:ro[a + 1]] for a in sel])
    sub_ro = np.arange(len(sel) + 1) * K
    ref = oracle_lib.plan_batch(cfg, wl["agents"][sel], sub_ro, sub_rows, None, 0, True)
    assert np.array_equal(ref["status"], r1["status"][sel])
    oks = ref["status"] == abi.OK
    assert np.abs(ref["x"][oks] - r1["x"][sel][oks]).max() < 1e-7


# ---- contract violations and verdict edge cases (round 2) ------------------------------------------------------
def test_understated_max_rows_fails_loudly_on_the_device(gpu_required):
    """An agent with more rows than dlscqp_device_batch.max_rows_per_agent declares is not planned with the surplus
    half-spaces dropped: it comes back DLSCQP_NUMERICAL; the others are untouched."""
    cfg, b, _ = fixtures.load("crowd3d_M5_400rows")
    opt = BatchTrajOptimizer(cfg)
    try:
        db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], None, 0, want_f64=True)
        db.launch()
        ref = db.results()
        k = np.diff(b["row_offsets"])
        for declared in (100, 200, 300):  # all-cached instantiation / register cache + HBM scan with a short flag array
            db.c.max_rows_per_agent = declared
            db.launch()
            got = db.results()
            over = k > declared
            assert over.any()
            assert (got["status"][over] == abi.NUMERICAL).all() and (got["goal_status"][over] == abi.NUMERICAL).all()
            assert np.array_equal(got["status"][~over], ref["status"][~over])
            ok = ~over & (ref["status"] == abi.OK)
            assert np.array_equal(got["x"][ok], ref["x"][ok])
    finally:
        opt.close()


def test_device_entry_rejects_missing_corridor_boxes(gpu_required):
    """world_use_octomap = 1 and sfc_boxes = NULL: the device entry refuses like the host entry does (it used to plan
    without the corridor rows and report OK)."""
    from dlsc_gc_planner_amd.planner import DlscqpError
    cfg, b, _ = fixtures.load("maze2d_M10_sfc_comm")
    opt = BatchTrajOptimizer(cfg)
    try:
        db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], None, 0)
        with pytest.raises(DlscqpError, match="sfc_boxes is NULL"):
            db.launch()
        with pytest.raises(DlscqpError, match="sfc_boxes is NULL"):
            opt.plan(b["agents"], b["row_offsets"], b["rows"], None, 0, True, True)
    finally:
        opt.close()


def test_phi_n_above_one_keeps_verdict_parity(gpu_required):
    """phi_n = 2, 3: Q_base carries lower-derivative terms, the kernel's weak-duality test must be off; half of the agents
    are infeasible."""
    from dlsc_gc_planner_amd.planner import DlscqpError
    with pytest.raises(DlscqpError, match="phi_n"):
        BatchTrajOptimizer(abi.make_config(phi_n=0))
    for name in ("empty3d_M5_phin2", "maze2d_M10_phin3"):
        ck, gk = randqp.CONFIGS[name]
        cfg = abi.make_config(**ck)
        b = randqp.random_batch(cfg, 128, seed=4711, **dict(gk, infeasible_frac=0.5))
        opt = BatchTrajOptimizer(cfg)
        try:
            got = opt.plan(b["agents"], b["row_offsets"], b["rows"], b["sfc"], 0, True, True)
        finally:
            opt.close()
        ref = oracle_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], 0, True)
        assert _compare(cfg, got, ref) >= 20
        assert (ref["status"] == abi.INFEASIBLE).sum() >= 30


def test_weak_duality_test_never_changes_a_verdict_on_the_device(gpu_required):
    """32 768 agents of the bench workload with the early infeasibility proof on and off (DLSCQP_DEBUG_NO_WD is read when
    the context is created): identical verdicts and solutions, fewer iterations with it."""
    import os
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    wl = workloads.generate(cfg, solver, n_swarms=512, swarm_size=64, seed=2000)
    try:
        db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_f64=True, want_iters=True)
        db.launch()
        a = db.results()
    finally:
        opt.close()
    os.environ["DLSCQP_DEBUG_NO_WD"] = "1"
    try:
        opt2 = BatchTrajOptimizer(cfg)
    finally:
        del os.environ["DLSCQP_DEBUG_NO_WD"]
    try:
        db2 = opt2.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_f64=True, want_iters=True)
        db2.launch()
        c = db2.results()
    finally:
        opt2.close()
    assert np.array_equal(a["status"], c["status"])
    ok = a["status"] == abi.OK
    assert np.array_equal(a["x"][ok], c["x"][ok])
    assert a["iters"][:, 0].max() < c["iters"][:, 0].max() and a["iters"][:, 0].sum() < c["iters"][:, 0].sum()
    assert (a["status"] == abi.INFEASIBLE).sum() > 1000


def test_tolerance_boundaries_on_the_device(gpu_required):
    """tests/edge_cases.py through the C ABI: QPs infeasible by 0.5x / 2x the assumed CPLEX feasibility tolerance, goal LPs
    with an empty interval below / above it, and the reference's own 1e-5 fallback boundary -- same side as the oracle."""
    import edge_cases as X
    for dim, M in ((3, 5), (2, 10)):
        cfg = abi.make_config(dim=dim, M=M)
        opt = BatchTrajOptimizer(cfg)
        try:
            cases = [X.opposing_rows_qp(cfg, g) for g in (0.0, 0.5e-6, 2e-6, 1e-3)]
            cases += [X.goal_lp_case(cfg, g, 0.0) for g in (0.5e-6, 2e-6)]
            cases += [X.goal_lp_fallback_case(cfg, v) for v in (0.5e-5, 2e-5)]
            agents = np.concatenate([c[0] for c in cases])
            rows = np.concatenate([c[1] for c in cases])
            ro = np.concatenate([[0], np.cumsum([len(c[1]) for c in cases])])
            got = opt.plan(agents, ro, rows, None, 0, True, True)
        finally:
            opt.close()
        ref = oracle_lib.plan_batch(cfg, agents, ro, rows, None, 0, True)
        _compare(cfg, got, ref)
        assert list(ref["status"][:4]) == [abi.OK, abi.OK, abi.INFEASIBLE, abi.INFEASIBLE]
        assert list(ref["goal_status"][4:]) == [abi.OK, abi.INFEASIBLE, abi.GOAL_FALLBACK, abi.INFEASIBLE]
