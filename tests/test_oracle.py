"""CPU tier: the oracle against the reference's own log (known answers), against an independent
solver's golden vectors, and against an independent numpy restatement of the model."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import fixtures
import model_twin
import oracle_lib as O


def _kat_problem(k, agent):
    c = k["config"]
    cfg = abi.make_config(dim=c["dim"], M=c["M"], n=c["n"], dt=c["dt"], control_input_weight=c["control_input_weight"],
                          terminal_weight=c["terminal_weight"], communication_range=c["communication_range"],
                          world_min=tuple(c["world_min"]), world_max=tuple(c["world_max"]), world_z_2d=c["world_z_2d"],
                          world_use_octomap=False)
    ag = np.zeros(1, abi.AGENT_DTYPE)
    start = np.array(agent["start"], np.float32)
    # first replanning step: the grid planner's waypoint is one 0.5 m cell towards the far side of the
    # maze, and the unconstrained goal LP returns the waypoint itself
    step = c["grid_resolution"] if start[0] < 2.0 else -c["grid_resolution"]
    wp = start + np.array([step, 0, 0], np.float32)
    ag["position"][0] = start
    ag["current_goal_point"][0] = wp
    ag["next_waypoint"][0] = wp
    ag["max_vel"], ag["max_acc"], ag["radius"], ag["nominal_velocity"] = c["max_vel"], c["max_acc"], c["radius"], c["nominal_velocity"]
    return cfg, ag


def test_known_answers_from_reference_log():
    """The committed run of the reference (CPLEX) pins the first replanning step of every agent."""
    k = fixtures.kat()
    rows = np.zeros(0, abi.ROW_DTYPE)
    for agent in k["agents"]:
        cfg, ag = _kat_problem(k, agent)
        r = O.traj_solve(cfg, ag, rows)
        assert r["status"] == abi.OK
        for t, key in ((0.1, "t0.1"), (0.2, "t0.2")):
            s = O.state_at(r["ctrl"], cfg.dt, t)
            exp = agent[key]
            # the log prints 6 significant digits
            assert np.allclose(s[0], exp["pos"], rtol=2e-6, atol=1e-6), (agent["id"], t, s[0], exp["pos"])
            assert np.allclose(s[1], exp["vel"], rtol=2e-5, atol=1e-6), (agent["id"], t, s[1], exp["vel"])
            assert np.allclose(s[2], exp["acc"], rtol=1e-4, atol=1e-5), (agent["id"], t, s[2], exp["acc"])


def test_terminal_segments_of_logged_run():
    k = fixtures.kat()
    cfg, ag = _kat_problem(k, k["agents"][0])
    assert O.terminal_segments(cfg, ag) == 7  # (2.0 - 0.5 + 1e-9) / 0.2 -> 7


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_dense_model_equals_numpy_twin(name):
    cfg, b, _ = fixtures.load(name)
    for a in range(0, len(b["agents"]), 5):
        rows = b["rows"][b["row_offsets"][a]:b["row_offsets"][a + 1]]
        box = None if b["sfc"] is None else b["sfc"][a]
        m1 = O.model_dense(cfg, b["agents"][a:a + 1], rows, box)
        m2 = model_twin.traj_model(cfg, b["agents"][a:a + 1], rows, box)
        for key in ("nv", "neq", "nin", "terminal_segments"):
            assert m1[key] == m2[key], key
        for key in ("Q", "c", "Aeq", "beq", "Ain", "lo", "hi", "lb", "ub"):
            assert np.allclose(m1[key], m2[key], rtol=1e-13, atol=1e-13), key
        assert abs(m1["c0"] - m2["c0"]) < 1e-12


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_oracle_matches_independent_solver(name):
    cfg, b, ans = fixtures.load(name)
    got = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    n_ok = fixtures.check_against_golden(cfg, b, ans, got)
    assert n_ok >= 8


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_goal_lp_against_linprog(name):
    cfg, b, ans = fixtures.load(name)
    for a in range(len(b["agents"])):
        rows = b["rows"][b["row_offsets"][a]:b["row_offsets"][a + 1]]
        box = None if b["sfc"] is None else b["sfc"][a]
        g = O.goal_solve(cfg, b["agents"][a:a + 1], rows, box)
        if ans["goal_feasible"][a]:
            assert g["status"] == abi.OK
            assert abs(g["t"] - ans["goal_t"][a]) < 1e-7
            cg = b["agents"]["current_goal_point"][a]
            wp = b["agents"]["next_waypoint"][a]
            exp = (cg - wp) * np.float32(g["t"]) + wp  # goal_optimizer.cpp:51 in float32
            assert np.array_equal(g["goal"], exp.astype(np.float32))
        else:
            assert g["status"] in (abi.INFEASIBLE, abi.GOAL_FALLBACK)
            if g["status"] == abi.GOAL_FALLBACK:
                assert np.array_equal(g["goal"], b["agents"]["current_goal_point"][a])


def test_goal_lp_early_return_and_fallback():
    cfg = abi.make_config(dim=3, M=5)
    ag = np.zeros(1, abi.AGENT_DTYPE)
    ag["current_goal_point"][0] = (1, 1, 1)
    ag["next_waypoint"][0] = (1, 1, 1.000001)
    g = O.goal_solve(cfg, ag, np.zeros(0, abi.ROW_DTYPE))
    assert g["status"] == abi.OK and np.array_equal(g["goal"], ag["next_waypoint"][0])  # goal_optimizer.cpp:12-14
    # a row that both end points violate by less than 1e-5: LP infeasible, "numerical error" branch
    ag["next_waypoint"][0] = (2, 1, 1)
    row = np.zeros(1, abi.ROW_DTYPE)
    row["normal"][0] = (0, 1, 0)
    row["point"][0] = (0, 1, 0)
    row["d"] = 5e-6
    row["m"], row["i"], row["slack"] = cfg.M - 1, cfg.n, -1
    g = O.goal_solve(cfg, ag, row)
    assert g["status"] == abi.GOAL_FALLBACK and np.array_equal(g["goal"], ag["current_goal_point"][0])
    row["d"] = 1e-3  # a real violation: QPFAILED (goal_optimizer.cpp:73-76, 122)
    assert O.goal_solve(cfg, ag, row)["status"] == abi.INFEASIBLE
    row["flags"] = abi.ROW_DYNAMIC  # dynamic obstacles are not part of the LP (goal_optimizer.cpp:177-179)
    assert O.goal_solve(cfg, ag, row)["status"] == abi.OK


def test_state_sampling_is_consistent_with_initial_state_rows():
    cfg, b, _ = fixtures.load("s64")
    got = O.plan_batch(cfg, b["agents"][:8], b["row_offsets"][:9], b["rows"][:b["row_offsets"][8]], None, 0, True)
    for a in range(8):
        if got["status"][a] != abi.OK:
            continue
        s = O.state_at(got["ctrl"][a], cfg.dt, 0.0)
        assert np.allclose(s[0], b["agents"]["position"][a], atol=1e-6)
        assert np.allclose(s[1], b["agents"]["velocity"][a], atol=2e-5)
        assert np.allclose(s[2], b["agents"]["acceleration"][a], atol=1e-3)


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_oracle_solutions_carry_a_kkt_certificate(name):
    """Optimality of every oracle answer is verified without any solver (tests/kkt.py)."""
    import kkt
    cfg, b, _ = fixtures.load(name)
    got = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    n = 0
    for a in range(len(b["agents"])):
        if got["status"][a] != abi.OK:
            continue
        ag = b["agents"][a:a + 1].copy()
        ag["current_goal_point"][0] = got["goal"][a]
        rows = b["rows"][b["row_offsets"][a]:b["row_offsets"][a + 1]]
        md = model_twin.traj_model(cfg, ag, rows, None if b["sfc"] is None else b["sfc"][a])
        c = kkt.certificate(md, got["x"][a])
        assert c["primal"] < 1e-8 and c["stationarity"] < 1e-8, (a, c)
        n += 1
    assert n >= 8
