"""CPU tier: the upper bound of the objective that the solver uses to prove infeasibility by weak duality
(gi_core.hpp, objective_upper_bound) really bounds the optimum of every feasible QP.

Three layers: (1) c' Q_base c is the integral of the squared jerk of a segment, (2) acceleration control points
bounded by a  =>  integral jerk^2 <= 36 a^2 / dt, (3) the assembled bound, restated here in numpy, is above the
oracle's optimal cost for every feasible agent of every golden batch (the oracle does not use the bound)."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import fixtures
import model_twin
import oracle_lib as O


def _bernstein(c, s):
    """Value at s in [0,1] of the Bernstein polynomial with control points c."""
    c = np.array(c, float)
    while len(c) > 1:
        c = (1 - s) * c[:-1] + s * c[1:]
    return c[0]


def _jerk_ctrl(c, dt):
    n = len(c) - 1
    v = n / dt * np.diff(c)
    a = (n - 1) / dt * np.diff(v)
    j = (n - 2) / dt * np.diff(a)
    return a, j


def test_q_base_is_the_jerk_integral():
    rng = np.random.default_rng(1)
    n, dt = 5, 0.2
    Qb = model_twin.q_base(n, 3, 1, dt)
    s, wq = np.polynomial.legendre.leggauss(8)
    s, wq = 0.5 * (s + 1), 0.5 * wq
    for _ in range(20):
        c = rng.normal(size=n + 1)
        _, j = _jerk_ctrl(c, dt)
        integral = dt * sum(w * _bernstein(j, t) ** 2 for t, w in zip(s, wq))
        assert np.isclose(c @ Qb @ c, integral, rtol=1e-9)


def test_jerk_integral_bound_from_acceleration_bounds():
    rng = np.random.default_rng(2)
    n, dt, amax = 5, 0.2, 2.0
    Qb = model_twin.q_base(n, 3, 1, dt)
    worst = 0.0
    for _ in range(2000):
        # random acceleration control points at their bounds (extreme points), integrate back to positions
        a = rng.choice([-amax, amax], size=n - 1)
        v = rng.normal() + np.concatenate([[0.0], np.cumsum(a) * dt / (n - 1)])
        c = rng.normal() + np.concatenate([[0.0], np.cumsum(v) * dt / n])
        a2, _ = _jerk_ctrl(c, dt)
        assert np.allclose(a2, a)
        worst = max(worst, c @ Qb @ c)
    assert worst <= 36 * amax * amax / dt * (1 + 1e-12)
    assert worst > 0.2 * 36 * amax * amax / dt  # and the bound is not absurdly loose


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_bound_is_above_every_feasible_optimum(name):
    cfg, b, ans = fixtures.load(name)
    if b["max_slack_obs"] > 0:
        pytest.skip("bound is only used without slack variables")
    dim, M = cfg.world_dimension, cfg.M
    res = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], 0, True)
    n_checked = 0
    for a in range(len(b["agents"])):
        if res["status"][a] != abi.OK:
            continue
        ag = b["agents"][a]
        goal = res["goal"][a].astype(float)  # the QP's goal is the goal LP's result
        one = b["agents"][a:a + 1].copy()
        one["current_goal_point"][0] = res["goal"][a]
        T = O.terminal_segments(cfg, one)
        amax = max(float(ag["max_acc"]), float(np.abs(ag["acceleration"]).max()))
        f_max = cfg.control_input_weight * 36.0 * amax * amax / cfg.dt * dim * M
        for k in range(dim):
            for m in range(M - T, M):
                lo, hi = float(cfg.world_min[k]), float(cfg.world_max[k])
                if cfg.planner_mode == abi.PLANNER_RECIPROCALRSFC and k == 2 and m == 0:
                    lo, hi = -100.0, 100.0
                if cfg.world_use_octomap and b["sfc"] is not None:
                    lo, hi = max(lo, float(b["sfc"][a, m, k])), min(hi, float(b["sfc"][a, m, 3 + k]))
                dbox = max(abs(lo - goal[k]), abs(hi - goal[k]))
                v0 = abs(float(ag["velocity"][k]))
                v1 = abs(float(ag["velocity"][k]) + float(ag["acceleration"][k]) * cfg.dt / (cfg.n - 1))
                dreach = abs(float(ag["position"][k]) - goal[k]) + max(float(ag["max_vel"]), v0, v1) * (m + 1) * cfg.dt
                f_max += cfg.terminal_weight * min(dbox, dreach) ** 2
        assert res["cost"][a] <= f_max, (a, res["cost"][a], f_max)
        n_checked += 1
    assert n_checked >= 5
