"""Agent-QPs on the tolerance boundaries where this code base had to GUESS what CPLEX does (test infrastructure).

CPLEX 22.1.1 is not available, and the reference's one logged run never comes near these boundaries, so the two choices
below are assumptions, stated in DESIGN.md and pinned here so that oracle, kernel source and HIP kernel at least agree:
  1. a QP whose rows cannot all be met is still "solved" if the unavoidable violation is at most 1e-6 in the reference's
     own row scaling (CPLEX's default feasibility tolerance, CPX_PARAM_EPRHS) -- qp_types.hpp: CPLEX_FEAS_TOL;
  2. the goal LP: rows that cannot all be met by more than that tolerance make the LP infeasible; GoalOptimizer::solve
     then calls it a "numerical error" and keeps the current goal unless the current goal itself violates a row by more
     than 1e-5 (goal_optimizer.cpp:54-81, a rule of the reference, not a guess)."""
import numpy as np

from dlsc_gc_planner_amd import abi


def _agent(cfg, pos, goal, wp):
    ag = np.zeros(1, abi.AGENT_DTYPE)
    ag["position"][0], ag["current_goal_point"][0], ag["next_waypoint"][0] = pos, goal, wp
    ag["max_vel"], ag["max_acc"], ag["radius"], ag["nominal_velocity"] = 1.0, 2.0, 0.15, 1.0
    ag["n_obs"] = 2
    return ag


def _row(m, i, normal, point, d, obs):
    r = np.zeros(1, abi.ROW_DTYPE)
    r["normal"][0], r["point"][0], r["d"], r["m"], r["i"], r["obs"], r["slack"] = normal, point, d, m, i, obs, -1
    return r


def opposing_rows_qp(cfg, gap):
    """Two half-spaces on control point (M-2, 2): x >= 0.3 and x <= 0.3 - gap.  gap > 0: infeasible by `gap` metres."""
    M = cfg.M
    ag = _agent(cfg, [0.0, 0.0, 1.0], [0.5, 0.0, 1.0], [0.5, 0.0, 1.0])
    rows = np.concatenate([_row(M - 2, 2, [1, 0, 0], [0, 0, 0], 0.3, 0), _row(M - 2, 2, [-1, 0, 0], [0, 0, 0], -(0.3 - gap), 1)])
    return ag, rows


def goal_lp_case(cfg, gap, goal_violation):
    """Goal LP  min t, goal(t) = wp + t (cg - wp), with two rows on the last control point that bound t from both sides with
    an empty interval of width `gap` (in t), and the current goal violating the second row by `goal_violation` metres."""
    M, n = cfg.M, cfg.n
    cg, wp = np.float32([1.0, 0.0, 1.0]), np.float32([0.0, 0.0, 1.0])
    ag = _agent(cfg, [0.5, 0.0, 1.0], cg, wp)
    # row A: x >= 0.6  -> t >= 0.6 ; row B: x <= 0.6 - gap -> t <= 0.6 - gap  (x(t) = t)
    a = _row(M - 1, n, [1, 0, 0], [0, 0, 0], 0.6, 0)
    # row B as a half-space that the current goal (x = 1) violates by goal_violation: -(x - p) - d >= 0 with p + ... chosen so
    # that the t-bound is 0.6 - gap; the violation at x = 1 is 1 - (0.6 - gap) = 0.4 + gap metres, far above 1e-5, so the
    # violation knob moves the current goal instead: cg.y off the plane of a third row
    b = _row(M - 1, n, [-1, 0, 0], [0, 0, 0], -(0.6 - gap), 1)
    rows = np.concatenate([a, b])
    return ag, rows


def goal_lp_fallback_case(cfg, goal_violation):
    """An LP made infeasible by a zero-gradient row (normal orthogonal to cg - wp) violated by `goal_violation` metres along the
    whole line: the current goal violates it by the same amount, which decides fallback (<= 1e-5) or QPFAILED (> 1e-5)."""
    M, n = cfg.M, cfg.n
    cg, wp = np.float32([1.0, 0.0, 1.0]), np.float32([0.0, 0.0, 1.0])
    ag = _agent(cfg, [0.5, 0.0, 1.0], cg, wp)
    # y >= goal_violation on a line with y = 0: a = 0, b = -goal_violation
    rows = _row(M - 1, n, [0, 1, 0], [0, 0, 0], goal_violation, 0)
    return ag, rows
