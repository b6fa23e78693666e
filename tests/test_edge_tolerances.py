"""CPU tier: verdicts on the tolerance boundaries that are assumptions about CPLEX (tests/edge_cases.py) -- the oracle and
the kernel's own source must take the same side; the HIP kernel is checked on the same cases under -m gpu."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import edge_cases as X
import emul_lib as E
import oracle_lib as O


def _both(cfg, ag, rows):
    ro = np.array([0, len(rows)], np.int64)
    return O.plan_batch(cfg, ag, ro, rows, None, 0, True), E.plan_batch(cfg, ag, ro, rows, None, 0, True)


@pytest.mark.parametrize("dim,M", [(3, 5), (2, 10)])
def test_qp_infeasible_by_half_and_by_twice_the_feasibility_tolerance(dim, M):
    cfg = abi.make_config(dim=dim, M=M)
    for gap, want in ((0.0, abi.OK), (0.5e-6, abi.OK), (2e-6, abi.INFEASIBLE), (1e-3, abi.INFEASIBLE)):
        ag, rows = X.opposing_rows_qp(cfg, gap)
        ref, got = _both(cfg, ag, rows)
        assert ref["status"][0] == want and got["status"][0] == want, (gap, ref["status"], got["status"])
        if want == abi.OK:
            assert np.abs(ref["x"] - got["x"]).max() < 1e-8


def test_goal_lp_empty_interval_below_and_above_the_tolerance():
    cfg = abi.make_config(dim=3, M=5)
    for gap, feasible in ((0.5e-6, True), (2e-6, False)):
        ag, rows = X.goal_lp_case(cfg, gap, 0.0)
        ref, got = _both(cfg, ag, rows)
        assert ref["goal_status"][0] == got["goal_status"][0]
        assert (ref["goal_status"][0] == abi.OK) == feasible, (gap, ref["goal_status"])
        if feasible:
            assert np.array_equal(ref["goal"], got["goal"]) and abs(float(ref["goal"][0, 0]) - 0.6) < 1e-5


def test_goal_lp_fallback_boundary_of_the_reference():
    """goal_optimizer.cpp:63-81: infeasible LP + current goal within 1e-5 of every row -> "numerical error", keep the goal;
    beyond 1e-5 -> QPFAILED."""
    cfg = abi.make_config(dim=3, M=5)
    for viol, want in ((0.5e-5, abi.GOAL_FALLBACK), (2e-5, abi.INFEASIBLE)):
        ag, rows = X.goal_lp_fallback_case(cfg, viol)
        ref, got = _both(cfg, ag, rows)
        assert ref["goal_status"][0] == want and got["goal_status"][0] == want, (viol, ref["goal_status"], got["goal_status"])
        if want == abi.GOAL_FALLBACK:
            assert np.array_equal(got["goal"][0], ag["current_goal_point"][0])
