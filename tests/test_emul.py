"""CPU tier: the solver kernel's own source (gi_core.hpp + host tables), built lane-serially for the
host (tests/emul), against the golden vectors and the oracle.  This is how the device algorithm is
checked in a container without a GPU; the same comparisons run against the real kernel under -m gpu."""
import os

import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import emul_lib as E
import fixtures
import oracle_lib as O
import randqp


@pytest.mark.parametrize("name", fixtures.NAMES)
def test_kernel_source_matches_golden(name):
    cfg, b, ans = fixtures.load(name)
    got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    assert fixtures.check_against_golden(cfg, b, ans, got) >= 8


@pytest.mark.parametrize("name", list(randqp.CONFIGS))
def test_kernel_source_matches_oracle_on_fresh_batches(name):
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    b = randqp.random_batch(cfg, 30, seed=99, **gk)
    ref = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    assert np.array_equal(got["goal_status"], ref["goal_status"])
    gok = ref["goal_status"] != abi.INFEASIBLE
    assert np.array_equal(got["goal"][gok], ref["goal"][gok])
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["status"] == abi.OK
    assert ok.sum() >= 5
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() < (1e-8 if cfg.control_input_weight <= 0.1 else 5e-8)  # (rows count as met within 1e-9 m)
    # (the objective's gradient grows with control_input_weight: 1e-7 relative at the launch files' 0.01, 1e-6 at the code default 1)
    assert np.allclose(got["cost"][ok], ref["cost"][ok], rtol=1e-7 if cfg.control_input_weight <= 0.1 else 1e-6, atol=1e-9)
    if b["max_slack_obs"]:
        assert np.abs(got["slack_cost"][ok] - ref["slack_cost"][ok]).max() < 1e-8


def test_lane_loops_have_no_cross_lane_hazard():
    """Running every lane loop backwards must not change a single bit."""
    cfg, b, _ = fixtures.load("forest3d_M5_dyn")
    fwd = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    os.environ["EMUL_REVERSE"] = "1"
    try:
        rev = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    finally:
        del os.environ["EMUL_REVERSE"]
    for k in ("status", "goal", "x", "cost", "iters"):
        assert np.array_equal(fwd[k], rev[k]), k


@pytest.mark.parametrize("name", ["forest3d_M10_dyn4", "forest3d_M10_dyn"])
def test_runtime_shape_instantiation_equals_the_static_one(name):
    """The runtime-shape instantiation (layout and loop bounds from the arguments, every structured row from the slot table)
    against the static-shape one that select_instance picks for the forest missions (bound / velocity / acceleration rows by
    coordinate): same verdicts, the same optimum to rounding (the pivot scores carry different row codes in their low bits, so
    the two may add the rows of a tie in another order)."""
    cfg, b, ans = fixtures.load(name)
    ref = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    os.environ["EMUL_GENERIC"] = "1"
    try:
        got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    finally:
        del os.environ["EMUL_GENERIC"]
    assert (ref["iters"][:, 0] > 3).sum() >= 3
    for k in ("status", "goal_status", "goal"):
        assert np.array_equal(ref[k], got[k]), k
    ok = ref["status"] == 0
    assert np.abs(ref["x"][ok] - got["x"][ok]).max() <= 1e-9
    assert np.allclose(ref["cost"][ok], got["cost"][ok], rtol=1e-9, atol=1e-12)
    assert fixtures.check_against_golden(cfg, b, ans, got) >= 8


def test_goal_only_and_traj_only_entry_points_agree_with_fused():
    cfg, b, _ = fixtures.load("s64")
    fused = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], None, 0, True, True)
    g = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], None, 0, True, False)
    assert np.array_equal(g["goal"], fused["goal"]) and np.array_equal(g["goal_status"], fused["goal_status"])
    ag2 = b["agents"].copy()
    gok = fused["goal_status"] != abi.INFEASIBLE
    ag2["current_goal_point"][gok] = fused["goal"][gok]
    t = E.plan_batch(cfg, ag2, b["row_offsets"], b["rows"], None, 0, False, True)
    assert np.array_equal(t["x"][gok], fused["x"][gok])


def test_host_tables_dimensions():
    # free Bernstein degrees of freedom per axis after the equalities: 3M - 2 (stop rows) / 3M
    assert E.table_info(abi.make_config(dim=3, M=5))["f"] == 13
    assert E.table_info(abi.make_config(dim=2, M=10))["f"] == 28
    assert E.table_info(abi.make_config(dim=2, M=4, planner_mode=abi.PLANNER_BVC))["f"] == 12
    i = E.table_info(abi.make_config(dim=3, M=5), 0, 120)
    assert i["n_slots"] == 81 + 69 + 57 and i["lds_bytes"] < 48 * 1024  # every column record in the working set (no spill record)
    with pytest.raises(ValueError):
        E.table_info(abi.make_config(dim=3, M=5, n=4))  # only n=5, phi=3 (traj_optimizer.cpp:209)


def test_empty_and_ragged_inputs():
    cfg = abi.make_config(dim=3, M=5)
    b = randqp.random_batch(cfg, 6, seed=5, max_rows=40)
    # agent 2 gets no rows at all, agent 4 only rows that must be ignored
    keep = []
    offs = [0]
    for a in range(6):
        r = b["rows"][b["row_offsets"][a]:b["row_offsets"][a + 1]].copy()
        if a == 2:
            r = r[:0]
        if a == 4:
            r["normal"] = 0
        keep.append(r)
        offs.append(offs[-1] + len(r))
    rows = np.concatenate(keep)
    ref = O.plan_batch(cfg, b["agents"], np.array(offs), rows, None, 0, True)
    got = E.plan_batch(cfg, b["agents"], np.array(offs), rows, None, 0, True)
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["status"] == abi.OK
    assert ok[2] and ok[4]
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() < 1e-8
    # zero agents
    z = E.plan_batch(cfg, b["agents"][:0], np.array([0]), rows[:0], None, 0, True)
    assert z["status"].shape == (0,)


def test_understated_max_rows_fails_loudly_instead_of_dropping_rows():
    """dlscqp_device_batch.max_rows_per_agent picks the instantiation and sizes the flag array; an agent with more rows
    than declared must come back DLSCQP_NUMERICAL, never OK with part of its half-spaces ignored."""
    cfg, b, _ = fixtures.load("crowd3d_M5_400rows")
    k = np.diff(b["row_offsets"])
    os.environ["EMUL_KMAX"] = "100"  # -> the all-cached instantiation (128 rows in registers)
    try:
        got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], None, 0, True)
    finally:
        del os.environ["EMUL_KMAX"]
    ref = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], None, 0, True)
    over = k > 100
    assert over.any() and (~over).any()
    assert (got["status"][over] == abi.NUMERICAL).all() and (got["goal_status"][over] == abi.NUMERICAL).all()
    assert np.array_equal(got["status"][~over], ref["status"][~over])
    ok = ~over & (ref["status"] == abi.OK)
    assert np.array_equal(got["x"][ok], ref["x"][ok])


def _with_and_without_weak_duality(cfg, b):
    with_cut = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    os.environ["DLSCQP_DEBUG_NO_WD"] = "1"
    try:
        without = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    finally:
        del os.environ["DLSCQP_DEBUG_NO_WD"]
    assert np.array_equal(with_cut["status"], without["status"])
    ok = with_cut["status"] == abi.OK
    assert np.array_equal(with_cut["x"][ok], without["x"][ok])
    return with_cut, without


@pytest.mark.parametrize("name", ["empty3d_M5", "maze2d_M10_sfc_comm", "crowd3d_M5_400rows"])
def test_weak_duality_test_never_changes_a_verdict(name):
    """The kernel's early infeasibility proof (f(x) above an upper bound of the objective) compiled in and out: same
    verdicts, same solutions, on batches with many infeasible agents."""
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    b = randqp.random_batch(cfg, 40, seed=321, **dict(gk, infeasible_frac=0.5))
    with_cut, _ = _with_and_without_weak_duality(cfg, b)
    assert (with_cut["status"] == abi.INFEASIBLE).sum() >= 10


def test_weak_duality_test_cuts_iterations_of_the_bench_workload_only():
    """3 072 agents of the bench workload (BASELINE.json configs[1]): the test fires (fewer iterations, lower maximum) and
    every verdict and solution stays what it is without it."""
    from dlsc_gc_planner_amd import workloads
    cfg = workloads.s64_config()
    wl = workloads.generate(cfg, lambda a, ro, r, s: E.plan_batch(cfg, a, ro, r, s, 0, True), n_swarms=48, swarm_size=64, seed=2000)
    b = dict(agents=wl["agents"], row_offsets=wl["row_offsets"], rows=wl["rows"], sfc=None, max_slack_obs=0)
    with_cut, without = _with_and_without_weak_duality(cfg, b)
    assert with_cut["iters"][:, 0].sum() < without["iters"][:, 0].sum()
    assert with_cut["iters"][:, 0].max() < without["iters"][:, 0].max()
    assert (with_cut["status"] == abi.INFEASIBLE).sum() >= 100


def test_phi_n_is_validated_and_switches_the_weak_duality_test_off():
    with pytest.raises(ValueError):
        E.table_info(abi.make_config(dim=3, M=5, phi_n=0))
    with pytest.raises(ValueError):
        E.table_info(abi.make_config(dim=3, M=5, phi_n=4))
    # with phi_n = 2 the jerk-only objective bound is not an upper bound: the infeasible agents take the long road
    ck, gk = randqp.CONFIGS["empty3d_M5_phin2"]
    cfg = abi.make_config(**ck)
    b = randqp.random_batch(cfg, 30, seed=77, **dict(gk, infeasible_frac=0.5))
    ref = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], None, 0, True)
    got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], None, 0, True)
    assert np.array_equal(got["status"], ref["status"]) and (ref["status"] == abi.INFEASIBLE).sum() >= 8


@pytest.mark.parametrize("name", ["crowd3d_M5_400rows", "forest3d_M10_dyn", "long3d_M12_comm"])
def test_rows_in_the_working_set_change_nothing(name):
    """Round 4: the rows right behind the register cache are decoded once into the working set (LDS on the device) and priced from
    there in every pass instead of from HBM (gi_core.hpp: "rows in LDS").  The same arithmetic per row: with the tier switched off
    (EMUL_LDS_BUDGET=0) the kernel's source gives the same bits -- verdicts, solutions, iteration counts."""
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    gk = dict(gk, max_rows=max(gk.get("max_rows", 60), 420), n_static_max=22, keep=0.8) if name != "crowd3d_M5_400rows" else gk
    b = randqp.random_batch(cfg, 40, seed=123, **gk)
    res = {}
    for budget in ("0", "40448", "9000"):
        os.environ["EMUL_LDS_BUDGET"] = budget
        try:
            res[budget] = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
        finally:
            del os.environ["EMUL_LDS_BUDGET"]
    assert np.diff(b["row_offsets"]).max() > 300 and res["0"]["iters"][:, 0].max() >= 2
    for k in ("status", "goal_status", "goal", "x", "cost", "iters"):
        assert np.array_equal(res["0"][k], res["40448"][k]), k
        assert np.array_equal(res["0"][k], res["9000"][k]), k  # a budget that holds only some of the rows: three tiers at once
