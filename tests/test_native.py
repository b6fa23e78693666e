"""CPU tier: the CPU-shaped solver behind bench.py's cpu_baseline (tests/native/native_qp.cpp) against the oracle -- identical verdicts
and sub-goals, solutions within 1e-6, costs within 1e-6 relative -- on every configuration of randqp.CONFIGS (the launch files'
shapes and the eight parameter sets off their values), plus the golden batches.  The solver is the kernel's mathematics written a
second time (scalar loops, its own data structures), so this is also an independent check of the method itself."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import native_lib
import oracle_lib
import randqp


@pytest.mark.parametrize("name", list(randqp.CONFIGS))
def test_native_solver_matches_the_oracle(name):
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    n = 60 if "M10" not in name and "M12" not in name else 30
    b = randqp.random_batch(cfg, n, seed=4711, **gk)
    ref = oracle_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    got = native_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
    assert np.array_equal(got["status"], ref["status"]), (name, np.nonzero(got["status"] != ref["status"])[0][:8])
    assert np.array_equal(got["goal_status"], ref["goal_status"])
    lp_ok = ref["goal_status"] != abi.INFEASIBLE
    assert np.array_equal(got["goal"][lp_ok], ref["goal"][lp_ok])
    ok = ref["status"] == abi.OK
    assert ok.any()
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() <= 1e-6
    assert (np.abs(got["cost"][ok] - ref["cost"][ok]) <= 1e-6 * np.maximum(1.0, np.abs(ref["cost"][ok]))).all()
    assert np.array_equal(got["ctrl"][ok], got["ctrl"][ok])  # (finite)
    assert np.abs(got["ctrl"][ok].astype(np.float64) - ref["ctrl"][ok]).max() <= 1e-5


def test_native_throughput_entry_point_runs():
    ck, gk = randqp.CONFIGS["empty3d_M5"]
    cfg = abi.make_config(**ck)
    b = randqp.random_batch(cfg, 64, seed=5, **gk)
    rate, n, el = native_lib.bench(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], threads=2, seconds=0.2)
    assert n >= 64 and rate > 0 and el >= 0.2
