"""Regression pins of the one parity defect the randomised sweep found in the range-space kernel (DESIGN.md section 2,
"Numerics"): a row that DEPENDS on the active rows (modulo the equality constraints) must be recognised as such -- taken
for an independent one it gets a step of 1e13 x rounding noise and an infeasible QP comes back as solved.  The batches
below are the ones of tests/fuzz_gpu.py on which the first version of the kernel accepted an infeasible problem (agents
17; 268 and 498; 145; 117; 181; 102); verdicts must equal the oracle's for every agent of them.  CPU tier: the kernel's source built
lane-serially; -m gpu: the HIP kernel through the C ABI."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import emul_lib
import oracle_lib
import randqp

CASES = [("lsc2d_M4_nostop", 1000, 600, [17]), ("lsc2d_M4_nostop", 1001, 600, [268, 498]), ("forest3d_M10_dyn", 1001, 250, [145]),
         ("forest3d_M10_dyn4", 1002, 250, [117]), ("long3d_M12_comm", 1002, 250, [181]),
         # ... and one where the row WAS measured in the whitened space and still passed for independent (|z|^2 = 1e-18 |N'a|^2: above
         # the fixed threshold, but rounding noise of its own measure): the iterate blew up to 1e23 and the agent cycled through 2 241
         # iterations to the iteration limit
         ("forest3d_M10_dyn", 1601, 250, [102])]


def _batch(name, seed, n):
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    return cfg, randqp.random_batch(cfg, n, seed=seed, **gk)


def _check(cfg, b, got, once_wrong):
    ref = oracle_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    assert (ref["status"][once_wrong] == abi.INFEASIBLE).all()  # the problems the first kernel called solved
    assert np.array_equal(got["status"], ref["status"]) and np.array_equal(got["goal_status"], ref["goal_status"])
    ok = ref["status"] == abi.OK
    assert np.abs(got["x"][ok] - ref["x"][ok]).max() < 1e-8
    assert got["iters"][:, 0].max() <= 400  # nobody cycles (the reduced dimension is at most 124)


@pytest.mark.parametrize("name,seed,n,once_wrong", CASES)
def test_kernel_source_recognises_dependent_rows(name, seed, n, once_wrong):
    cfg, b = _batch(name, seed, n)
    got = emul_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
    _check(cfg, b, got, once_wrong)


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed,n,once_wrong", CASES)
def test_hip_kernel_recognises_dependent_rows(gpu_required, name, seed, n, once_wrong):
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    cfg, b = _batch(name, seed, n)
    opt = BatchTrajOptimizer(cfg)
    try:
        db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_f64=True, want_iters=True)
        db.launch()
        got = db.results()
    finally:
        opt.close()
    _check(cfg, b, got, once_wrong)
