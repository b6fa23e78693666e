"""The dense step direction of the plan kernel (gi_core.hpp: dense_dx -- the weights of the active velocity / acceleration rows by
coordinate in a table, u gathered per control point, dx = G u) normally starts at eight such rows, which few agents of the
fixtures reach.  DLSCQP_DENSE_MIN=1 (read when a context builds its tables) sends every iteration with one such row down that
path: the answers must be the golden ones, the same as the sparse path's to rounding -- and not bit for bit the same, or
the path was not taken.  CPU tier: the kernel's source built lane-serially; -m gpu: the HIP kernel through the C ABI, also on
the 1 710 known answers of the reference's logged run."""
import os

import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import emul_lib
import fixtures

SHAPES = ["s64", "empty3d_M5", "crowd3d_M5_400rows", "maze2d_M10_sfc_comm"]  # the static shapes without slack variables
NO_SUCH_ROWS = {"crowd3d_M5_400rows"}  # a standing crowd: no velocity / acceleration row is ever active (the path must still be harmless)


class dense_from_one_row:
    def __enter__(self):
        os.environ["DLSCQP_DENSE_MIN"] = "1"

    def __exit__(self, *a):
        del os.environ["DLSCQP_DENSE_MIN"]


def _compare(cfg, b, ans, sparse, dense):
    assert fixtures.check_against_golden(cfg, b, ans, dense) >= 1
    for k in ("status", "goal_status", "goal"):
        assert np.array_equal(sparse[k], dense[k]), k
    ok = sparse["status"] == abi.OK
    assert np.abs(sparse["x"][ok] - dense["x"][ok]).max() < 1e-9
    return int((sparse["x"][ok] != dense["x"][ok]).any(axis=1).sum())


@pytest.mark.parametrize("name", SHAPES)
def test_kernel_source_dense_path_gives_the_golden_answers(name):
    cfg, b, ans = fixtures.load(name)
    args = (cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
    sparse = emul_lib.plan_batch(*args)
    with dense_from_one_row():
        dense = emul_lib.plan_batch(*args)
    differ = _compare(cfg, b, ans, sparse, dense)
    assert differ >= 1 or name in NO_SUCH_ROWS, "no agent took the dense path"


@pytest.mark.gpu
@pytest.mark.parametrize("name", SHAPES)
def test_hip_dense_path_gives_the_golden_answers(gpu_required, name):
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    cfg, b, ans = fixtures.load(name)
    res = []
    for forced in (False, True):
        ctx = dense_from_one_row() if forced else None
        if ctx:
            ctx.__enter__()
        try:
            opt = BatchTrajOptimizer(cfg)
        finally:
            if ctx:
                ctx.__exit__()
        try:
            db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_f64=True, want_iters=True)
            db.launch()
            res.append(db.results())
        finally:
            opt.close()
    differ = _compare(cfg, b, ans, res[0], res[1])
    assert differ >= 1 or name in NO_SUCH_ROWS, "no agent took the dense path"


@pytest.mark.gpu
def test_hip_dense_path_on_the_logged_run(gpu_required):
    """2-D, M = 10 (the Gram table read from L2, sixty control points): the 1 710 known answers with the dense path from the
    first active velocity / acceleration row on -- the same 27 misses by name, the same plans as the oracle's."""
    import log_replay as R
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    from test_log_replay import assert_the_known_misses
    out = R.run()
    with dense_from_one_row():
        opt = BatchTrajOptimizer(out["cfg"])
    try:
        got = opt.plan(out["agents"], out["row_offsets"], out["rows"], out["sfc"], 0, True, True)
    finally:
        opt.close()
    assert (got["status"] == abi.OK).all()
    sc = R.score_batch(out, got["ctrl"], got["status"])
    assert (sc <= 1.0).sum() >= 1680 and sc.max() <= 2.0
    assert_the_known_misses(sc)
    ref_ctrl = np.stack([s["ctrl"] for s in out["sols"]])
    assert np.abs(got["ctrl"].astype(np.float64) - ref_ctrl).max() <= 5e-7
