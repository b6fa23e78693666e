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
        assert np.allclose(got["cost"][ok], ref["cost"][ok], rtol=1e-7, atol=1e-9)
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
