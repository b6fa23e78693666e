"""-m gpu: the infeasibility certificate of the HIP kernel (dlscqp_device_batch.out_conflict) for EVERY infeasible verdict of the
nineteen configurations of tests/randqp.py, through the C ABI, checked with plain numpy against the numpy twin of the reference's
model -- no oracle, no solver (tests/conflict_check.py) -- and the model export (dlscqp_write_lp) handed to HiGHS.
The reference does this with CPLEX's conflict refiner and exportModel (traj_optimizer.cpp:45-52, 113-146)."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import conflict_check
import randqp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(randqp.CONFIGS))
def test_every_infeasible_verdict_of_the_device_comes_with_a_certificate(gpu_required, name):
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    n = 400 if "M10" not in name and "M12" not in name else 150
    b = randqp.random_batch(cfg, n, seed=9191, **gk)
    opt = BatchTrajOptimizer(cfg)
    try:
        plain = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
        plain.launch(True, True)
        want = plain.results()
        db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
        db.want_conflict()
        db.launch(True, True)
        got = db.results()
        rec_all = db.conflicts()
        # asking for certificates switches the early proof by weak duality off: same verdicts, same plans
        assert np.array_equal(got["status"], want["status"]) and np.array_equal(got["goal_status"], want["goal_status"])
        ok = got["status"] == abi.OK
        assert np.array_equal(got["ctrl"][ok], want["ctrl"][ok])
        ro = b["row_offsets"]
        checked = loose = 0
        for a in range(n):
            rec = rec_all[a]
            if not (got["status"][a] == abi.INFEASIBLE and got["goal_status"][a] != abi.INFEASIBLE):
                assert rec[0] == 0
                continue
            assert rec[0] >= 1, (name, a, "INFEASIBLE without a certificate")
            rows = b["rows"][ro[a]:ro[a + 1]]
            sfc = None if b["sfc"] is None else b["sfc"][a]
            mem, n_con, gap = opt.conflict_decode(b["agents"][a], rows, rec, sfc, b["max_slack_obs"], goal=got["goal"][a])
            assert len(mem) == int(rec[0])
            _, _, resid = conflict_check.check_certificate(cfg, b["agents"][a], rows, sfc, mem, gap, goal=got["goal"][a], row_scale=rec[2])
            loose += resid > 1e-9
            checked += 1
        assert checked == int(((got["status"] == abi.INFEASIBLE) & (got["goal_status"] != abi.INFEASIBLE)).sum())
        assert loose <= max(1, checked // 20)
    finally:
        opt.close()


def test_the_exported_model_of_the_device_context(gpu_required, tmp_path):
    """dlscqp_write_lp through the C ABI: an infeasible and a solved agent's constraints, read back, go to HiGHS."""
    ck, gk = randqp.CONFIGS["maze2d_M10_sfc_comm"]
    cfg = abi.make_config(**ck)
    b = randqp.random_batch(cfg, 80, seed=77, **gk)
    opt = BatchTrajOptimizer(cfg)
    try:
        got = opt.plan(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True, True)
        ro = b["row_offsets"]
        done = 0
        for want in (abi.INFEASIBLE, abi.OK):
            a = next(a for a in range(80) if got["status"][a] == want and got["goal_status"][a] != abi.INFEASIBLE)
            path = tmp_path / ("agent_%d.lp" % a)
            opt.write_lp(b["agents"][a], b["rows"][ro[a]:ro[a + 1]], path, None if b["sfc"] is None else b["sfc"][a], b["max_slack_obs"], goal=got["goal"][a])
            assert conflict_check.lp_feasible(path) == (want == abi.OK)
            done += 1
        assert done == 2
    finally:
        opt.close()
