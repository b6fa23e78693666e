"""The reference's failure path (traj_optimizer.cpp:45-52, 113-146) without CPLEX: the infeasibility certificate of the kernel's
source (lane-serial build, tests/emul) over the nineteen configurations of tests/randqp.py, checked with plain numpy against the
numpy twin of the model -- no oracle, no solver (tests/conflict_check.py) --, and the model export as a CPLEX-LP file, read back
and handed to HiGHS.  The same checks run on the device in tests/test_gpu_conflict.py."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import conflict_check
import emul_lib as E
import randqp


@pytest.mark.parametrize("name", list(randqp.CONFIGS))
def test_every_infeasible_verdict_comes_with_a_certificate(name):
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    n = 160 if "M10" not in name and "M12" not in name else 60
    b = randqp.random_batch(cfg, n, seed=4242, **gk)
    plain = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
    got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_conflict=True)
    # asking for certificates switches the early proof by weak duality off: the verdicts are the same
    assert np.array_equal(got["status"], plain["status"]) and np.array_equal(got["goal_status"], plain["goal_status"])
    ro = b["row_offsets"]
    checked = loose = 0
    for a in range(n):
        rec = got["conflict"][a]
        qp_infeasible = got["status"][a] == abi.INFEASIBLE and got["goal_status"][a] != abi.INFEASIBLE
        if not qp_infeasible:
            assert rec[0] == 0  # no certificate for a solved agent, nor where the goal LP already failed
            continue
        assert rec[0] >= 1, (name, a, "INFEASIBLE without a certificate")
        rows = b["rows"][ro[a]:ro[a + 1]]
        sfc = None if b["sfc"] is None else b["sfc"][a]
        mem, n_con, gap = E.conflict_decode(cfg, b["agents"][a], rows, rec, sfc, b["max_slack_obs"], goal=got["goal"][a])
        assert len(mem) == int(rec[0]) and all(0 <= m["ref_index"] < n_con + 2 * 1000 for m in mem)
        _, _, resid = conflict_check.check_certificate(cfg, b["agents"][a], rows, sfc, mem, gap, goal=got["goal"][a], row_scale=rec[2])
        loose += resid > 1e-9
        checked += 1
    assert loose <= max(1, checked // 20)  # the residual of sum lambda a on the equalities' null space: <= 1e-9 but for a stray agent (<= 1e-6)
    if "infeasible_frac" not in gk or gk.get("infeasible_frac", 0.1) > 0:
        assert checked >= 1 or (got["status"] == abi.INFEASIBLE).sum() == 0


def test_the_exported_model_has_the_verdict_of_the_solve(tmp_path):
    """dlscqp_write_lp: one infeasible and one solved agent of a batch with corridor boxes, communication range and slack variables, as
    .lp files; their constraints go to HiGHS (scipy): infeasible / feasible as the kernel's source said.  The rows are the twin's."""
    import model_twin
    seen = set()
    for name in ("maze2d_M10_sfc_comm", "forest3d_M5_dyn"):
        ck, gk = randqp.CONFIGS[name]
        cfg = abi.make_config(**ck)
        b = randqp.random_batch(cfg, 80, seed=77, **gk)
        got = E.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
        ro = b["row_offsets"]
        for want, pick in ((abi.INFEASIBLE, "infeasible"), (abi.OK, "solved")):
            cand = [a for a in range(80) if got["status"][a] == want and got["goal_status"][a] != abi.INFEASIBLE]
            if not cand:
                continue
            a = cand[0]
            rows = b["rows"][ro[a]:ro[a + 1]]
            sfc = None if b["sfc"] is None else b["sfc"][a]
            path = tmp_path / ("%s_%s.lp" % (name, pick))
            E.write_lp(cfg, b["agents"][a], rows, path, sfc, b["max_slack_obs"], goal=got["goal"][a])
            assert conflict_check.lp_feasible(path) == (want == abi.OK), (name, pick)
            # the file's rows are the twin's rows, in the reference's order (equalities of the end of the horizon last)
            _, A, lo, hi, lb, ub = conflict_check.parse_lp(path)
            tw = model_twin.traj_model(cfg, b["agents"][a], rows, sfc)
            assert A.shape == (tw["neq"] + tw["nin"], tw["nv"])
            assert np.allclose(np.nan_to_num(lb, neginf=-1e300), tw["lb"]) and np.allclose(np.nan_to_num(ub, posinf=1e300), tw["ub"])
            eq = lo == hi
            assert int(eq.sum()) == tw["neq"]
            assert np.allclose(np.sort(A[~eq].sum(1)), np.sort(tw["Ain"].sum(1)), rtol=1e-12, atol=1e-12)
            seen.add((name, pick))
    assert ("maze2d_M10_sfc_comm", "solved") in seen and any(p == "infeasible" for _, p in seen)
