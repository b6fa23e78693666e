"""-m gpu: the HIP path against the reference's own CPLEX run -- 1 710 known answers (tests/test_log_replay.py explains the
fixture).  The oracle re-enacts the logged run on the host to reconstruct the inputs of every goal LP and trajectory QP;
the whole run then goes through the C ABI as ONE batch of 1 710 agent-QPs (2-D, M = 10, corridor boxes, communication
range, up to nine neighbours each) and the plans are compared with the numbers the reference printed."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import log_replay as R

pytestmark = pytest.mark.gpu


def test_hip_path_reproduces_the_logged_cplex_run(gpu_required):
    out = R.run()
    cfg = out["cfg"]
    opt = BatchTrajOptimizer(cfg)
    try:
        got = opt.plan(out["agents"], out["row_offsets"], out["rows"], out["sfc"], 0, True, True)
        db = opt.make_device_batch(out["agents"], out["row_offsets"], out["rows"], out["sfc"], 0, want_f64=True)
        db.launch()
        dev = db.results()
    finally:
        opt.close()
    # verdicts: the reference's run solved all 1 710 QPs (a QPFAILED would have ended it, multi_sync_simulator.cpp:519-523)
    assert (got["status"] == abi.OK).all() and (got["goal_status"] != abi.INFEASIBLE).all()
    sc = R.score_batch(out, got["ctrl"], got["status"])
    ok = sc <= 1.0
    print("HIP vs the reference's log: %d of %d (agent, step) pairs within the log's precision, worst %.2f" % (ok.sum(), ok.size, sc.max()))
    assert ok.sum() >= 1680 and sc.max() <= 2.0
    from test_log_replay import assert_the_known_misses
    assert_the_known_misses(sc)  # the same 27 answers as the oracle and the kernel source, by name
    act = out["kat"]["active"]
    assert int((ok & (act[:, :, 0] > 0)).sum()) >= 40      # velocity limit active
    assert int((ok & (act[:, :, 1] > 0)).sum()) >= 10      # acceleration limit active
    assert int((ok & (act[:, :, 4] > 0)).sum()) >= 200     # corridor / world box active
    assert int((ok & (act[:, :, 5] > 0)).sum()) >= 400     # DLSC-GC half-space active
    # and against the oracle on the same inputs: sub-goals bit-exact, control points to float32 rounding
    ref_goal = np.stack([s["goal"] for s in out["sols"]])
    assert np.array_equal(got["goal"], ref_goal)
    ref_ctrl = np.stack([s["ctrl"] for s in out["sols"]])
    assert np.abs(got["ctrl"].astype(np.float64) - ref_ctrl).max() <= 5e-7
    ref_x = np.stack([s["x"] for s in out["sols"]])
    assert np.abs(dev["x"] - ref_x).max() < 1e-8
    assert np.array_equal(dev["ctrl"], got["ctrl"])


def test_device_closed_loop_reproduces_the_logged_run(gpu_required, tmp_path):
    """The whole pipeline from the mission file alone, on the device: waypoints from the host planner (row N4), half-spaces
    (N1), corridor boxes (N2), goal LP + QP, state update -- ten agents, 171 steps.  Every waypoint equals the one the log
    pins, the states stay on the reference's log (drift: accumulated last-digit differences between CPLEX's and this
    solver's answers), the mission ends when the reference's did (total_flight_time 34 s in log/summary_DLSCGC_10agents.csv),
    and the result file holds the log's numbers."""
    from dlsc_gc_planner_amd.harness import ClosedLoop
    kat, w = R.load_kat(), R.load_world()
    cfg = R.make_cfg()
    log = kat["log"]
    world = dict(boxes=w["boxes"], resolution=w["world_resolution"], grid_resolution=w["grid_resolution"], pibt="reference")
    path = tmp_path / "result_maze10.csv"
    goal = np.array(w["goal"], np.float32)
    sim = ClosedLoop(cfg, np.array(w["start"], np.float32), goal, radius=R.QUAD["radius"], downwash=R.QUAD["downwash"],
                     max_vel=R.QUAD["max_vel"], max_acc=R.QUAD["max_acc"], nominal_velocity=R.QUAD["nominal_velocity"],
                     waypoint_step=w["grid_resolution"], world=world, result_csv=str(path))
    K, N = len(kat["waypoints"]), sim.N
    score = np.zeros((K, N))
    done, fails, wp_diff = None, 0, 0
    try:
        for k in range(K):
            status, _ = sim.step()
            fails += int((status != abi.OK).sum())
            wp_diff += int((np.abs(sim.wp.cpu().numpy() - kat["waypoints"][k]).max(axis=1) > 1e-6).sum())
            plans = sim.plans.cpu().numpy()
            for i in range(N):
                e1, e2 = R.log_states(log, k, i)
                score[k, i] = R.mismatch(plans[i], cfg.dt, e1, e2)
            p = sim.pos.cpu().numpy()
            if done is None and (np.linalg.norm(p - goal, axis=1) <= 0.1).all():  # isFinished, multi_sync_simulator.cpp:563-585
                done = (k + 1) * cfg.dt
    finally:
        sim.close()
    print("device closed loop vs the reference's log: waypoint differences %d, worst score %.2f, %d of %d within print precision, "
          "mission complete at %.1f s" % (wp_diff, score.max(), (score <= 1).sum(), score.size, done or -1))
    assert fails == 0 and wp_diff == 0
    assert score.max() < 10.0 and (score[:10] <= 1.0).all()  # the first ten replans of all ten agents at the log's precision
    assert (score <= 1.0).sum() >= 800
    assert done is not None and abs(done - 34.0) < 1e-9  # total_flight_time of the reference's run
    # the result file against the log, number by number (both print six significant digits)
    lines = open(path).read().strip().split("\n")
    rows = np.array([np.array(l.split(","), float).reshape(N, 12) for l in lines[1:1 + 2 * 20]])
    assert np.allclose(rows[:, :, 1], (0.1 * np.arange(40))[:, None], atol=1e-9)
    assert np.abs(rows[:, :, 2:5] - log[:40, :, 0:3]).max() < 2e-5
    assert np.abs(rows[:, :, 5:8] - log[:40, :, 3:6]).max() < 2e-4
