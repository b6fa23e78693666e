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
