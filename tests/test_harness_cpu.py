"""CPU tier, row N3: the pieces of the closed-loop harness that need no GPU -- state sampling of a plan
(Trajectory::getStateAt) against the oracle's restatement, and the result file layout of the reference."""
import numpy as np

from dlsc_gc_planner_amd.harness import ResultCsv, plan_state_at
import fixtures
import oracle_lib as O


def test_plan_state_sampling_equals_oracle_state_at():
    rng = np.random.default_rng(0)
    plans = rng.normal(size=(4, 5, 6, 3)).astype(np.float32)
    for tau in (0.0, 0.05, 0.1, 0.15):
        p, v, a = plan_state_at(plans, 5, 0.2, tau)
        for q in range(4):
            s = O.state_at(plans[q], 0.2, tau)
            assert np.array_equal(p[q], s[0]) and np.array_equal(v[q], s[1]) and np.array_equal(a[q], s[2])


def test_result_csv_layout_is_the_reference_layout(tmp_path):
    k = fixtures.kat()
    path = tmp_path / "result.csv"
    plans = np.zeros((2, 5, 6, 3), np.float32)
    plans[0, :, :, 0] = -1.0
    plans[1, :, :, 0] = 5.0
    plans[:, :, :, 2] = 1.0
    w = ResultCsv(str(path), 2, n_obs=1)
    w.write_step(0.0, plans, 5, 0.2, 0.0123, lambda tau: (np.float32([[1, 2, 3]]) + np.float32(tau), [0.25]))
    w.close()
    lines = open(path).read().strip().split("\n")
    assert lines[0] == ",".join([k["columns"].split(" per agent")[0]] * 2 + ["obs_id,t,px,py,pz,size"])
    assert len(lines) == 3  # future_time 0 and 0.1 of one 0.2 s step (multi_sync_simulator.cpp:772-773)
    assert lines[1] == "0,0,-1,0,1,0,0,0,0,0,0,0.0123,1,0,5,0,1,0,0,0,0,0,0,0.0123,0,0,1,2,3,0.25"
    assert lines[2].startswith("0,0.1,-1,0,1,0,0,0,0,0,0,0.0123,1,0.1,5,0,1,") and lines[2].endswith(",0,0.1,1.1,2.1,3.1,0.25")


def test_spin_obstacle_model():
    """SpinObstacle::getSpinObstacle (obstacle.hpp:113-149): circle around the axis through the start point, speed along
    the tangent; checked against the closed form for a vertical axis and by invariants for a tilted one."""
    from dlsc_gc_planner_amd.harness import SpinObstacles
    obs = [dict(type="spin", axis_position=[0.0, 0.0, 1.0], axis_ori=[0, 0, 1], start=[2.0, 0.0, 1.0], speed=1.0, size=0.3, max_acc=2.0, downwash=1.0),
           dict(type="spin", axis_position=[1.0, -1.0, 0.5], axis_ori=[1, 2, 2], start=[3.0, 0.0, -1.5], speed=0.7, size=0.2, max_acc=1.0, downwash=0)]
    s = SpinObstacles(obs)
    assert s.downwash[1] == 1.0  # mission.cpp:237: downwash 0 means 1
    for t in (0.0, 0.3, 2.0, 11.7):
        p, v = s.state(t)
        w = 0.5
        assert np.allclose(p[0], [2 * np.cos(w * t), 2 * np.sin(w * t), 1.0], atol=1e-6)
        assert np.allclose(v[0], [-np.sin(w * t), np.cos(w * t), 0.0], atol=1e-6)
        a = np.array(obs[1]["start"]) - np.array(obs[1]["axis_position"])
        nrm = np.array(obs[1]["axis_ori"]) / 3.0
        assert np.dot(a, nrm) == 0.0  # start point in the plane through the axis point: tangent speed = speed
        q = p[1].astype(np.float64) - np.array(obs[1]["axis_position"])
        assert abs(np.linalg.norm(q) - np.linalg.norm(a)) < 1e-6 and abs(np.dot(q, nrm)) < 1e-6
        assert abs(np.linalg.norm(v[1]) - 0.7) < 1e-6 and abs(np.dot(v[1], q)) < 1e-6 and abs(np.dot(v[1], nrm)) < 1e-6
    # finite differences of the position give the velocity
    p0, v0 = s.state(1.0)
    p1, _ = s.state(1.001)
    assert np.allclose((p1.astype(np.float64) - p0) / 1e-3, v0, atol=2e-3)
