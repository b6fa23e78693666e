"""Row N3: the reference's own input files (a mission JSON and a world CSV, committed as data under tests/golden/data) read at
run time by dlsc_gc_planner_amd.mission, against the numbers the fixtures hold; under -m gpu the mission is flown from the
files alone and ends like the reference's logged run."""
import json
import os

import numpy as np
import pytest

from dlsc_gc_planner_amd import mission

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")


def test_mission_and_world_readers_reproduce_the_fixture_numbers():
    fx = json.load(open(os.path.join(HERE, "golden", "world_maze1.json")))
    m = mission.read_mission(os.path.join(DATA, "maze10_1.json"), world_dimension=2, world_z_2d=1.0)
    assert np.array_equal(m["start"], np.float32(fx["start"])) and np.array_equal(m["goal"], np.float32(fx["goal"]))
    assert list(m["world_min"]) + list(m["world_max"]) == [np.float32(v) for v in fx["world"]]
    assert (m["radius"] == fx["radius"]).all() and (m["max_vel"] == 1.0).all() and (m["max_acc"] == 2.0).all() and (m["downwash"] == 2.0).all()
    boxes = mission.read_world_csv(os.path.join(DATA, "maze1.csv"))
    assert boxes.shape == (50, 6) and np.allclose(boxes, np.array(fx["boxes"]), atol=1e-7)
    f = mission.read_mission(os.path.join(DATA, "forest_spin4_100_1.json"))
    assert len(f["obstacles"]) == 4 and all(o["type"] == "spin" and o["downwash"] >= 1 for o in f["obstacles"])
    assert f["start"].shape == (10, 3) and f["start"][:, 2].std() >= 0  # 3-D: z kept
    cfg, prm = mission.make_config(m, "simulation")
    assert cfg.M == 10 and cfg.world_dimension == 2 and cfg.world_use_octomap == 1 and cfg.communication_range == 3.0
    cfg3, prm3 = mission.make_config(f, "testall_DLSCGC_3D")
    assert cfg3.M == 10 and cfg3.world_dimension == 3 and cfg3.slack_collision_weight == 100.0 and prm3["goal_threshold"] == 0.6


@pytest.mark.gpu
def test_flying_the_reference_files_ends_like_the_reference_run(gpu_required, tmp_path):
    """maze10_1.json + maze1.csv with launch/simulation.launch's parameters: total_flight_time 34 s and safety_ratio_agent
    1.00058 in the reference's log/summary_DLSCGC_10agents.csv."""
    out = tmp_path / "result.csv"
    s = mission.fly(os.path.join(DATA, "maze10_1.json"), os.path.join(DATA, "maze1.csv"), "simulation", str(out))
    assert s["total_flight_time"] is not None and abs(s["total_flight_time"] - 34.0) < 1e-9
    assert abs(s["safety_ratio_agent"] - 1.00058) < 2e-4 and s["qp_failsafes"] == 0
    assert abs(s["total_flight_distance"] - 134.096) < 0.7  # summed over 0.2 s chords here, over 0.1 s samples in the reference
    lines = open(out).read().strip().split("\n")
    assert len(lines) == 1 + 2 * 170
    f = mission.fly(os.path.join(DATA, "forest_spin4_100_1.json"), os.path.join(DATA, "forest1.csv"), "testall_DLSCGC_3D")
    assert f["total_flight_time"] is not None and f["total_flight_time"] < 20.0 and f["safety_ratio_obs"] >= 1.0 and f["safety_ratio_agent"] >= 1.0
