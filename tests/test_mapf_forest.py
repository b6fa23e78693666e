"""CPU tier: the host waypoint planner (row N4, csrc/grid_mapf.cpp) in a world with DYNAMIC obstacles -- warning nodes, edges
into them, obstacles of interest from collision alerts, escape goals -- returns, step by step, the waypoints recorded in
tests/golden/mapf_forest_steps.npz by the implementation that rebuilt map, adjacency and every distance table from scratch
each step (generator: tests/golden/make_mapf_forest.py).  The static-world behaviour is pinned to the reference's logged
run in tests/test_log_replay.py; this pins the incremental map and the lazy distance tables to the plain restatement."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_planner_with_dynamic_obstacles_returns_the_recorded_waypoints():
    import make_mapf_forest
    want = np.load(os.path.join(HERE, "golden", "mapf_forest_steps.npz"))
    got = make_mapf_forest.drive()
    assert len(np.unique(want["waypoint"].reshape(len(want["waypoint"]), -1), axis=0)) >= 60  # the crowd moves: new waypoints all along
    assert (want["rc"] == 0).all()
    for k in ("rc", "group", "desired", "waypoint"):
        assert np.array_equal(want[k], got[k]), k
