"""-m gpu, row N2: corridor boxes grown on the device (dlscqp_generate_sfc_device) against the oracle's restatement
in the reference's maze world, and fed to the QP kernel as its SFC input without leaving HBM."""
import json
import os

import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import oracle_lib as O
from test_sfc import _flights, mixed_classes_check

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_corridors_match_oracle_and_feed_the_qp(gpu_required):
    m = json.load(open(os.path.join(HERE, "golden", "world_maze1.json")))
    cfg = abi.make_config(dim=2, M=10, dt=0.2, world_use_octomap=True, communication_range=3.0,
                          world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    w = O.SfcWorld(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], margin=m["radius"],
                   grid_resolution=m["grid_resolution"], max_vel=m["max_vel"], dt=cfg.dt)
    start, frames = _flights(m)
    N = len(start)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_sfc_generator(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], np.full(N, m["radius"]),
                                     np.full(N, m["max_vel"]), m["grid_resolution"])
        gen.initialize(torch.from_numpy(start).cuda())
        ref = np.stack([w.initialize(p, cfg.M)[1] for p in start])
        assert np.array_equal(gen.sfcs.cpu().numpy(), ref) and (gen.codes.cpu().numpy() == 0).all()
        for traj, goals, wps in frames:
            gen.update(torch.from_numpy(traj).cuda(), torch.from_numpy(goals).cuda(), torch.from_numpy(wps).cuda())
            codes = gen.codes.cpu().numpy()
            for a in range(N):
                code, ref[a] = w.construct(traj[a], goals[a], wps[a], ref[a])
                assert code == codes[a]
            assert np.array_equal(gen.sfcs.cpu().numpy(), ref)
        # the corridor tensor is the QP's SFC input: fresh corridors around the start cells, every agent plans a short
        # move inside its corridor; compared with the oracle's QP on the oracle's boxes
        gen.initialize(torch.from_numpy(start).cuda())
        ref = np.stack([w.initialize(p, cfg.M)[1] for p in start])
        rng = np.random.default_rng(4)
        goals = start.copy()
        goals[:, :2] += rng.uniform(-0.1, 0.1, (N, 2)).astype(np.float32)
        agents = np.zeros(N, abi.AGENT_DTYPE)
        agents["position"] = start
        agents["current_goal_point"], agents["next_waypoint"] = goals, goals
        agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = 1.0, 2.0, m["radius"], 1.0
        none = np.zeros(N + 1, np.int64)
        db = opt.make_device_batch(agents, none, np.zeros(0, abi.ROW_DTYPE), ref, 0, want_f64=True)
        db.c.sfc_boxes = gen.sfcs.data_ptr()
        db.launch()
        got = db.results()
        gen.close()
    finally:
        opt.close()
        w.close()
    exp = O.plan_batch(cfg, agents, none, np.zeros(0, abi.ROW_DTYPE), ref, 0, True)
    assert np.array_equal(got["status"], exp["status"]) and np.array_equal(got["goal_status"], exp["goal_status"])
    ok = exp["status"] == abi.OK
    assert ok.sum() >= 5
    assert np.abs(got["x"][ok] - exp["x"][ok]).max() < 1e-7
    x = got["x"][ok].reshape(-1, 2, cfg.M, cfg.n + 1)
    lo, hi = ref[ok][:, :, :2].transpose(0, 2, 1)[..., None], ref[ok][:, :, 3:5].transpose(0, 2, 1)[..., None]
    inside = (x >= lo - 1e-6) & (x <= hi + 1e-6)
    inside[:, :, 0, :3] = True
    assert inside.all()


def _device_generate(opt, m, radius, vmax, sfcs=None, position=None, initial_traj=None, goals=None, waypoints=None):
    """emul_lib.generate_sfc's signature over dlscqp_generate_sfc_device: a fresh world and generator per call (as a caller
    that changes its agents would), boxes in and out through the host"""
    gen = opt.make_sfc_generator(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], radius, vmax, m["grid_resolution"])
    try:
        if position is not None:
            gen.initialize(torch.from_numpy(position).cuda())
        else:
            gen.sfcs.copy_(torch.from_numpy(sfcs))
            gen.update(torch.from_numpy(initial_traj).cuda(), torch.from_numpy(goals).cuda(), torch.from_numpy(waypoints).cuda())
        return gen.sfcs.cpu().numpy().copy(), gen.codes.cpu().numpy().copy()
    finally:
        gen.close()


@pytest.mark.parametrize("variant", ["turns", "tables_on_first_launch", "in_sequence", "records"])
def test_device_corridors_of_mixed_radius_and_speed_classes(gpu_required, variant, monkeypatch):
    """The variants of tests/test_sfc.py on the device (the world reads DLSCQP_SFC_SVT / DLSCQP_SFC_RECORDS when it is created).
    "turns": the per-radius-class tables built by dlscqp_world_prepare from the host's radii (SfcGenerator does that);
    "tables_on_first_launch": without that call, by the first launch from the radii it reads back."""
    if variant == "tables_on_first_launch":
        monkeypatch.setenv("DLSCQP_WORLD_PREPARE", "0")
    if variant == "in_sequence":
        monkeypatch.setenv("DLSCQP_SFC_SVT", "0")
    if variant == "records":
        monkeypatch.setenv("DLSCQP_SFC_RECORDS", "1")
    m = json.load(open(os.path.join(HERE, "golden", "world_maze1.json")))
    cfg = abi.make_config(dim=2, M=10, dt=0.2, world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    opt = BatchTrajOptimizer(cfg)
    try:
        mixed_classes_check(m, lambda radius, vmax, **kw: _device_generate(opt, m, radius, vmax, **kw))
    finally:
        opt.close()


def test_device_corridors_in_open_space(gpu_required):
    """tests/test_sfc.py: test_device_source_open_space on the device (rounds of more than 64 trials)."""
    m = json.load(open(os.path.join(HERE, "golden", "world_maze1.json")))
    m["boxes"] = [[-1.8, -0.1, 0.25, 0.2, 0.2, 0.5]]
    m["world"] = [-2.0, -0.3, 0.0, 6.0, 4.3, 6.0]
    cfg = abi.make_config(dim=3, M=5, dt=0.2, world_use_octomap=True, world_min=tuple(m["world"][:3]), world_max=tuple(m["world"][3:]))
    pos = np.float32([[2.0, 2.0, 3.0], [0.5, 1.0, 1.0], [5.5, 4.0, 5.5]])
    opt = BatchTrajOptimizer(cfg)
    try:
        for vmax in (1.0, 25.0):
            w = O.SfcWorld(m["boxes"], m["world_resolution"], m["world"][:3], m["world"][3:], margin=m["radius"],
                           grid_resolution=m["grid_resolution"], max_vel=vmax, dt=0.2)
            try:
                dev, codes = _device_generate(opt, m, np.full(3, m["radius"]), np.full(3, vmax), position=pos)
                ref = np.stack([w.initialize(p, cfg.M)[1] for p in pos])
                assert (codes == 0).all() and np.array_equal(dev, ref)
            finally:
                w.close()
    finally:
        opt.close()
