"""-m gpu: the communication-range neighbour lists on the device (dlscqp_neighbours_device: broadcastMsgs' filter,
multi_sync_simulator.cpp:482-503) and BASELINE.json configs[4] flown as a MOVING crowd with them: 2 048 agents in one
2-D arena, every step neighbour search -> corridor boxes -> half-spaces (presolve) -> goal LP + QP -> state update, all in
HBM."""
import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.harness import ClosedLoop
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import oracle_lib as O

pytestmark = pytest.mark.gpu


def host_lists(pos, rng_, local=None):
    """The reference's filter in numpy: LInfinityDistance (util.hpp:131-140: float32 differences, compared in double)."""
    pos = np.asarray(pos, np.float32)
    ids = np.arange(len(pos)) if local is None else np.asarray(local)
    out = []
    for a in ids:
        d = np.abs(pos[a][None, :] - pos).max(axis=1).astype(np.float64)
        keep = ~(d > rng_) if rng_ > 0 else np.ones(len(pos), bool)
        keep[a] = False
        out.append(np.nonzero(keep)[0])
    return out


@pytest.mark.parametrize("way", ["cells", "direct"])
@pytest.mark.parametrize("dim,rng_", [(3, 3.0), (2, 3.0), (3, -1.0), (3, 0.7)])
def test_device_lists_equal_the_reference_filter(gpu_required, dim, rng_, way, monkeypatch):
    """way: through the cells of one communication range (four launches: more than 256 agents), or every agent a candidate of every
    other (one launch: the reference's missions) -- here both on the same 700 agents."""
    monkeypatch.setenv("DLSCQP_NBR_DIRECT_MAX", "100000" if way == "direct" else "0")
    N, pitch = 700, 256
    cfg = abi.make_config(dim=dim, M=5, communication_range=rng_, world_min=(-8.0, -8.0, 0.0), world_max=(8.0, 8.0, 2.5))
    r = np.random.default_rng(3)
    pos = r.uniform(-9.0, 9.0, (N, 3)).astype(np.float32)  # some agents outside the world box: cells are clamped
    pos[:, 2] = r.uniform(0.0, 2.5, N) if dim == 3 else 1.0
    pos[:40] = pos[40:80] + np.float32(rng_ if rng_ > 0 else 1.0) * np.array([1, 0, 0], np.float32)  # pairs exactly at the range
    local = np.sort(r.choice(N, 300, replace=False)).astype(np.int32)
    opt = BatchTrajOptimizer(cfg)
    try:
        for loc, p in ((None, pitch if rng_ > 0 else N - 1), (local, pitch if rng_ > 0 else N - 1), (None, 20)):
            n_local = N if loc is None else len(loc)
            gen = opt.make_lsc_generator(N, np.arange(n_local + 1) * p, np.zeros(n_local * p, np.int32), np.full(N, 0.15), np.full(N, 2.0),
                                         local_ids=loc)
            ns = opt.make_neighbour_search(gen, p)
            # positions as a strided view, like the agent records the closed loop keeps them in
            rec = torch.zeros((N, 26), dtype=torch.float32, device="cuda")
            rec[:, 0:3] = torch.from_numpy(pos).cuda()
            ns.search(rec[:, 0:3])
            counts, ids = ns.lists_host()
            want = host_lists(pos, rng_, loc)
            over = int(ns.overflow.cpu()[0])
            cut = 0
            for k in range(n_local):
                w = want[k]
                if len(w) > p:
                    cut += 1
                    w = w[:p]  # a list that does not fit is cut to its smallest ids
                assert counts[k] == len(w), (k, counts[k], len(w))
                assert np.array_equal(ids[k, :counts[k]], w), k
            assert (over != 0) == (cut > 0)
            assert np.array_equal(ns.row_counts.cpu().numpy(), counts * (cfg.M * (cfg.n + 1) - cfg.phi))
            if p == 20:
                assert cut > 0 or rng_ == 0.7
    finally:
        opt.close()


def test_search_after_the_grid_changed(gpu_required, monkeypatch):
    """the search has no memset of its own -- its last kernel leaves the cell counts at zero for the next search.  That
    holds for the SAME grid only: dlscqp_update_param with another communication range lays the scratch out differently (what were cell
    starts are counts now).  Ranges 0.7 -> 3.0 -> 0.7 -> 1.3 on one context: every search's lists equal the reference filter."""
    monkeypatch.setenv("DLSCQP_NBR_DIRECT_MAX", "0")
    N, p = 700, 256
    r = np.random.default_rng(11)
    pos = r.uniform(-9.0, 9.0, (N, 3)).astype(np.float32)
    pos[:, 2] = r.uniform(0.0, 2.5, N)
    mk = lambda rng_: abi.make_config(dim=3, M=5, communication_range=rng_, world_min=(-8.0, -8.0, 0.0), world_max=(8.0, 8.0, 2.5))
    opt = BatchTrajOptimizer(mk(0.7))
    try:
        gen = opt.make_lsc_generator(N, np.arange(N + 1) * p, np.zeros(N * p, np.int32), np.full(N, 0.15), np.full(N, 2.0))
        ns = opt.make_neighbour_search(gen, p)
        dpos = torch.from_numpy(pos).cuda()
        for rng_ in (0.7, 3.0, 0.7, 1.3, 3.0):
            opt.update_param(mk(rng_))
            for _ in range(2):  # (the second search of a grid runs on the counts the first one left)
                ns.search(dpos)
                counts, ids = ns.lists_host()
                want = host_lists(pos, rng_)
                for k in range(N):
                    w = want[k][:p]
                    assert counts[k] == len(w), (rng_, k, counts[k], len(w))
                    assert np.array_equal(ids[k, :counts[k]], w), (rng_, k)
    finally:
        opt.close()


def test_x2048_flies_as_a_moving_crowd(gpu_required):
    """2 048 agents cross one arena for 60 steps; at several steps (also late ones, when the crowd is interleaved) the
    device's neighbour lists equal the reference filter on that step's positions, and on the last checked step a sample of
    agents is re-planned by the CPU oracle from the step's own inputs (rows, corridor boxes and agent records read back
    from HBM)."""
    cfg = workloads.x2048_config()
    N, pitch = 2048, 256  # the crowd gets denser towards the centre of the arena
    start, goal = workloads.swapping_pairs(cfg, N)
    sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5,
                     neighbours="device", nbr_pitch=pitch, corridor_margin=0.5, prune=True, order_hint=True)
    per = cfg.M * (cfg.n + 1) - cfg.phi
    try:
        sim.keep_inputs = True
        its, fails, seps = [], 0, []
        checked_lists = 0
        for k in range(60):
            check = k in (0, 1, 10, 30, 45, 59)
            if check:
                pos_before = sim.pos.cpu().numpy().copy()
            sim.step()
            res = sim.batch.results()
            its.append(res["iters"][:, 0].mean())
            fails += int((res["status"] != abi.OK).sum())
            if check:
                counts, ids = sim.nbr.lists_host()
                assert int(sim.nbr.overflow.cpu()[0]) == 0, (k, counts.max())
                want = host_lists(pos_before, cfg.communication_range)
                for a in range(N):
                    assert counts[a] == len(want[a]) and np.array_equal(ids[a, :counts[a]], want[a]), (k, a)
                checked_lists += 1
            seps.append(sim.min_separation())
        # ---- the last step's plans against the oracle, on that step's own inputs
        recs, rows_off, rows, boxes, _ = sim.step_inputs()
        rc = np.diff(rows_off)
        sample = np.random.default_rng(2).choice(N, 24, replace=False)
        worst, n_ok = 0.0, 0
        for a in sample:
            r = rows[rows_off[a]:rows_off[a + 1]]
            ref = O.plan_batch(cfg, recs[a:a + 1], np.array([0, len(r)], np.int64), r, boxes[a:a + 1], 0, True)
            assert ref["goal_status"][0] == res["goal_status"][a]
            assert ref["status"][0] == res["status"][a], (a, ref["status"][0], res["status"][a])
            n_ok += int(ref["status"][0] == abi.OK)
            if ref["status"][0] == abi.OK:
                worst = max(worst, float(np.abs(ref["ctrl"][0].astype(np.float64) - res["ctrl"][a]).max()))
        assert worst <= 1e-5 and n_ok >= 20
        d0 = np.linalg.norm(start[:, :2] - goal[:, :2], axis=1).mean()
        d1 = np.linalg.norm(sim.pos.cpu().numpy()[:, :2] - goal[:, :2], axis=1).mean()
        print("moving crowd, %d agents, 60 steps: mean iterations per step %.2f (max %.2f), rows kept per agent %.0f of %.0f, "
              "fail-safes %d of %d, min separation ratio %.4f, mean distance to goal %.2f -> %.2f m, lists checked at %d steps, "
              "oracle sample worst |ctrl - oracle| %.2g" % (N, np.mean(its), np.max(its), rc.mean(), (sim.nbr.counts.cpu().numpy() * per).mean(),
                                                            fails, 60 * N, min(seps), d0, d1, checked_lists, worst))
        assert np.mean(its[20:]) > 0.3  # a crowd that moves keeps the solver busy
        assert min(seps) >= 1.0 - 1e-5  # DLSC: never below the collision distance (float32 positions)
        assert d1 < d0 - 0.5  # counter-flowing pairs in a dense crowd: slow, but they get there
        assert fails <= 0.02 * 60 * N
    finally:
        sim.close()


def test_boxes_of_an_open_world_come_out_of_the_state_update(gpu_required, monkeypatch):
    """dlscqp_advance.out_boxes: the next step's per-segment boxes written by the state-update kernel are, bit for bit, what the
    harness computed with five tensor operations at the start of that step -- and so are the flights."""
    cfg = workloads.x2048_config()
    N = 128
    start, goal = workloads.swapping_pairs(cfg, N)
    flights = []
    for mode in ("0", "1"):
        monkeypatch.setenv("DLSCQP_BOX_IN_ADVANCE", mode)
        sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5,
                         neighbours="device", nbr_pitch=64, corridor_margin=0.5, prune=True)
        try:
            assert sim.box_in_advance == (mode == "1")
            rec = []
            for k in range(12):
                sim.step()
                # (the boxes the step's launch READ: with the kernel's boxes box_t already holds the next step's)
                rec.append((sim.plans.cpu().numpy().copy(), sim.batch.results()["status"].copy()))
            boxes_next = sim.box_t.cpu().numpy().copy()
            if mode == "0":  # what the tensor operations would hand the next step
                import torch
                lo = torch.maximum(sim.init_buf.amin(dim=2) - sim.box_margin, sim.box_lo)
                hi = torch.minimum(sim.init_buf.amax(dim=2) + sim.box_margin, sim.box_hi)
                boxes_next = torch.cat([lo, hi], dim=2).cpu().numpy()
            flights.append((rec, boxes_next))
        finally:
            sim.close()
    for (p0, s0), (p1, s1) in zip(flights[0][0], flights[1][0]):
        assert np.array_equal(s0, s1) and np.array_equal(p0, p1)
    assert np.array_equal(flights[0][1], flights[1][1])
