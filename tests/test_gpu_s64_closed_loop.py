"""-m gpu: BASELINE.json configs[1] flown with the WHOLE replanning step on the device -- neighbour lists (the workload's rule: the
five nearest agents of the swarm, dlscqp_neighbours_knn_device), DLSC-GC half-spaces with the presolve and the tail rule (4 x 27 + 12
rows per agent at most), goal LP + QP in the kernel's own launch order, state update -- what bench.py's `s64_closed_loop` times at
512 swarms.  Checked here on a few swarms: the device's lists against the workload generator's rule on the step's own positions, the
rows' bookkeeping, and on a late step the plans of a sample of agents against the CPU oracle on that step's own inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.harness import ClosedLoop
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,group,k", [(64, 64, 5), (200, 64, 5), (7, 3, 5), (130, 130, 8)])
def test_the_k_nearest_of_the_block(gpu_required, n, group, k):
    opt = BatchTrajOptimizer(workloads.s64_config())
    try:
        rng = np.random.default_rng(n)
        pos = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
        pos[1] = pos[0]  # two agents on one spot: distance 0, ties by id
        if n > 20:
            pos[11] = pos[10] + np.float32(0.5) * (pos[12] - pos[10])  # equidistant candidates
        pitch = k
        ids = torch.full((n, pitch), -1, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
        p = torch.from_numpy(pos).cuda()
        a = abi.NeighbourKnn()
        a.n_total, a.group, a.k, a.pitch = n, group, k, pitch
        a.position, a.position_stride = p.data_ptr(), 3
        a.out_nbr_ids, a.out_nbr_counts = ids.data_ptr(), cnt.data_ptr()
        assert opt.lib.dlscqp_neighbours_knn_device(opt.handle, C.byref(a), None) == abi.SUCCESS
        torch.cuda.synchronize()
        got, gc = ids.cpu().numpy(), cnt.cpu().numpy()
        for s in range(n):
            g0 = (s // group) * group
            members = np.array([q for q in range(g0, min(g0 + group, n)) if q != s], dtype=np.int64)
            d = ((pos[members].astype(np.float64) - pos[s].astype(np.float64)) ** 2).sum(1)
            order = members[np.lexsort((members, d))][:k]
            assert gc[s] == len(order) and np.array_equal(got[s, :gc[s]], order), (s, got[s], order)
    finally:
        opt.close()


def test_s64_flies_with_every_stage_on_the_device(gpu_required):
    cfg = workloads.s64_config()
    n_swarms = 6
    N = n_swarms * 64
    start, goal = workloads.swarm_missions(cfg, n_swarms=n_swarms, swarm_size=64, seed=2000)
    # the same swarms as the host generator's (bench.py's batches)
    wl = workloads.generate(cfg, lambda *a: None, n_swarms=n_swarms, swarm_size=64, seed=2000, steps=0)
    assert np.array_equal(wl["agents"]["position"], start)
    sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=5.0, neighbours="knn",
                     knn=(64, 5), tail=(4, 2), prune=True, order_hint=True)
    try:
        sim.keep_inputs = True
        its, fails = [], 0
        for step in range(14):
            pos_before = sim.pos.cpu().numpy().copy()
            sim.step()
            res = sim.batch.results()
            its.append(res["iters"][:, 0].mean())
            fails += int((res["status"] != abi.OK).sum())
            if step in (0, 5, 13):
                counts, ids = sim.nbr.lists_host()
                want = workloads._neighbours(pos_before, 5, 64)
                # (the generator sorts by the norm, the device by the squared distance: the same order up to exact ties)
                assert (counts == 5).all() and np.array_equal(np.sort(ids, 1), np.sort(want, 1)), step
                rc = sim.row_counts.cpu().numpy()
                assert rc.max() <= 120 and rc.min() >= 0
        recs, rows_off, rows, _, _ = sim.step_inputs()
        # rows of an agent: full neighbours first (every segment), the fifth with its last two segments only
        per_agent = np.diff(rows_off)
        assert per_agent.max() <= 120
        sample = np.random.default_rng(3).choice(N, 32, replace=False)
        worst, n_ok = 0.0, 0
        for a in sample:
            r = rows[rows_off[a]:rows_off[a + 1]]
            ref = O.plan_batch(cfg, recs[a:a + 1], np.array([0, len(r)], np.int64), r, None, 0, True)
            assert ref["goal_status"][0] == res["goal_status"][a]
            assert ref["status"][0] == res["status"][a], (a, ref["status"][0], res["status"][a])
            n_ok += int(ref["status"][0] == abi.OK)
            if ref["status"][0] == abi.OK:
                worst = max(worst, float(np.abs(ref["ctrl"][0].astype(np.float64) - res["ctrl"][a]).max()))
        assert worst <= 1e-5 and n_ok >= 24
        d0 = np.linalg.norm(start - goal, axis=1).mean()
        d1 = np.linalg.norm(sim.pos.cpu().numpy() - goal, axis=1).mean()
        print("S64 on the device, %d agents, 14 steps: mean iterations per step %.2f, rows kept per agent %.1f of 120, fail-safes %d of %d, "
              "mean distance to goal %.2f -> %.2f m, oracle sample worst |ctrl - oracle| %.2g" % (N, np.mean(its), per_agent.mean(), fails, 14 * N, d0, d1, worst))
        assert np.mean(its[3:]) > 0.3 and d1 < d0 - 1.0 and fails <= 0.1 * 14 * N
    finally:
        sim.close()


def test_two_agents_per_wavefront_write_the_same_rows(gpu_required, monkeypatch):
    """dlscqp_lsc_pair_kernel (short lists: two agents share a wavefront) against one wavefront per agent: the same rows at the
    same places, the same counts, the same flight -- with an odd number of agents (the last wavefront has one)."""
    cfg = workloads.s64_config()
    start, goal = workloads.swarm_missions(cfg, n_swarms=3, swarm_size=64, seed=2001)
    start, goal = start[:-1], goal[:-1]  # 191 agents
    flights = []
    for pairs in ("0", "1"):
        monkeypatch.setenv("DLSCQP_LSC_PAIRS", pairs)
        sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=5.0, neighbours="knn",
                         knn=(64, 5), tail=(4, 2), prune=True)
        try:
            sim.keep_inputs = True
            rec = []
            for step in range(8):
                sim.step()
                _, rows_off, rows, _, _ = sim.step_inputs()
                rec.append((rows_off.copy(), rows.copy(), sim.plans.cpu().numpy().copy()))
            flights.append(rec)
        finally:
            sim.close()
    for k, (a, b) in enumerate(zip(*flights)):
        assert np.array_equal(a[0], b[0]), k
        assert a[1].tobytes() == b[1].tobytes(), k
        assert np.array_equal(a[2], b[2]), k
    assert np.diff(flights[0][-1][0]).mean() > 20  # rows were written
