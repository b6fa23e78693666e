"""-m gpu: BASELINE.json configs[4], the 2048-agent stress case (2-D, M = 10, SFC boxes, communication range 3, about
2 500 DLSC-GC rows per agent, i.e. ten times the kernel's register cache: the rows are scanned from HBM).  The rows are
generated on the device (row N1) and never visit the host, except for a sample of agents that the oracle re-solves."""
import time

import numpy as np
import pytest
import torch

from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_x2048_stress_batch(gpu_required):
    cfg = workloads.x2048_config()
    w = workloads.x2048(cfg)
    N = len(w["agents"])
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(N, w["nbr_offsets"], w["nbr_ids"], w["radius"], w["downwash"])
        plans, goals = torch.from_numpy(w["plans"]).cuda(), torch.from_numpy(w["goals"]).cuda()
        gen.generate(goals, plans=plans)
        ro = gen.row_offsets
        db = opt.make_device_batch(w["agents"], ro, np.zeros(0, abi.ROW_DTYPE), w["sfc"], 0, want_f64=True, want_iters=True)
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(db.c, k, gen.soa[k].data_ptr())
        ro32 = torch.from_numpy(ro.astype(np.int32)).cuda()
        db.c.row_offsets = ro32.data_ptr()
        db.c.max_rows_per_agent = int(np.diff(ro).max())
        db.launch()
        got = db.results()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            gen.generate(goals, plans=plans)
            db.launch()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        _, rows = gen.rows_host()
    finally:
        opt.close()
    ok = got["status"] == abi.OK
    print("x2048: rows/agent %.0f, solved %.3f, iterations mean %.2f max %d, generation + LP + QP %.2f ms per step (%.0f agent-QP/s)"
          % (np.diff(ro).mean(), ok.mean(), got["iters"][:, 0].mean(), got["iters"][:, 0].max(), ms, N / ms * 1e3))
    assert ok.mean() > 0.9
    # ---- oracle on a sample of agents (same rows, downloaded) ---------------------------------------------------
    sample = np.random.default_rng(1).choice(N, 24, replace=False)
    for a in sample:
        r = rows[ro[a]:ro[a + 1]]
        ref = O.plan_batch(cfg, w["agents"][a:a + 1], np.array([0, len(r)], np.int64), r, w["sfc"][a:a + 1], 0, True)
        assert ref["goal_status"][0] == got["goal_status"][a]
        assert ref["status"][0] == got["status"][a], (a, ref["status"][0], got["status"][a])
        if ref["status"][0] == abi.OK:
            assert np.abs(ref["x"][0] - got["x"][a]).max() < 1e-6
            assert np.abs(ref["ctrl"][0] - got["ctrl"][a]).max() <= 1e-6
    # ---- size-independent properties on all 2048 agents ---------------------------------------------------------
    dim, M, n = cfg.world_dimension, cfg.M, cfg.n
    x = got["x"][ok].reshape(-1, dim, M, n + 1)
    ag, sfc = w["agents"][ok], w["sfc"][ok]
    assert np.abs(x[:, :, 0, 0] - ag["position"][:, :dim]).max() < 1e-9
    assert np.abs(x[:, :, 1:, 0] - x[:, :, :-1, n]).max() < 1e-9  # C0
    assert np.abs(x[:, :, -1, n] - x[:, :, -1, n - 2]).max() < 1e-9  # terminal stop
    v = (x[:, :, :, 1:] - x[:, :, :, :-1]) * n / cfg.dt
    v[:, :, 0, :2] = 0
    assert np.abs(v).max() <= 1.0 + 1e-6
    # safe flight corridor: every control point of segment m inside box m (first three points of segment 0 are fixed)
    lo, hi = sfc[:, :, :dim].transpose(0, 2, 1)[..., None], sfc[:, :, 3:3 + dim].transpose(0, 2, 1)[..., None]
    inside = (x >= lo - 1e-6) & (x <= hi + 1e-6)
    inside[:, :, 0, :3] = True
    assert inside.all()
    # communication range (traj_optimizer.cpp:490-513): segment end points stay within range/2 - r of every earlier start
    lim = 0.5 * cfg.communication_range - 0.15
    for m in range(M):
        for mi in range(m + 1):
            assert np.abs(x[:, :, m, n] - x[:, :, mi, 0]).max() <= lim + 1e-6


def test_x2048_rank_plans_its_interior_agents_before_the_exchange(gpu_required):
    """The N > 1 step on the stress batch, one rank of four on one GPU: the rank owns a contiguous
    id range; `sharding.halo_agents` says which of its agents have a neighbour on another rank.  Interior agents are
    placed first in the launch order and planned from a plan table whose remote part has NOT arrived yet (poisoned here),
    the halo agents afterwards from the gathered table (second launch over the remaining launch slots).  The result equals,
    bit for bit, the plain step: full table, one launch."""
    from dlsc_gc_planner_amd import sharding
    cfg = workloads.x2048_config()
    w = workloads.x2048(cfg)
    N = len(w["agents"])
    world, rank = 4, 1
    lo, hi = sharding.shard_range(N, rank, world)
    off, ids = w["nbr_offsets"].astype(np.int64), w["nbr_ids"]
    loc_off, loc_ids = (off[lo:hi + 1] - off[lo]).astype(np.int32), ids[off[lo]:off[hi]]
    halo = sharding.halo_agents(loc_off, loc_ids, lo, hi)
    interior = [a for a in range(hi - lo) if a not in set(halo)]
    assert len(halo) > 50 and len(interior) > 50, (len(halo), len(interior))
    order = np.array(interior + halo, np.int32)
    opt = BatchTrajOptimizer(cfg)
    try:
        gen = opt.make_lsc_generator(N, loc_off, loc_ids, w["radius"], w["downwash"], local_ids=np.arange(lo, hi))
        plans, goals = torch.from_numpy(w["plans"]).cuda(), torch.from_numpy(w["goals"]).cuda()
        ro = gen.row_offsets
        db = opt.make_device_batch(w["agents"][lo:hi], ro, np.zeros(0, abi.ROW_DTYPE), w["sfc"][lo:hi], 0, want_iters=True)
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(db.c, k, gen.soa[k].data_ptr())
        ro32 = torch.from_numpy(ro.astype(np.int32)).cuda()
        db.c.row_offsets = ro32.data_ptr()
        db.c.max_rows_per_agent = int(np.diff(ro).max())
        # plain step
        gen.generate(goals, plans=plans)
        db.launch()
        ref = db.results()
        # interior first, from a table without the other ranks' plans; then the halo from the gathered table
        late_plans, late_goals = plans.clone(), goals.clone()
        remote = torch.ones(N, dtype=torch.bool, device="cuda")
        remote[lo:hi] = False
        late_plans[remote] = 1.0e6
        late_goals[remote] = -1.0e6
        db.flip_outputs()
        db.set_order(order)
        gen.generate(late_goals, plans=late_plans)
        db.launch(offset=0, count=len(interior))
        gen.generate(goals, plans=plans)  # "table()": the exchange has completed
        db.launch(offset=len(interior), count=len(halo))
        got = db.results()
    finally:
        opt.close()
    assert (ref["status"] == abi.OK).mean() > 0.9
    for k in ("status", "goal_status", "goal", "ctrl"):
        assert np.array_equal(ref[k], got[k]), k
    print("x2048 rank %d of %d: %d interior agents planned before the exchange, %d halo agents after it" % (rank, world, len(interior), len(halo)))


def test_x2048_with_the_generators_presolve(gpu_required):
    """dlscqp_lsc_gen.prune on the full stress batch: rows implied by an agent's corridor boxes are never written; the plan
    kernel scans what is left.  Verdicts identical, sub-goals bit-equal, solutions within 1e-7 of the run on all rows."""
    cfg = workloads.x2048_config()
    w = workloads.x2048(cfg)
    N = len(w["agents"])
    opt = BatchTrajOptimizer(cfg)
    try:
        plans, goals = torch.from_numpy(w["plans"]).cuda(), torch.from_numpy(w["goals"]).cuda()
        res, ms, kept = {}, {}, None
        for mode in ("all", "pruned"):
            gen = opt.make_lsc_generator(N, w["nbr_offsets"], w["nbr_ids"], w["radius"], w["downwash"])
            ro = gen.row_offsets
            db = opt.make_device_batch(w["agents"], ro, np.zeros(0, abi.ROW_DTYPE), w["sfc"], 0, want_f64=True, want_iters=True)
            for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
                setattr(db.c, k, gen.soa[k].data_ptr())
            ro32 = torch.from_numpy(ro.astype(np.int32)).cuda()
            db.c.row_offsets = ro32.data_ptr()
            db.c.max_rows_per_agent = int(np.diff(ro).max())
            if mode == "pruned":
                db.use_row_counts(gen.enable_pruning(db.t["sfc"].view(torch.float32).view(N, cfg.M, 6)))
            gen.generate(goals, plans=plans)
            db.launch()
            res[mode] = db.results()
            if mode == "pruned":
                kept = gen.row_counts.cpu().numpy()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gen.generate(goals, plans=plans)
                db.launch()
            torch.cuda.synchronize()
            ms[mode] = (time.perf_counter() - t0) / 20 * 1e3
    finally:
        opt.close()
    a, b = res["all"], res["pruned"]
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["goal_status"], b["goal_status"])
    assert np.array_equal(a["goal"], b["goal"])
    ok = a["status"] == abi.OK
    assert ok.mean() > 0.9 and np.abs(a["x"][ok] - b["x"][ok]).max() < 1e-7
    full = np.diff(ro)
    print("x2048 presolve: %.0f of %.0f rows per agent kept; generation + LP + QP %.3f ms with all rows, %.3f ms pruned" % (
        kept.mean(), full.mean(), ms["all"], ms["pruned"]))
    assert kept.sum() < 0.5 * full.sum()
