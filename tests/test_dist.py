"""CPU tier: the N>1 path (contiguous agent shards + ONE all-gather) with world_size 2 over gloo.
The per-shard solve is done by the lane-serial test build of the kernel source; the collective and
the shard arithmetic are the product's (dlsc_gc_planner_amd/sharding.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dlsc_gc_planner_amd import sharding


def _worker(rank, world, port, n_agents, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import emul_lib as E
    import fixtures
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, b, _ = fixtures.load("s64")
    lo, hi = sharding.shard_range(n_agents, rank, world)
    ro = b["row_offsets"]
    got = E.plan_batch(cfg, b["agents"][lo:hi], ro[lo:hi + 1] - ro[lo], b["rows"][ro[lo]:ro[hi]], None, 0, True)
    local = torch.from_numpy(got["ctrl"])
    table = sharding.gather_plans(local, n_agents, dist, rank, world)
    q.put((rank, table.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_agents):
    import emul_lib as E
    import fixtures
    cfg, b, _ = fixtures.load("s64")
    ro = b["row_offsets"]
    full = E.plan_batch(cfg, b["agents"][:n_agents], ro[:n_agents + 1], b["rows"][:ro[n_agents]], None, 0, True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + n_agents
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_agents, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert res[r].shape == full["ctrl"].shape
        assert np.array_equal(res[r], full["ctrl"])


def test_two_ranks_even_shards():
    _run(64)


def test_two_ranks_uneven_shards():
    _run(33)


def test_shard_ranges_cover_all_agents():
    for n in (0, 1, 7, 64, 2048, 2049):
        for w in (1, 2, 3, 8):
            r = [sharding.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_permute_batch_moves_agents_with_their_rows():
    import fixtures
    from dlsc_gc_planner_amd import workloads
    cfg, b, _ = fixtures.load("s64")
    ro = b["row_offsets"]
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(b["agents"]))
    pa, pro, prows = workloads.permute_batch(b["agents"], ro, b["rows"], perm)
    assert pro[0] == 0 and pro[-1] == ro[-1] and pro.dtype == ro.dtype
    for s, a in enumerate(perm):
        assert pa[s].tobytes() == b["agents"][a].tobytes()
        assert prows[pro[s]:pro[s + 1]].tobytes() == b["rows"][ro[a]:ro[a + 1]].tobytes()


def _exchange_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, tail = 5, (2, 3)
    x = sharding.PlanExchange(dist, world, n_local, (6, 3, 1), torch.device("cpu"))
    seen = []
    local = torch.zeros((n_local,) + tail)
    goal = torch.zeros((n_local, 3))
    status = torch.zeros(n_local, dtype=torch.int32)
    for k in range(5):
        # "step k": the rank's own output; overwritten in place right after the post, like the kernel's output buffers
        local[:] = 100.0 * rank + k + torch.arange(n_local, dtype=torch.float32)[:, None, None]
        goal[:] = -local[:, 0, :]
        status[:] = -7 * rank - k
        x.post(local, goal, status)
        local.fill_(-1.0)  # step k+1 writes the output buffers while exchange k is in flight
        goal.fill_(-1.0)
        status.fill_(99)
        if k % 2 == 1 or k == 4:
            ctrl, g, st = x.unpack(x.table(), (torch.float32, torch.float32, torch.int32))
            assert torch.equal(g, -ctrl[:, :3])
            assert st.flatten().tolist() == [-7 * r - k for r in range(world) for _ in range(n_local)]
            seen.append((k, ctrl.reshape((-1,) + tail).clone().numpy()))
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_plan_exchange_over_gloo():
    """PlanExchange: exchange k is posted asynchronously and snapshots the output of step k; table() returns the gathered
    table of the most recent post on every rank, also when the output buffer has been overwritten meanwhile."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert [k for k, _ in res[r]] == [1, 3, 4]
        for k, tab in res[r]:
            exp = np.concatenate([np.broadcast_to((100.0 * g + k + np.arange(5, dtype=np.float32))[:, None, None], (5, 2, 3)) for g in range(2)])
            assert np.array_equal(tab, exp), (r, k)


def test_halo_agents_are_the_ones_with_remote_neighbours():
    # two swarms of three agents, everyone sees everyone in its swarm; a seventh agent sees agent 0 and agent 5
    nbr = {0: [1, 2], 1: [0, 2], 2: [0, 1], 3: [4, 5], 4: [3, 5], 5: [3, 4], 6: [0, 5]}
    for lo, hi, exp in ((0, 3, []), (3, 6, []), (0, 4, [3]), (3, 7, [3]), (0, 7, [])):
        ids, off = [], [0]
        for a in range(lo, hi):
            ids += nbr[a]
            off.append(len(ids))
        assert sharding.halo_agents(off, ids, lo, hi) == exp, (lo, hi)


# ---- BASELINE.json configs[2]: a mission flown in closed loop with its agents sharded over two ranks -------------------
def _closed_loop(cfg, m, steps, rank=0, world=1, pg=None):
    """ClosedLoop's data flow (dlsc_gc_planner_amd/harness.py) on the CPU for the agents [lo, hi) of a mission: half-spaces
    from the table of everybody's previous plans and sub-goals (lane-serial build of the device generator, local_ids),
    goal LP + QP (lane-serial build of the kernel), fail-safe / sub-goal / state := plan(dt), then the exchange
    (sharding.PlanExchange: control points, sub-goal, status of the local agents -> table on every rank).  Every agent
    is every other agent's neighbour in these missions, so all agents are halo agents: the next step waits for the table."""
    import emul_lib as E
    from dlsc_gc_planner_amd import abi
    start, goal = np.array(m["start"], np.float32), np.array(m["goal"], np.float32)
    N, M, n = len(start), cfg.M, cfg.n
    lo, hi = sharding.shard_range(N, rank, world)
    nl = hi - lo
    ids = np.arange(N)
    nbr_ids = np.concatenate([np.delete(ids, a) for a in range(lo, hi)]).astype(np.int32)
    nbr_off = (np.arange(nl + 1) * (N - 1)).astype(np.int32)
    radius, downwash = np.full(N, m["radius"]), np.full(N, m["downwash"])
    # tables every rank holds (all agents) and the local state
    plans_t, goals_t = np.zeros((N, M, n + 1, 3), np.float32), start.copy()
    pos, vel, acc = start[lo:hi].copy(), np.zeros((nl, 3), np.float32), np.zeros((nl, 3), np.float32)
    agents = np.zeros(nl, abi.AGENT_DTYPE)
    agents["id"], agents["n_obs"] = ids[lo:hi], N - 1
    agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = m["max_vel"], m["max_acc"], m["radius"], m["nominal_velocity"]
    xchg = None if pg is None else sharding.PlanExchange(pg, world, nl, (M * (n + 1) * 3, 3, 1), torch.device("cpu"))
    s1, s2 = np.float32(n / cfg.dt), np.float32((n - 1) / cfg.dt)
    for k in range(steps):
        d = goal[lo:hi] - pos
        dist_ = np.sqrt((d * d).sum(1, dtype=np.float32).astype(np.float64))
        wp = np.where((dist_ <= 0.5)[:, None], goal[lo:hi], pos + d / np.maximum(dist_, 1e-9).astype(np.float32)[:, None] * np.float32(0.5))
        if k == 0:
            ro, rows = E.generate_lsc(cfg, None, goals_t, radius, downwash, nbr_off, nbr_ids, local_ids=ids[lo:hi],
                                      position=start, velocity=np.zeros_like(start))
            init = np.repeat(np.repeat(pos[:, None, None, :], M, 1), n + 1, 2)
        else:
            ro, rows = E.generate_lsc(cfg, plans_t, goals_t, radius, downwash, nbr_off, nbr_ids, local_ids=ids[lo:hi])
            p = plans_t[lo:hi]
            init = np.concatenate([p[:, 1:], np.repeat(p[:, -1:, -1:], n + 1, axis=2)], axis=1)
        agents["position"], agents["velocity"], agents["acceleration"] = pos, vel, acc
        agents["current_goal_point"], agents["next_waypoint"] = goals_t[lo:hi], wp
        got = E.plan_batch(cfg, agents, ro, rows, None, 0, True)
        ok = got["status"] == abi.OK
        new = np.where(ok[:, None, None, None], got["ctrl"], init).astype(np.float32)
        cg = np.where((got["goal_status"] != abi.INFEASIBLE)[:, None], got["goal"], goals_t[lo:hi]).astype(np.float32)
        c0, c1, c2 = new[:, 1, 0], new[:, 1, 1], new[:, 1, 2]
        v0, v1 = (c1 - c0) * s1, (c2 - c1) * s1
        pos, vel, acc = c0.copy(), v0, (v1 - v0) * s2
        if xchg is None:
            plans_t, goals_t = new, cg
        else:
            xchg.post(torch.from_numpy(new), torch.from_numpy(cg), torch.from_numpy(got["status"].astype(np.int32)))
            ctrl_all, goal_all, st_all = xchg.unpack(xchg.table(), (torch.float32, torch.float32, torch.int32))
            plans_t = ctrl_all.reshape(N, M, n + 1, 3).numpy().copy()
            goals_t = goal_all.numpy().copy()
            assert np.array_equal(st_all.flatten().numpy()[lo:hi], got["status"])
    return plans_t, goals_t


def _mission_cfg(name):
    import json
    from dlsc_gc_planner_amd import abi
    here = os.path.dirname(os.path.abspath(__file__))
    m = json.load(open(os.path.join(here, "golden", "missions_empty.json")))[name]
    w = m["world"]
    cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                          world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]))
    return cfg, m


def _mission_worker(rank, world, port, name, steps, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, m = _mission_cfg(name)
    plans, goals = _closed_loop(cfg, m, steps, rank, world, dist)
    q.put((rank, plans, goals))
    dist.barrier()
    dist.destroy_process_group()


def test_mission_sharded_over_two_ranks_equals_one_process():
    """empty60 (BASELINE.json configs[2]: 60 agents, 1 593 half-spaces each) flown in closed loop for three replanning
    steps with agents 0-29 on rank 0 and 30-59 on rank 1 and one plan exchange per step: the table of plans and sub-goals
    on either rank equals the single-process run bit for bit."""
    name, steps = "empty60_1", 3
    cfg, m = _mission_cfg(name)
    ref_plans, ref_goals = _closed_loop(cfg, m, steps)
    assert np.abs(ref_plans[:, -1, -1] - np.array(m["start"], np.float32)).max() > 0.2  # the agents are on their way
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_mission_worker, args=(r, 2, port, name, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (a, b) for r, a, b in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(res[r][0], ref_plans) and np.array_equal(res[r][1], ref_goals), r


# ---- BASELINE.json configs[4] (stress batch): the interior / halo step over two gloo ranks -------------------------------
def _stress_inputs(n_agents=192):
    from dlsc_gc_planner_amd import workloads
    cfg = workloads.x2048_config()
    return cfg, workloads.x2048(cfg, n_agents=n_agents, seed=5001)


def _stress_plan(cfg, w, ids, plans, goals, agents):
    """Half-spaces (device source, host build) + goal LP + QP (kernel source, host build) of the agents `ids` from the table
    (plans, goals) of everybody's previous plans and sub-goals."""
    import emul_lib as E
    ids = np.asarray(ids, np.int64)
    off = w["nbr_offsets"].astype(np.int64)
    cnt = off[ids + 1] - off[ids]
    lo = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    nb = np.concatenate([w["nbr_ids"][off[a]:off[a + 1]] for a in ids]) if len(ids) else np.zeros(0, np.int32)
    ro, rows = E.generate_lsc(cfg, plans, goals, w["radius"], w["downwash"], lo, nb, local_ids=ids.astype(np.int32))
    return E.plan_batch(cfg, agents[ids], ro, rows, w["sfc"][ids], 0, True)


def _stress_advance(cfg, prev_plans, agents, res, ids):
    """What dlscqp_advance_device does between two steps, for the agents `ids`: fail-safe, sub-goal, state := plan(dt)."""
    ok = res["status"] == 0
    shifted = np.concatenate([prev_plans[ids][:, 1:], np.repeat(prev_plans[ids][:, -1:, -1:], cfg.n + 1, axis=2)], axis=1)
    plans = np.where(ok[:, None, None, None], res["ctrl"], shifted).astype(np.float32)
    ag = agents[ids].copy()
    gok = res["goal_status"] != 1
    ag["current_goal_point"] = np.where(gok[:, None], res["goal"], ag["current_goal_point"])
    s1, s2 = np.float32(cfg.n / cfg.dt), np.float32((cfg.n - 1) / cfg.dt)
    c0, c1, c2 = plans[:, 1, 0], plans[:, 1, 1], plans[:, 1, 2]
    v0, v1 = (c1 - c0) * s1, (c2 - c1) * s1
    ag["position"], ag["velocity"], ag["acceleration"] = c0, v0, (v1 - v0) * s2
    return plans, ag


def _stress_worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, w = _stress_inputs()
    N = len(w["agents"])
    lo, hi = sharding.shard_range(N, rank, world)
    own = np.arange(lo, hi)
    off = w["nbr_offsets"].astype(np.int64)
    halo = sharding.halo_agents(off[lo:hi + 1] - off[lo], w["nbr_ids"][off[lo]:off[hi]], lo, hi)
    interior = [a for a in range(hi - lo) if a not in set(halo)]
    E3 = cfg.M * (cfg.n + 1) * 3
    xchg = sharding.PlanExchange(dist, world, hi - lo, (E3, 3, 1), torch.device("cpu"))
    # step 0: everybody holds the initial table; plan the own shard, post the plans
    r0 = _stress_plan(cfg, w, own, w["plans"], w["goals"], w["agents"])
    xchg.post(torch.from_numpy(r0["ctrl"].reshape(hi - lo, -1)), torch.from_numpy(r0["goal"]), torch.from_numpy(r0["status"].reshape(-1, 1)))
    own_plans, own_agents = _stress_advance(cfg, w["plans"], w["agents"], r0, own)
    # step 1, interior agents first: the table holds only this rank's new plans, the remote part has NOT arrived (poison)
    plans1 = np.full_like(w["plans"], 1.0e6)
    goals1 = np.full_like(w["goals"], -1.0e6)
    agents1 = w["agents"].copy()
    plans1[lo:hi], goals1[lo:hi], agents1[lo:hi] = own_plans, own_agents["current_goal_point"], own_agents
    res_i = _stress_plan(cfg, w, lo + np.asarray(interior, np.int64), plans1, goals1, agents1)
    # the exchange has completed: the whole table (fail-safe for agents whose QP failed: their shifted previous plan)
    ctrl_all, goal_all, st_all = xchg.unpack(xchg.table(), (torch.float32, torch.float32, torch.int32))
    ctrl_all = ctrl_all.numpy().reshape(N, cfg.M, cfg.n + 1, 3)
    st_all = st_all.numpy().reshape(-1)
    shifted = np.concatenate([w["plans"][:, 1:], np.repeat(w["plans"][:, -1:, -1:], cfg.n + 1, axis=2)], axis=1)
    plans1 = np.where((st_all == 0)[:, None, None, None], ctrl_all, shifted).astype(np.float32)
    # a failed goal LP keeps the previous sub-goal; the LP verdict does not travel, but the posted sub-goal of such an agent is its
    # old one already (dlscqp_advance_device writes current_goal before the exchange packs it): emulate by the same rule
    goals1 = goal_all.numpy().copy()
    res_h = _stress_plan(cfg, w, lo + np.asarray(halo, np.int64), plans1, goals1, agents1)
    out = {k: np.zeros((hi - lo,) + res_i[k].shape[1:], res_i[k].dtype) for k in ("status", "goal_status", "goal", "ctrl")}
    for k in out:
        out[k][interior] = res_i[k]
        out[k][halo] = res_h[k]
    q.put((rank, lo, hi, len(interior), len(halo), out))
    dist.barrier()
    dist.destroy_process_group()


def test_stress_batch_interior_then_halo_over_two_ranks_equals_one_process():
    """DESIGN.md section 5 on the stress workload (every shard cuts through the crowd): two gloo ranks plan step 0, exchange
    their plans with ONE all-gather, and plan step 1 in two parts -- the interior agents before the exchange has been
    awaited (the remote part of their plan table is poison), the halo agents after it.  Bit-identical to one process that
    holds the complete table all along."""
    cfg, w = _stress_inputs()
    N = len(w["agents"])
    allids = np.arange(N)
    r0 = _stress_plan(cfg, w, allids, w["plans"], w["goals"], w["agents"])
    plans1, agents1 = _stress_advance(cfg, w["plans"], w["agents"], r0, allids)
    # the sub-goal table after step 0 as the exchange carries it: the LP's result (kept as is for the comparison)
    ref = _stress_plan(cfg, w, allids, plans1, r0["goal"], agents1)
    assert (ref["status"] == 0).mean() > 0.5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_stress_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, lo, hi, n_int, n_halo, out in got:
        assert n_int > 5 and n_halo > 5, (rank, n_int, n_halo)
        for k in ("status", "goal_status", "goal", "ctrl"):
            assert np.array_equal(out[k], ref[k][lo:hi]), (rank, k)


# ---- the moving crowd over two gloo ranks: rank-local neighbour lists of fixed pitch -------------------------------------
PITCH = 64


def _crowd_inputs(n_agents=96):
    from dlsc_gc_planner_amd import workloads
    cfg = workloads.x2048_config()
    w = workloads.x2048(cfg, n_agents=n_agents, seed=5002)
    return cfg, w


def _crowd_plan(cfg, w, ids, plans, goals, agents):
    """One step of the agents `ids` as the device loop does it: communication-range neighbour lists of fixed pitch from the
    positions of EVERYBODY's previous plans (the stand-in of dlscqp_neighbours_device), half-spaces, goal LP + QP."""
    import emul_lib as E
    ids = np.asarray(ids, np.int32)
    pos = plans[:, 1, 0]  # where every agent is at the start of this step: its previous plan at dt
    counts, nb, _, over = E.neighbours(cfg, pos, PITCH, local_ids=ids)
    assert over == 0
    per = cfg.M * (cfg.n + 1) - cfg.phi
    ro, rows = E.generate_lsc(cfg, plans, goals, w["radius"], w["downwash"], np.arange(len(ids) + 1) * PITCH, nb.ravel(), local_ids=ids,
                              nbr_counts=counts)
    kept = np.concatenate([rows[ro[k]:ro[k] + counts[k] * per] for k in range(len(ids))])
    ro2 = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * per)])
    return E.plan_batch(cfg, agents[ids], ro2, kept, w["sfc"][ids], 0, True), counts


def _crowd_worker(rank, world, port, steps, q):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, w = _crowd_inputs()
    N = len(w["agents"])
    lo, hi = sharding.shard_range(N, rank, world)
    own = np.arange(lo, hi)
    E3 = cfg.M * (cfg.n + 1) * 3
    xchg = sharding.PlanExchange(dist, world, hi - lo, (E3, 3, 1), torch.device("cpu"))
    plans, goals, agents = w["plans"].copy(), w["goals"].copy(), w["agents"].copy()
    n_remote = 0
    for _ in range(steps):
        res, counts = _crowd_plan(cfg, w, own, plans, goals, agents)
        new_plans, new_agents = _stress_advance(cfg, plans, agents, res, own)
        xchg.post(torch.from_numpy(new_plans.reshape(hi - lo, -1)), torch.from_numpy(new_agents["current_goal_point"].copy()),
                  torch.from_numpy(res["status"].reshape(-1, 1)))
        ctrl_all, goal_all, _ = xchg.unpack(xchg.table(), (torch.float32, torch.float32, torch.int32))
        plans = ctrl_all.numpy().reshape(N, cfg.M, cfg.n + 1, 3).copy()
        goals = goal_all.numpy().copy()
        agents[lo:hi] = new_agents
    q.put((rank, plans, goals))
    dist.barrier()
    dist.destroy_process_group()


def test_moving_crowd_with_rank_local_neighbour_lists_over_two_ranks_equals_one_process():
    """BASELINE.json configs[4] as the device loop flies it (neighbour lists rebuilt every step from everybody's positions,
    fixed pitch, ascending ids), three steps, agents sharded over two gloo ranks with one plan exchange per step: the table
    of plans and sub-goals on either rank equals the single-process run bit for bit -- the lists of a shard are the rows of
    the global lists."""
    steps = 3
    cfg, w = _crowd_inputs()
    N = len(w["agents"])
    allids = np.arange(N)
    plans, goals, agents = w["plans"].copy(), w["goals"].copy(), w["agents"].copy()
    for _ in range(steps):
        res, counts = _crowd_plan(cfg, w, allids, plans, goals, agents)
        plans, agents = _stress_advance(cfg, plans, agents, res, allids)
        goals = agents["current_goal_point"].copy()
    assert counts.min() >= 3 and counts.max() < PITCH
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_crowd_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res2 = {r: (a, b) for r, a, b in (q.get(timeout=600) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(res2[r][0], plans) and np.array_equal(res2[r][1], goals), r
