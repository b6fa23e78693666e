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
