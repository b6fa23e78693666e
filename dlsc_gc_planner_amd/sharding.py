"""Multi-GPU layout of one replanning step: agents sharded by contiguous id range, one all-gather.

The reference plans agents one after another in a single process and "broadcasts" every agent's
plan to every other by a copy loop (MultiSyncSimulator::broadcastMsgs,
/root/reference/src/multi_sync_simulator.cpp:468-514, payload AgentManager::getAgent,
agent_manager.cpp:249-263).  Here rank g owns agents [g*N/G, (g+1)*N/G), solves them with one
kernel launch, and ONE collective (torch.distributed all_gather_into_tensor = RCCL all-gather over
xGMI on GPUs, gloo in the CPU tests) rebuilds the full table of solved control points on every
rank.  No other communication exists on the path.
"""


def shard_range(n_agents, rank, world):
    """Contiguous id range of `rank`; the first n_agents % world ranks own one agent more."""
    base, rem = divmod(n_agents, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_agents, world):
    return [shard_range(n_agents, r, world)[1] - shard_range(n_agents, r, world)[0] for r in range(world)]


def gather_plans(local_ctrl, n_agents, dist, rank, world):
    """All-gather the [n_local, M, n+1, 3] control points of every rank into [n_agents, M, n+1, 3].

    Shards may differ by one agent, so every rank pads to the largest shard and ONE
    all_gather_into_tensor moves the fused buffer; the pad rows are dropped afterwards."""
    import torch
    sizes = shard_sizes(n_agents, world)
    pad = max(sizes)
    tail = tuple(local_ctrl.shape[1:])
    send = local_ctrl
    if local_ctrl.shape[0] != pad:
        send = torch.zeros((pad,) + tail, dtype=local_ctrl.dtype, device=local_ctrl.device)
        send[:local_ctrl.shape[0]] = local_ctrl
    recv = torch.empty((world * pad,) + tail, dtype=local_ctrl.dtype, device=local_ctrl.device)
    dist.all_gather_into_tensor(recv, send.contiguous())
    if all(s == pad for s in sizes):
        return recv
    parts = [recv[r * pad:r * pad + sizes[r]] for r in range(world)]
    return torch.cat(parts, 0)
