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


def halo_agents(nbr_offsets, nbr_ids, lo, hi):
    """Local indices (0-based within the shard [lo, hi)) of the agents that have a neighbour on another rank.

    Agent a's replanning step k+1 reads the step-k plans of its neighbours only (the agents within communication
    range, multi_sync_simulator.cpp:483-492).  For an *interior* agent those are all local, so its half-spaces can be
    generated from the rank's own output of step k; only the *halo* agents returned here have to wait for the
    gathered table.  Shards that are unions of whole swarms (bench.py's S64 workload) have no halo."""
    import numpy as np
    ids = np.asarray(nbr_ids, dtype=np.int64)
    off = np.asarray(nbr_offsets, dtype=np.int64)
    remote = np.concatenate([[0], np.cumsum((ids < lo) | (ids >= hi))])
    return [int(a) for a in np.nonzero(remote[off[1:]] - remote[off[:-1]])[0]]


class PlanExchange:
    """The per-step all-gather of solved control points, posted asynchronously and double-buffered.

    post(local_ctrl) snapshots the rank's [n_local, M, n+1, 3] output of step k and starts ONE
    all_gather_into_tensor of it on a communication stream (RCCL over xGMI; gloo in the CPU tests), so the exchange of
    step k runs while the launch of step k+1 plans the interior agents; table() makes the current stream (CUDA) or the
    host (CPU) wait for the most recent exchange and returns the gathered [world * n_local, ...] table -- what the halo
    agents of step k+1, the host log and any consumer of other ranks' plans read.  Two buffer pairs: the exchange of
    step k must be complete before step k+2 is posted, which the class enforces itself."""

    def __init__(self, dist, world, n_local, tail, dtype, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.cuda = device.type == "cuda"
        self.send = [torch.empty((n_local,) + tuple(tail), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [torch.empty((world * n_local,) + tuple(tail), dtype=dtype, device=device) for _ in range(2)]
        self.done = [None, None]  # CUDA: event on the communication stream; CPU: Work handle
        self.comm = torch.cuda.Stream(device) if self.cuda else None
        self.k = 0
        self.last = None

    def post(self, local_ctrl):
        torch = self.torch
        b = self.k & 1
        self.k += 1
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            if self.done[b] is not None:
                cur.wait_event(self.done[b])  # exchange k-2 has read send[b] and filled recv[b]
            self.send[b].copy_(local_ctrl)
            ready = cur.record_event()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ready)
                self.dist.all_gather_into_tensor(self.recv[b], self.send[b])
                self.done[b] = self.comm.record_event()
        else:
            if self.done[b] is not None:
                self.done[b].wait()
            self.send[b].copy_(local_ctrl)
            self.done[b] = self.dist.all_gather_into_tensor(self.recv[b], self.send[b], async_op=True)
        self.last = b

    def table(self):
        b = self.last
        if self.cuda:
            self.torch.cuda.current_stream(self.device).wait_event(self.done[b])
        else:
            self.done[b].wait()
        return self.recv[b]
