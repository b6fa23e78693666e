"""Multi-GPU layout of one replanning step: agents sharded by contiguous id range, one all-gather.

The reference plans agents one after another in a single process and "broadcasts" every agent's
plan to every other by a copy loop (MultiSyncSimulator::broadcastMsgs,
/root/reference/src/multi_sync_simulator.cpp:468-514, payload AgentManager::getAgent,
agent_manager.cpp:249-263).  Here rank g owns agents [g*N/G, (g+1)*N/G), solves them with one
kernel launch, and ONE collective (torch.distributed all_gather_into_tensor = RCCL all-gather over
xGMI on GPUs, gloo in the CPU tests) rebuilds the full table of solved control points on every
rank.  No other communication exists on the path.
"""
import os
import time


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
    """The per-step all-gather of the solved plans, posted asynchronously and double-buffered.

    post(ctrl, goal, status) packs the rank's output of step k -- per agent the dim*M*(n+1) float32 control points, the
    float32 sub-goal and the int32 status: what AgentManager::getAgent broadcasts (agent_manager.cpp:249-263) and what the
    other ranks' half-space generators need -- into one [n_local, width] buffer and starts ONE all_gather_into_tensor of
    it on a communication stream (RCCL over xGMI; gloo in the CPU tests), so the exchange of step k runs while the launch
    of step k+1 plans the interior agents; table() makes the current stream (CUDA) or the host (CPU) wait for the most
    recent exchange and returns the gathered [world * n_local, width] table (unpack() splits it) -- what the halo agents of
    step k+1, the host log and any consumer of other ranks' plans read.  Two buffer pairs: the exchange of step k must
    be complete before step k+2 is posted, which the class enforces itself.
    post(..., pack_on_comm=True) (CUDA) also moves the packing to the communication stream, so the compute stream gets
    nothing but an event; the caller then keeps the posted tensors untouched until it has called before_overwrite()
    ahead of the launch after next (DeviceBatch.flip_outputs double-buffers the outputs for that)."""

    def __init__(self, dist, world, n_local, widths, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.cuda = device.type == "cuda"
        self.widths = tuple(int(w) for w in widths)
        width = sum(self.widths)
        self.send = [torch.empty((n_local, width), dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = [torch.empty((world * n_local, width), dtype=torch.float32, device=device) for _ in range(2)]
        self.done = [None, None]  # CUDA: event on the communication stream; CPU: Work handle
        self.packed = [None, None]  # CUDA, pack_on_comm: the inputs of post k have been read
        self._reuse = [None, None]
        self._src_read = [None] * 8  # post k: "its tensors have been read" (before_overwrite with a rotation of tables)
        self.comm = torch.cuda.Stream(device) if self.cuda else None
        self.k = 0
        self.last = None
        self._ev = None  # enable_timing(): per exchange (ready, packed, gathered) events; per table(): (before, after) the wait

    def enable_timing(self, on=True):
        """CUDA: record events around the pack, the all-gather and the consumer's wait of every following exchange;
        times() returns what they measured (a diagnosis of where a multi-GPU step's time goes, not a product path)."""
        self._ev = {"post": [], "wait": []} if (on and self.cuda) else None

    def times(self):
        """dict of float arrays in ms, one entry per exchange: 'pack' (packing the rank's output), 'all_gather' (the
        collective on the communication stream, from the end of the pack), 'lag' (from the compute stream's 'output ready'
        event to the end of the collective: what an un-overlapped exchange costs), 'wait' (how long the consumer's stream
        really stood still in table())."""
        import numpy as np
        if not self._ev:
            return {}
        self.torch.cuda.synchronize(self.device)
        post, wait = self._ev["post"], self._ev["wait"]
        out = {"pack": np.array([a.elapsed_time(b) for a, b, c in post]),
               "all_gather": np.array([b.elapsed_time(c) for a, b, c in post]),
               "lag": np.array([a.elapsed_time(c) for a, b, c in post]),
               "wait": np.array([a.elapsed_time(b) for a, b in wait])}
        return out

    def _pack(self, b, parts):
        torch = self.torch
        flat = []
        for p, w in zip(parts, self.widths):
            p = p.reshape(p.shape[0], -1)
            if p.dtype != torch.float32:
                p = p.view(torch.float32)  # int32 status travels as raw bits
            assert p.shape[1] == w
            flat.append(p)
        torch.cat(flat, dim=1, out=self.send[b])

    def before_overwrite(self, stream=None, tables=None):
        """pack_on_comm: make `stream` (default: the current one) wait until the tensors of the post before last have been read --
        or, for a caller that rotates through `tables` (3..8) output tables, of the post that read the table the next step writes
        (post k - tables: include/dlscqp.h, dlscqp_exchange_before_overwrite_rotating)."""
        if tables is not None and tables > 2:
            assert tables <= len(self._src_read)
            k = self.k - tables
            e = self._src_read[k % len(self._src_read)] if k >= 0 else None
        else:
            e = self.packed[self.k & 1]
        if e is not None:
            (self.torch.cuda.current_stream(self.device) if stream is None else stream).wait_event(e)

    def post(self, *parts, pack_on_comm=False, stream=None):
        """stream: the stream the parts are ready on (default: the current one)."""
        torch = self.torch
        b = self.k & 1
        self.k += 1
        timed = self._ev is not None
        ev = (lambda: torch.cuda.Event(enable_timing=True)) if timed else (lambda: torch.cuda.Event())
        if self.cuda and pack_on_comm:
            if timed:
                ready, t0, self.packed[b], self.done[b] = ev(), ev(), ev(), ev()
            else:  # untimed: the events of exchange k-2 are recorded again (an event object per exchange costs the host ~10 us)
                if self._reuse[b] is None:
                    self._reuse[b] = (ev(), ev(), ev())
                ready, self.packed[b], self.done[b] = self._reuse[b]
                t0 = None
            ready.record(torch.cuda.current_stream(self.device) if stream is None else stream)
            with torch.cuda.stream(self.comm):  # in order behind exchange k-2, which used the same buffers
                self.comm.wait_event(ready)
                if timed:
                    t0.record(self.comm)
                self._pack(b, parts)
                self.packed[b].record(self.comm)
                slot = (self.k - 1) % len(self._src_read)
                if self._src_read[slot] is None:
                    self._src_read[slot] = torch.cuda.Event()
                self._src_read[slot].record(self.comm)
                self.dist.all_gather_into_tensor(self.recv[b], self.send[b])
                self.done[b].record(self.comm)
            if timed:
                self._ev["post"].append((t0, self.packed[b], self.done[b]))
        elif self.cuda:
            cur = torch.cuda.current_stream(self.device) if stream is None else stream
            if self.done[b] is not None:
                cur.wait_event(self.done[b])  # exchange k-2 has read send[b] and filled recv[b]
            t0 = ev()
            t0.record(cur)
            with torch.cuda.stream(cur):  # (the packing kernel goes where torch's current stream is)
                self._pack(b, parts)
            ready = ev()
            ready.record(cur)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ready)
                self.dist.all_gather_into_tensor(self.recv[b], self.send[b])
                self.done[b] = ev()
                self.done[b].record(self.comm)
            if timed:
                self._ev["post"].append((t0, ready, self.done[b]))
        else:
            if self.done[b] is not None:
                self.done[b].wait()
            self._pack(b, parts)
            self.done[b] = self.dist.all_gather_into_tensor(self.recv[b], self.send[b], async_op=True)
        self.last = b

    def table(self, stream=None):
        b = self.last
        if self.cuda:
            cur = self.torch.cuda.current_stream(self.device) if stream is None else stream
            if self._ev is not None:
                w0 = self.torch.cuda.Event(enable_timing=True)
                w0.record(cur)
            cur.wait_event(self.done[b])
            if self._ev is not None:
                w1 = self.torch.cuda.Event(enable_timing=True)
                w1.record(cur)
                self._ev["wait"].append((w0, w1))
        else:
            self.done[b].wait()
        return self.recv[b]

    def unpack(self, table, dtypes=None):
        """Columns of a gathered table as the parts that were posted (dtypes: per part, default float32)."""
        out, o = [], 0
        for i, w in enumerate(self.widths):
            col = table[:, o:o + w]
            if dtypes is not None and dtypes[i] != self.torch.float32:
                col = col.contiguous().view(dtypes[i])
            out.append(col)
            o += w
        return out


def exchange_unique_id(lib, dist=None, rank=0, src=0):
    """The 128 bytes every rank hands to dlscqp_exchange_create: made by rank `src` (dlscqp_exchange_unique_id = ncclGetUniqueId) and
    passed on over the host's own channel -- here torch.distributed's object broadcast (any backend); dist=None: one rank."""
    import ctypes as C
    from . import abi
    buf = (C.c_char * abi.EXCHANGE_ID_BYTES)()
    if dist is None or rank == src:
        rc = lib.dlscqp_exchange_unique_id(C.cast(buf, C.c_void_p))
        if rc != abi.SUCCESS:
            raise RuntimeError("dlscqp_exchange_unique_id failed (%d): %s" % (rc, lib.dlscqp_last_error().decode()))
    if dist is not None and dist.get_world_size() > 1:
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=src)
        buf.raw = box[0]
    return bytes(buf.raw)


class NativeExchange:
    """The per-step exchange of the solved plans issued by the library itself (include/dlscqp.h: dlscqp_exchange_*): ONE
    ncclAllGather from librccl on a communication stream the exchange owns -- no torch.distributed call on the data path, whose
    host cost per collective (~0.08 ms) bounded the step of round 3.  Same record layout and the same post / before_overwrite /
    table / unpack interface as PlanExchange (which stays for the CPU tier's gloo tests); torch only owns the memory of the two
    gathered tables, so that `table()` can hand them out as tensors.
    opt: BatchTrajOptimizer of this rank's device; unique_id: exchange_unique_id(...) (None with world == 1); local_only: world 1
    without librccl (the "collective" is a device copy)."""

    def __init__(self, opt, rank, world, n_local, unique_id=None, local_only=False):
        import ctypes as C
        import torch
        from . import abi
        self.torch, self.C, self.abi, self.opt, self.lib = torch, C, abi, opt, opt.lib
        cfg = opt.cfg
        self.device = torch.device("cuda", cfg.device)
        self.rank, self.world, self.n_local = rank, world, n_local
        E = cfg.M * (cfg.n + 1) * 3
        self.widths = (E, 3, 1)
        self.W = E + 4
        self.recv = [torch.zeros((world * n_local, self.W), dtype=torch.float32, device=self.device) for _ in range(2)]
        c = abi.ExchangeConfig()
        c.rank, c.world, c.n_local, c.local_only = rank, world, n_local, int(bool(local_only))
        self._id = None if unique_id is None else C.create_string_buffer(unique_id, abi.EXCHANGE_ID_BYTES)
        c.unique_id = None if self._id is None else C.cast(self._id, C.c_void_p)
        for b in range(2):
            c.recv_buffers[b] = self.recv[b].data_ptr()
        h = C.c_void_p()
        rc = self.lib.dlscqp_exchange_create(opt.handle, C.byref(c), C.byref(h))
        if rc != abi.SUCCESS:
            raise RuntimeError("dlscqp_exchange_create failed (%d): %s" % (rc, self.lib.dlscqp_last_error().decode()))
        self.handle = h
        self.k = 0
        self.last = None
        self._timed = 0
        self._waits = []

    def close(self):
        if getattr(self, "handle", None):
            self.lib.dlscqp_exchange_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != self.abi.SUCCESS:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.dlscqp_last_error().decode()))

    def _stream(self, stream):
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        return self.C.c_void_p(s.cuda_stream)

    def post(self, ctrl, goal, status, pack_on_comm=True, stream=None):
        """ctrl [n_local, ...] float32, goal [n_local, 3] float32, status [n_local(, 1)] int32, contiguous CUDA tensors, complete on
        `stream` (default: the current one).  The packing runs on the communication stream.  pack_on_comm=True: keep the tensors
        untouched until before_overwrite() ahead of the launch after next; False (PlanExchange's default, what a caller that
        rewrites the tensors right away needs): `stream` waits until the communication stream has read them."""
        torch = self.torch
        assert ctrl.is_contiguous() and goal.is_contiguous() and status.is_contiguous()
        assert ctrl.dtype == torch.float32 and goal.dtype == torch.float32 and status.dtype == torch.int32
        assert ctrl.numel() == self.n_local * self.widths[0] and goal.numel() == self.n_local * 3 and status.numel() == self.n_local
        self._keep = (ctrl, goal, status)
        self._check(self.lib.dlscqp_exchange_post(self.handle, ctrl.data_ptr(), goal.data_ptr(), status.data_ptr(), self._stream(stream)),
                    "dlscqp_exchange_post")
        if not pack_on_comm:
            # the caller goes on to overwrite the three tensors on its stream (PlanExchange packed on that stream): it waits until
            # the communication stream has read them
            self._check(self.lib.dlscqp_exchange_wait_packed(self.handle, self._stream(stream)), "dlscqp_exchange_wait_packed")
        self.last = self.k & 1
        self.k += 1

    def post_records(self, records, stream=None):
        """records [n_local, W] float32, contiguous: what the plan kernel wrote with DeviceBatch.use_records -- sent as it lies, no
        packing pass; untouched until before_overwrite() ahead of the launch after next (it then waits for the collective)."""
        assert records.is_contiguous() and records.dtype == self.torch.float32 and records.shape == (self.n_local, self.W)
        self._keep = (records,)
        self._check(self.lib.dlscqp_exchange_post_records(self.handle, records.data_ptr(), self._stream(stream)), "dlscqp_exchange_post_records")
        self.last = self.k & 1
        self.k += 1

    def before_overwrite(self, stream=None, tables=None):
        if tables is not None and tables > 2:
            self._check(self.lib.dlscqp_exchange_before_overwrite_rotating(self.handle, self._stream(stream), int(tables)), "dlscqp_exchange_before_overwrite_rotating")
        else:
            self._check(self.lib.dlscqp_exchange_before_overwrite(self.handle, self._stream(stream)), "dlscqp_exchange_before_overwrite")

    def table(self, stream=None):
        """Make `stream` (default: the current one) wait for the exchange posted last; returns the gathered [world * n_local, W] table."""
        torch = self.torch
        cur = torch.cuda.current_stream(self.device) if stream is None else stream
        if self._timed:  # how long the consumer's stream stands still for the table
            w0 = torch.cuda.Event(enable_timing=True)
            w0.record(cur)
        self._check(self.lib.dlscqp_exchange_wait(self.handle, self._stream(stream), None), "dlscqp_exchange_wait")
        if self._timed:
            w1 = torch.cuda.Event(enable_timing=True)
            w1.record(cur)
            self._waits.append((w0, w1))
        return self.recv[self.last]

    def unpack(self, table, dtypes=None):
        out, o = [], 0
        for i, w in enumerate(self.widths):
            col = table[:, o:o + w]
            if dtypes is not None and dtypes[i] != self.torch.float32:
                col = col.contiguous().view(dtypes[i])
            out.append(col)
            o += w
        return out

    def enable_timing(self, on=True, capacity=4096):
        self._timed = capacity if on else 0
        self._waits = []
        self._check(self.lib.dlscqp_exchange_set_timing(self.handle, self._timed), "dlscqp_exchange_set_timing")

    def times(self):
        """ms per exchange: 'pack', 'all_gather', 'lag' (plans ready on the compute side -> table gathered); 'wait': how long the
        stream given to table() stood still for the gathered table (one entry per table() call)."""
        import numpy as np
        C = self.C
        if not self._timed:
            return {}
        self.torch.cuda.synchronize(self.device)
        cap = self._timed
        p, a, l = (np.zeros(cap, np.float32) for _ in range(3))
        n = C.c_int32(0)
        self._check(self.lib.dlscqp_exchange_get_times(self.handle, p.ctypes.data, a.ctypes.data, l.ctypes.data, cap, C.byref(n)), "dlscqp_exchange_get_times")
        return {"pack": p[:n.value].astype(float), "all_gather": a[:n.value].astype(float), "lag": l[:n.value].astype(float),
                "wait": np.array([w0.elapsed_time(w1) for w0, w1 in self._waits])}


def make_exchange(opt, dist, rank, world, n_local, device):
    """The exchange a sharded loop uses: the library's own (NativeExchange: RCCL called from C) on GPUs with an RCCL-capable group or
    no group at all; PlanExchange over torch.distributed otherwise (gloo: the CPU tier, and the two-ranks-on-one-device test hook).
    The native exchange is CHECKED before it is handed out: every rank posts a table filled with its rank number and looks for
    every rank's number in the gathered one; the ranks agree on the outcome (one all-reduce), and if any of them could not build or
    verify its exchange all of them fall back to torch.distributed -- a multi-GPU run must not depend on a path that no one-GPU
    box can exercise with more than one rank.  `exchange.issued_by` says which one it is.
    The library's communicator is its own, beside torch.distributed's: while an exchange is in flight no torch.distributed collective
    may be issued on another stream in an order that differs between the ranks (bench.py and harness.ShardedLoop issue none inside
    their timed loops).  DLSCQP_NATIVE_EXCHANGE=0 keeps torch.distributed; DLSCQP_EXCHANGE_WATCHDOG_S bounds the attempt (90 s)."""
    import sys
    backend = None if dist is None else dist.get_backend()
    E = opt.cfg.M * (opt.cfg.n + 1) * 3
    native = os.environ.get("DLSCQP_NATIVE_EXCHANGE", "1") != "0"  # (0: torch.distributed, whatever the backend)
    if device.type == "cuda" and backend in (None, "nccl") and (native or dist is None):
        import torch
        x, ok, why = None, 1, ""
        if dist is not None and world > 1:
            # Before anything collective: can EVERY rank reach librccl through the library?  (A rank that cannot would leave the
            # others waiting -- in the broadcast of the id if it is rank 0, in ncclCommInitRank otherwise.)  Each rank makes an id of
            # its own, throws it away, and the ranks agree.
            try:
                exchange_unique_id(opt.lib, None, rank)
            except Exception as e:  # noqa: BLE001
                ok, why = 0, str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                print("dlsc_gc_planner_amd: the library's own exchange is not used (%s): torch.distributed instead" % (why or "another rank cannot load librccl"),
                      file=sys.stderr)
                x = PlanExchange(dist, world, n_local, (E, 3, 1), device)
                x.issued_by = "torch.distributed"
                return x
        # Building the communicator (ncclCommInitRank) and the first collective run under a WATCHDOG: a rank that gets stuck inside
        # RCCL (its peers then wait in the same call) must cost the run a minute, not the run -- after `budget` seconds the attempt
        # counts as failed on this rank, the ranks agree below, and the stuck call is left behind on its helper thread.
        import threading
        box = {}
        cancelled = threading.Event()  # set when the watchdog gives the attempt up: the helper thread then touches neither the GPU nor RCCL again

        def attempt():
            try:
                torch.cuda.set_device(device)  # (the current device is per thread)
                uid = exchange_unique_id(opt.lib, dist, rank) if world > 1 else None
                if cancelled.is_set():
                    return
                try:
                    xx = NativeExchange(opt, rank, world, n_local, uid)
                except Exception:  # noqa: BLE001
                    if dist is not None or world > 1:
                        raise
                    xx = NativeExchange(opt, rank, world, n_local, None, local_only=True)  # one rank, no librccl: a device copy
                box["x"] = xx
                if cancelled.is_set():  # a slow ncclCommInitRank came back after all: no collective on the library's communicator now,
                    return              # the run is on torch.distributed's (two communicators posting side by side can deadlock the ranks)
                probe = torch.full((n_local, E + 4), float(rank + 1), dtype=torch.float32, device=device)
                xx.post_records(probe)
                if cancelled.is_set():
                    return
                got = xx.table()[:, 0].clone()
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(device))
                box["event"] = (done, got, probe)
            except Exception as e:  # noqa: BLE001 -- whatever went wrong, the run goes on over torch.distributed
                box["error"] = str(e)

        budget = float(os.environ.get("DLSCQP_EXCHANGE_WATCHDOG_S", "90"))
        deadline = time.time() + budget  # ONE budget for the thread and for the first collective's completion
        th = threading.Thread(target=attempt, daemon=True)
        th.start()
        th.join(budget)
        if th.is_alive():
            cancelled.set()
            # the thread may still be inside the library with `opt`'s context: the optimizer must outlive it (BatchTrajOptimizer.close)
            opt.busy_threads = getattr(opt, "busy_threads", []) + [th]
            ok, why = 0, "no answer from RCCL within %.0f s (ncclCommInitRank / first ncclAllGather)" % budget
        elif "error" in box:
            ok, why = 0, box["error"]
        else:
            x = box["x"]
            done, got, _probe = box["event"]
            t_end = max(deadline, time.time() + 1.0)
            while not done.query() and time.time() < t_end:  # (never a blocking wait on a collective that may not come back)
                time.sleep(0.002)
            if not done.query():
                ok, why, x = 0, "the first ncclAllGather did not complete within %.0f s" % budget, None  # (left behind, not destroyed: that would wait too)
            else:
                want = (torch.arange(world, device=device, dtype=torch.float32) + 1.0).repeat_interleave(n_local)
                if not torch.equal(got, want):
                    ok, why = 0, "the gathered table does not hold every rank's block"
        if dist is not None and world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item()) if ok else 0
        if ok:
            x.issued_by = "library"
            return x
        if x is not None:
            x.close()
        print("dlsc_gc_planner_amd: the library's own exchange is not used (%s): torch.distributed instead" % (why or "another rank failed"), file=sys.stderr)
    if dist is None:
        raise RuntimeError("no process group and no native exchange: %s" % why)
    x = PlanExchange(dist, world, n_local, (E, 3, 1), device)
    x.issued_by = "torch.distributed"
    return x
