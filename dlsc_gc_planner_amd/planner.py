"""Python host-side mirror of the reference's optimizer interface, over the C ABI.

`BatchTrajOptimizer` plays the role of MATP::TrajOptimizer + MATP::GoalOptimizer
(/root/reference/include/traj_optimizer.hpp:24-33, goal_optimizer.hpp:17-24) for a whole batch of
agents: same inputs (Agent, the LSC/SFC content of CollisionConstraints), same outputs
(control points as float32 point3d, total_qp_cost, slack costs, goal point), and the same failure
convention (a per-agent verdict that the caller maps to PlanningReport::QPFAILED).

All arithmetic happens in the HIP library (csrc/libdlscqp.so); there is no CPU path here.
torch is used only as plumbing for device-resident buffers and streams.
"""
import ctypes as C
import os

import numpy as np

from . import abi


class DlscqpError(RuntimeError):
    pass


class QPFailed(RuntimeError):
    """PlanningReport::QPFAILED (traj_optimizer.cpp:152, goal_optimizer.cpp:122)."""


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class BatchTrajOptimizer:
    def __init__(self, cfg):
        # torch bundles its own HIP runtime; when both are used in one process torch has to initialise
        # it first (the other order leaves torch with "No HIP GPUs are available")
        try:
            import torch
            torch.cuda.is_available()
        except ImportError:
            pass
        self.lib = abi.load_library()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.dlscqp_create(C.byref(cfg), C.byref(h))
        if rc != abi.SUCCESS:
            raise DlscqpError("dlscqp_create failed (%d): %s" % (rc, self.lib.dlscqp_last_error().decode()))
        self.handle = h

    def close(self):
        # a helper thread that was left behind inside the library with this context (sharding.make_exchange's watchdog) must not find
        # it destroyed: the context then stays (a leak of one context in a run that has already lost its native exchange)
        if any(t.is_alive() for t in getattr(self, "busy_threads", [])):
            return
        if getattr(self, "handle", None):
            self.lib.dlscqp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != abi.SUCCESS:
            raise DlscqpError("%s failed (%d): %s" % (what, rc, self.lib.dlscqp_last_error().decode()))

    def update_param(self, cfg):
        """TrajOptimizer::updateParam (traj_optimizer.cpp:167-169)."""
        self._check(self.lib.dlscqp_update_param(self.handle, C.byref(cfg)), "dlscqp_update_param")
        self.cfg = cfg

    # ---- the reference's failure path on the host: model export, conflict (traj_optimizer.cpp:45-52, 113-146) -----------------------
    def write_lp(self, agent, rows, path, sfc=None, max_slack_obs=0, goal=None):
        """The agent's model as a CPLEX-LP file (cplex.exportModel): rows in populatebyrow's order, named c0, c1, ..."""
        agent = np.ascontiguousarray(agent, dtype=abi.AGENT_DTYPE).reshape(1)
        rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
        sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
        goal = None if goal is None else np.ascontiguousarray(goal, dtype=np.float32)
        self._check(self.lib.dlscqp_write_lp(self.handle, _ptr(agent), _ptr(rows) if len(rows) else None, len(rows), None if sfc is None else _ptr(sfc),
                                             max_slack_obs, None if goal is None else _ptr(goal), str(path).encode()), "dlscqp_write_lp")

    def conflict_decode(self, agent, rows, record, sfc=None, max_slack_obs=0, goal=None):
        """One record of DeviceBatch.conflicts() -> (members, n_con, gap): members = list of dicts {ref_index, kind, var, coef, lo, hi,
        lam} -- rows of the reference's model, in its scaling, numbered like the conflict refiner's `infeas` array."""
        agent = np.ascontiguousarray(agent, dtype=abi.AGENT_DTYPE).reshape(1)
        rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
        sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
        goal = None if goal is None else np.ascontiguousarray(goal, dtype=np.float32)
        record = np.ascontiguousarray(record, dtype=np.float64)
        cap = max(int(record[0]), 1)
        out = (abi.ConflictMember * cap)()
        n, ncon, gap = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        self._check(self.lib.dlscqp_conflict_decode(self.handle, _ptr(agent), _ptr(rows) if len(rows) else None, len(rows),
                                                    None if sfc is None else _ptr(sfc), max_slack_obs, None if goal is None else _ptr(goal),
                                                    _ptr(record), out, cap, C.byref(n), C.byref(ncon), C.byref(gap)), "dlscqp_conflict_decode")
        mem = [dict(ref_index=int(m.ref_index), kind=abi.MODEL_KINDS[m.kind], var=[int(m.var[t]) for t in range(m.n)],
                    coef=[float(m.coef[t]) for t in range(m.n)], lo=float(m.lo), hi=float(m.hi), lam=float(m.lam)) for m in out[:n.value]]
        return mem, int(ncon.value), float(gap.value)

    def fused_step(self, batch, gen, nbr=None, sfc=None, trap=False, stream=None):
        """dlscqp_fused_step_device: the steady-state step of a small mission as ONE launch.  `batch`: DeviceBatch (with its
        advance_after), `gen`: LscGenerator, `nbr`: NeighbourSearch or None, `sfc`: SfcGenerator or None -- each prepared with
        launch=False --, trap: the generator's waypoint trap (prepared the same way).  Returns False, with nothing launched, where
        the library has no fused build for the case (the caller then makes the calls one by one)."""
        torch = batch.torch
        s = torch.cuda.current_stream(batch.device) if stream is None else stream
        batch.c.do_goal, batch.c.do_traj, batch.c.launch_offset, batch.c.launch_count = 1, 1, 0, 0
        rc = self.lib.dlscqp_fused_step_device(self.handle, None if nbr is None else C.byref(nbr.c), None if sfc is None else sfc.world,
                                               None if sfc is None else C.byref(sfc.c), C.byref(gen.c),
                                               C.byref(gen.trap_c) if trap else None, C.byref(batch.c), C.c_void_p(s.cuda_stream))
        if rc == abi.ERR_UNSUPPORTED:
            return False
        self._check(rc, "dlscqp_fused_step_device")
        return True

    def lds_bytes(self, max_slack_obs=0, max_rows=0):
        return int(self.lib.dlscqp_lds_bytes(self.handle, max_slack_obs, max_rows))

    # ---- host-buffer entry points -------------------------------------------------------------
    def plan(self, agents, row_offsets, rows, sfc=None, max_slack_obs=0, do_goal=True, do_traj=True):
        """Goal LP and/or trajectory QP for a batch held in host (numpy) memory."""
        agents = np.ascontiguousarray(agents, dtype=abi.AGENT_DTYPE)
        rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
        row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int64)
        sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
        N = len(agents)
        M, n = self.cfg.M, self.cfg.n
        out = dict(goal=np.zeros((N, 3), np.float32), goal_status=np.zeros(N, np.int32),
                   ctrl=np.zeros((N, M, n + 1, 3), np.float32), cost=np.zeros(N),
                   slack_cost=np.zeros((N, max(max_slack_obs, 1))), status=np.full(N, -1, np.int32))
        if do_goal and do_traj:
            rc = self.lib.dlscqp_plan_batch(self.handle, N, _ptr(agents), _ptr(row_offsets), _ptr(rows), _ptr(sfc),
                                            max_slack_obs, _ptr(out["goal"]), _ptr(out["goal_status"]),
                                            _ptr(out["ctrl"]), _ptr(out["cost"]), _ptr(out["slack_cost"]),
                                            _ptr(out["status"]))
            self._check(rc, "dlscqp_plan_batch")
        elif do_traj:
            rc = self.lib.dlscqp_solve_batch(self.handle, N, _ptr(agents), _ptr(row_offsets), _ptr(rows), _ptr(sfc),
                                             max_slack_obs, _ptr(out["ctrl"]), _ptr(out["cost"]),
                                             _ptr(out["slack_cost"]), _ptr(out["status"]))
            self._check(rc, "dlscqp_solve_batch")
        else:
            rc = self.lib.dlscqp_goal_batch(self.handle, N, _ptr(agents), _ptr(row_offsets), _ptr(rows), _ptr(sfc),
                                            _ptr(out["goal"]), _ptr(out["goal_status"]))
            self._check(rc, "dlscqp_goal_batch")
        return out

    def solve(self, agent, rows, sfc=None):
        """TrajOptimizer::solve for one agent: returns (ctrl [M][n+1][3] float32, total_qp_cost,
        slack_cost) or raises QPFailed like the reference (traj_optimizer.cpp:152)."""
        agent = np.ascontiguousarray(agent, dtype=abi.AGENT_DTYPE).reshape(1)
        nso = int(agent["n_slack_obs"][0])
        r = self.plan(agent, np.array([0, len(rows)], np.int64), rows,
                      None if sfc is None else np.asarray(sfc, np.float32)[None], nso, do_goal=False)
        if r["status"][0] != abi.OK:
            raise QPFailed("QPFAILED (verdict %d)" % r["status"][0])
        return r["ctrl"][0], float(r["cost"][0]), r["slack_cost"][0, :nso]

    def solve_goal(self, agent, rows, sfc=None):
        """GoalOptimizer::solve for one agent (goal_optimizer.cpp:7-136)."""
        agent = np.ascontiguousarray(agent, dtype=abi.AGENT_DTYPE).reshape(1)
        r = self.plan(agent, np.array([0, len(rows)], np.int64), rows,
                      None if sfc is None else np.asarray(sfc, np.float32)[None], 0, do_goal=True, do_traj=False)
        if r["goal_status"][0] == abi.INFEASIBLE:
            raise QPFailed("QPFAILED (goal LP infeasible)")
        return r["goal"][0]

    # ---- device-resident entry point ------------------------------------------------------------
    def make_device_batch(self, agents, row_offsets, rows, sfc=None, max_slack_obs=0, want_f64=False,
                          want_iters=False, device=None):
        """Upload a batch once (SoA rows) and allocate its outputs in HBM; returns a DeviceBatch
        that can be launched any number of times without touching the host."""
        return DeviceBatch(self, agents, row_offsets, rows, sfc, max_slack_obs, want_f64, want_iters, device)

    def make_lsc_generator(self, n_total, nbr_offsets, nbr_ids, radius, downwash, local_ids=None, device=None, dyn=None):
        """Device-resident DLSC-GC half-space generator for a fixed neighbour structure; `dyn`
        describes the dynamic obstacles every agent sees (radius, max_acc, downwash, size_prediction, horizon)."""
        return LscGenerator(self, n_total, nbr_offsets, nbr_ids, radius, downwash, local_ids, device, dyn)

    def make_neighbour_search(self, gen, pitch, device=None):
        """Communication-range neighbour lists of `gen`'s local agents, rebuilt on the device every step
        (dlscqp_neighbours_device; `gen` must have been made with nbr_offsets = arange(n_local + 1) * pitch)."""
        return NeighbourSearch(self, gen, pitch, device)

    def make_knn_search(self, gen, group, k, device=None):
        """The synthetic swarms' neighbour rule on the device (dlscqp_neighbours_knn_device): the k nearest agents of an agent's own
        block of `group` consecutive ids; `gen` must have been made with nbr_offsets = arange(n + 1) * k and no local_ids."""
        return KnnSearch(self, gen, group, k, device)

    def make_sfc_generator(self, boxes, resolution, world_min, world_max, radius, max_vel, grid_resolution, device=None):
        """Device-resident safe-flight-corridor generator for a world given as axis-aligned boxes."""
        return SfcGenerator(self, boxes, resolution, world_min, world_max, radius, max_vel, grid_resolution, device)

    def make_advance(self, batch, plans, current_goal, initial_traj=None, waypoints=None, desired_goal=None,
                     waypoint_step=0.0, hold_on_lp_failure=False, boxes=None, box_margin=0.0, box_min=None, box_max=None,
                     host_pack=None, alert_obs=0, alert_threshold=0.0):
        """The bookkeeping between two launches of `batch` as one kernel (dlscqp_advance_device): fail-safe, sub-goal,
        state := plan(dt), the next step's agent records and initial trajectory.  All arguments are CUDA tensors that
        stay where they are (float32; plans / initial_traj [N, M, n+1, 3], current_goal / waypoints / desired_goal
        [N, 3]); returns a callable(stream=None).  boxes [N, M, 6] (with initial_traj): the kernel also writes the next step's
        boxes of a world without obstacles -- bounding box of each segment of initial_traj, box_margin wider, cut to
        [box_min, box_max].  host_pack: a PINNED float32 host tensor of 3 N + 3 N + N M (n+1) 3 + N alert_obs elements the kernel
        writes position | current_goal | plans | alert into (alert: slack cost of obstacle j in the last QP > alert_threshold), for a
        host-side waypoint planner: no device-to-host copy, the host reads it once the stream has run dry."""
        # dlscqp_advance reads dense arrays: not after DeviceBatch.use_records (one strided record per agent)
        assert batch.c.out_record_stride == 0, "make_advance: the batch writes its plans as records (use_records); dlscqp_advance reads dense arrays"
        a = abi.Advance()
        a.n_agents = batch.n_agents
        a.status, a.goal_status = batch.out_status.data_ptr(), batch.out_goal_status.data_ptr()
        a.ctrl, a.goal = batch.out_ctrl.data_ptr(), batch.out_goal.data_ptr()
        for t in (plans, current_goal, initial_traj, waypoints, desired_goal):
            assert t is None or (t.is_contiguous() and t.dtype == self._torch().float32)
        a.plans, a.current_goal = plans.data_ptr(), current_goal.data_ptr()
        a.initial_traj = None if initial_traj is None else initial_traj.data_ptr()
        a.agents = batch.t["agents"].data_ptr()
        a.waypoints = None if waypoints is None else waypoints.data_ptr()
        a.desired_goal = None if desired_goal is None else desired_goal.data_ptr()
        a.waypoint_step = float(waypoint_step)
        a.waypoint_hold_on_lp_failure = int(bool(hold_on_lp_failure))
        if boxes is not None:
            assert initial_traj is not None and boxes.is_contiguous() and boxes.dtype == self._torch().float32
            a.out_boxes = boxes.data_ptr()
            a.box_margin = float(box_margin)
            for k in range(3):
                a.box_min[k], a.box_max[k] = float(box_min[k]), float(box_max[k])
        if host_pack is not None:
            N, E = batch.n_agents, self.cfg.M * (self.cfg.n + 1) * 3
            assert host_pack.is_pinned() and host_pack.dtype == self._torch().float32 and host_pack.is_contiguous()
            assert host_pack.numel() == N * (6 + E + int(alert_obs)), "host_pack: 3 N + 3 N + N M (n+1) 3 + N alert_obs floats"
            a.out_host_pack = host_pack.data_ptr()
            if alert_obs:
                assert batch.out_slack.dtype == self._torch().float64 and batch.out_slack.is_contiguous() and batch.out_slack.shape[1] >= alert_obs
                a.alert_slack_cost, a.alert_slack_stride = batch.out_slack.data_ptr(), int(batch.out_slack.shape[1])
                a.n_alert_obs, a.alert_threshold = int(alert_obs), float(alert_threshold)
        keep = (batch, plans, current_goal, initial_traj, waypoints, desired_goal, boxes, host_pack)
        torch = self._torch()

        def run(stream=None, _keep=keep):
            s = torch.cuda.current_stream(batch.device) if stream is None else stream
            self._check(self.lib.dlscqp_advance_device(self.handle, C.byref(a), C.c_void_p(s.cuda_stream)), "dlscqp_advance_device")
        run.args = a  # (DeviceBatch.set_advance_after: the same bookkeeping in the plan kernel's epilogue instead of a launch of its own)
        return run

    def _torch(self):
        import torch
        return torch

    def set_timing(self, capacity):
        self._check(self.lib.dlscqp_set_timing(self.handle, int(capacity)), "dlscqp_set_timing")

    def kernel_times_ms(self, capacity):
        buf = np.zeros(max(capacity, 1), np.float32)
        n = C.c_int32(0)
        self._check(self.lib.dlscqp_get_kernel_times(self.handle, _ptr(buf), int(capacity), C.byref(n)),
                    "dlscqp_get_kernel_times")
        return buf[:n.value].copy()


class DeviceBatch:
    """A batch resident in HBM (torch tensors as plain device buffers)."""

    def __init__(self, opt, agents, row_offsets, rows, sfc, max_slack_obs, want_f64, want_iters, device):
        import torch
        self.torch = torch
        self.opt = opt
        cfg = opt.cfg
        dev = torch.device("cuda", cfg.device if device is None else device)
        self.device = dev
        agents = np.ascontiguousarray(agents, dtype=abi.AGENT_DTYPE)
        rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
        row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int64)
        N, R = len(agents), len(rows)
        self.n_agents, self.n_rows = N, R
        M, n, dim = cfg.M, cfg.n, cfg.world_dimension
        pad = R + 64
        soa = dict(nx=np.zeros(pad, np.float32), ny=np.zeros(pad, np.float32), nz=np.zeros(pad, np.float32),
                   px=np.zeros(pad, np.float32), py=np.zeros(pad, np.float32), pz=np.zeros(pad, np.float32),
                   d=np.zeros(pad, np.float64), key=np.zeros(pad, np.uint32))
        opt.lib.dlscqp_rows_to_soa(_ptr(rows), R, *[_ptr(soa[k]) for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key")])

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

        self.t = dict(agents=up(agents), row_offsets=up(row_offsets.astype(np.int32)))
        for k, v in soa.items():
            self.t[k] = up(v)
        self.t["sfc"] = None if sfc is None else up(np.ascontiguousarray(sfc, dtype=np.float32))
        self.out_goal = torch.zeros((N, 3), dtype=torch.float32, device=dev)
        self.out_goal_status = torch.zeros(N, dtype=torch.int32, device=dev)
        self.out_ctrl = torch.zeros((N, M, n + 1, 3), dtype=torch.float32, device=dev)
        self.out_cost = torch.zeros(N, dtype=torch.float64, device=dev)
        self.out_status = torch.full((N,), -1, dtype=torch.int32, device=dev)
        self.out_slack = torch.zeros((N, max(max_slack_obs, 1)), dtype=torch.float64, device=dev)
        nxv = dim * M * (n + 1) + M * max_slack_obs
        self.out_x = torch.zeros((N, nxv), dtype=torch.float64, device=dev) if want_f64 else None
        self.out_iters = torch.zeros((N, 4), dtype=torch.int32, device=dev) if want_iters else None
        kmax = int(np.max(np.diff(row_offsets))) if N else 0
        b = abi.DeviceBatch()
        b.n_agents, b.max_slack_obs, b.do_goal, b.do_traj = N, max_slack_obs, 1, 1
        b.max_rows_per_agent = kmax
        b.agents = self.t["agents"].data_ptr()
        b.row_offsets = self.t["row_offsets"].data_ptr()
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(b, k, self.t[k].data_ptr())
        b.sfc_boxes = None if self.t["sfc"] is None else self.t["sfc"].data_ptr()
        b.out_goal = self.out_goal.data_ptr()
        b.out_goal_status = self.out_goal_status.data_ptr()
        b.out_ctrl = self.out_ctrl.data_ptr()
        b.out_ctrl_f64 = None if self.out_x is None else self.out_x.data_ptr()
        b.out_cost = self.out_cost.data_ptr()
        b.out_slack_cost = self.out_slack.data_ptr() if max_slack_obs > 0 else None
        b.out_status = self.out_status.data_ptr()
        b.out_iters = None if self.out_iters is None else self.out_iters.data_ptr()
        b.agent_order = None
        self._row_offsets_host = row_offsets
        self.order = None
        self.c = b
        self.max_rows = kmax
        self.max_slack_obs = max_slack_obs

    def want_conflict(self):
        """dlscqp_device_batch.out_conflict: the following launches also write, for every agent whose trajectory QP is INFEASIBLE, the
        rows that prove it and their multipliers (the reference's conflict refiner, traj_optimizer.cpp:118-146).  A debugging aid:
        the kernel's early infeasibility proof is off meanwhile.  conflicts() returns the records."""
        w = int(self.opt.lib.dlscqp_conflict_doubles(self.opt.handle, self.max_slack_obs))
        self.out_conflict = self.torch.zeros((self.n_agents, w), dtype=self.torch.float64, device=self.device)
        self.c.out_conflict = self.out_conflict.data_ptr()
        self.c.conflict_stride = w
        return w

    def conflicts(self):
        """[n_agents, dlscqp_conflict_doubles()] float64 (host): record a is [members, violation, scale, 0, (row id, multiplier) ...]"""
        self.torch.cuda.synchronize(self.device)
        return self.out_conflict.cpu().numpy()

    def use_workspace(self):
        """A buffer of this batch's own for the launch's spill records (dlscqp_device_batch.workspace): launches of several batches
        of one context may then overlap, and the batch can be captured into a graph without a first eager launch."""
        need = int(self.opt.lib.dlscqp_workspace_bytes(self.opt.handle, self.n_agents, self.max_slack_obs, self.max_rows))
        if need > 0:
            self._workspace = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
            self.c.workspace = self._workspace.data_ptr()
            self.c.workspace_bytes = need
        return need

    def use_row_counts(self, counts):
        """counts: int32 CUDA tensor [n_agents] (LscGenerator.enable_pruning) or None: rows per agent from row_offsets on."""
        self._row_counts = counts
        self.c.row_counts = None if counts is None else counts.data_ptr()

    def set_order(self, order):
        """Launch permutation (workgroup b plans agent order[b]); None restores the identity.  Meant to be the
        agents sorted by the previous replanning step's iteration count, most expensive first."""
        if order is None:
            self.order = None
            self.c.agent_order = None
            return
        order = np.ascontiguousarray(order, dtype=np.int32)
        assert order.shape == (self.n_agents,) and np.array_equal(np.sort(order), np.arange(self.n_agents))
        self.order = self.torch.from_numpy(order).to(self.device)
        self.c.agent_order = self.order.data_ptr()

    def use_row_pitch(self, pitch=None):
        """Rows at a fixed pitch (dlscqp_device_batch.row_pitch): agent a's rows start at a * pitch, so the kernel computes the
        address instead of loading row_offsets[a] -- one dependent load less in front of the rows.  pitch=None: take it from the
        row_offsets the batch was made with (they must be 0, p, 2p, ...); 0 switches it off.  The caller keeps the layout:
        whoever replaces row_offsets afterwards (neighbour lists of varying length) must switch the pitch off."""
        if pitch is None:
            d = np.diff(self._row_offsets_host)
            assert len(d) and self._row_offsets_host[0] == 0 and d[0] > 0 and (d == d[0]).all(), "row_offsets are not at a fixed pitch"
            pitch = int(d[0])
        self.c.row_pitch = int(pitch)

    def set_hint(self, iters):
        """Launch order found by the kernel itself (dlscqp_device_batch.hint_iters): `iters` = the [n_agents, 4] int32 CUDA tensors
        out_iters of up to three previous replanning steps of the same agents, newest first; None / () switches it off.  No sort,
        no stream operation between two launches; replaces set_order / update_order (an explicit order wins if both are set)."""
        iters = [] if iters is None else list(iters)[:3]
        for k in range(3):
            self.c.hint_iters[k] = None
        for k, h in enumerate(iters):
            assert h.dtype == self.torch.int32 and h.numel() == 4 * self.n_agents and h.is_contiguous()
            self.c.hint_iters[k] = h.data_ptr()
        self.c.hint_steps = len(iters)
        self._keep_hint = iters

    def new_hint_masks(self):
        """A zeroed table of class masks for this batch's agents (dlscqp_device_batch.hint_mask_*): [ceil(n / 64), 8] int64."""
        return self.torch.zeros(((self.n_agents + 63) // 64, 8), dtype=self.torch.int64, device=self.device)

    def set_hint_masks(self, mask_in=None, mask_out=None, mask_zero=None):
        """Launch order from class masks (dlscqp_device_batch.hint_mask_in / _out / _zero): `mask_in` = the table the previous step's
        launch of the same agents wrote (None: id order), `mask_out` = the table this launch writes (all zero when it starts),
        `mask_zero` = a table this launch clears for a later launch to write.  Tables from new_hint_masks(); three in rotation."""
        for name, t in (("hint_mask_in", mask_in), ("hint_mask_out", mask_out), ("hint_mask_zero", mask_zero)):
            assert t is None or (t.dtype == self.torch.int64 and t.numel() == 8 * ((self.n_agents + 63) // 64) and t.is_contiguous())
            setattr(self.c, name, None if t is None else t.data_ptr())
        self._keep_masks = (mask_in, mask_out, mask_zero)

    def set_advance_after(self, advance):
        """dlscqp_device_batch.advance_after: `advance` = what make_advance returned for THIS batch (or None to take it back) -- every
        following launch also does the bookkeeping between two launches, in its epilogue; the caller then no longer calls `advance`."""
        self._advance_after = advance  # keep the structure alive
        self.c.advance_after = None if advance is None else C.addressof(advance.args)

    def rotate_hint_masks(self):
        """Closed loops (one batch, launched step after step): call before every launch.  Three tables rotate: the one the previous
        launch wrote orders this launch, this launch writes the one the launch before last cleared, and clears the third."""
        if not hasattr(self, "_mask_ring"):
            self._mask_ring = [self.new_hint_masks() for _ in range(3)]
            self._mask_step = 0
        r, k = self._mask_ring, self._mask_step
        self.set_hint_masks(r[(k - 1) % 3] if k > 0 else None, r[k % 3], r[(k + 1) % 3])
        self._mask_step = k + 1

    def rotate_hint(self, depth=2):
        """Closed loops (one batch, launched step after step): call before every launch.  The iteration counts rotate through
        depth + 1 buffers, the launch writes one of them and takes its order from the `depth` previous steps' -- a launch must
        never read the counts it is writing (its workgroups would disagree about the order)."""
        if getattr(self, "_hint_bufs", None) is None:
            assert self.out_iters is not None, "make_device_batch(..., want_iters=True)"
            self._hint_bufs = [self.out_iters] + [self.torch.zeros_like(self.out_iters) for _ in range(depth)]
            self._hint_k = 0
        k, n = self._hint_k, len(self._hint_bufs)
        cur = self._hint_bufs[k % n]
        self.out_iters = cur
        self.c.out_iters = cur.data_ptr()
        self.set_hint([self._hint_bufs[(k - j) % n] for j in range(1, n) if k - j >= 0])
        self._hint_k += 1

    def update_order(self, stream=None, iters=None, history=()):
        """Launch order of the NEXT launch from the iteration counts of the last one, on the device
        (dlscqp_launch_order_device): the agents that needed the most iterations first.  Two order buffers alternate,
        so the launch that is still reading its order is not disturbed.  `iters`: the [n_agents, 4] int32 counts of
        another batch of the same agents (the previous replanning step's) instead of this batch's own."""
        torch = self.torch
        if iters is None:
            assert self.out_iters is not None, "make_device_batch(..., want_iters=True)"
            iters = self.out_iters
        assert iters.dtype == torch.int32 and iters.numel() == 4 * self.n_agents and iters.is_contiguous()
        if getattr(self, "_orders", None) is None:
            self._orders = [torch.zeros(self.n_agents, dtype=torch.int32, device=self.device) for _ in range(2)]
            self._order_i = 0
        self._order_i ^= 1
        o = self._orders[self._order_i]
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        if history:  # the counts of one or two steps before `iters` ([n_agents, 4] int32 CUDA tensors, newest first)
            hs = [iters] + list(history)[:2]
            for h in hs:
                assert h.dtype == torch.int32 and h.numel() == 4 * self.n_agents and h.is_contiguous()
            ptrs = (C.c_void_p * len(hs))(*[h.data_ptr() for h in hs])
            rc = self.opt.lib.dlscqp_launch_order_history_device(self.opt.handle, ptrs, len(hs), self.n_agents, C.c_void_p(o.data_ptr()),
                                                                 C.c_void_p(s.cuda_stream))
        else:
            rc = self.opt.lib.dlscqp_launch_order_device(self.opt.handle, C.c_void_p(iters.data_ptr()), self.n_agents,
                                                         C.c_void_p(o.data_ptr()), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_launch_order_device")
        self.order = o
        self.c.agent_order = o.data_ptr()

    def flip_outputs(self):
        """Double-buffer the outputs another stream reads (control points, sub-goal, status): after the call `out_ctrl`,
        `out_goal` and `out_status` are the set the NEXT launch writes, and the previous set stays untouched until the
        flip after next -- time for a plan exchange on a communication stream to read it."""
        torch = self.torch
        if not hasattr(self, "_alt"):
            self._alt = (torch.zeros_like(self.out_ctrl), torch.zeros_like(self.out_goal), torch.full_like(self.out_status, -1))
        cur = (self.out_ctrl, self.out_goal, self.out_status)
        self.out_ctrl, self.out_goal, self.out_status = self._alt
        self._alt = cur
        self.c.out_ctrl, self.c.out_goal, self.c.out_status = self.out_ctrl.data_ptr(), self.out_goal.data_ptr(), self.out_status.data_ptr()

    def use_outputs(self, ctrl, goal, status):
        """Write control points, sub-goal and status into the caller's tensors (e.g. slices of one table that several batches
        fill and one plan exchange reads): [N, M, n+1, 3] float32, [N, 3] float32, [N] int32, contiguous."""
        assert ctrl.shape == self.out_ctrl.shape and goal.shape == self.out_goal.shape and status.shape == self.out_status.shape
        assert ctrl.is_contiguous() and goal.is_contiguous() and status.is_contiguous()
        assert ctrl.dtype == self.out_ctrl.dtype and goal.dtype == self.out_goal.dtype and status.dtype == self.out_status.dtype
        self.out_ctrl, self.out_goal, self.out_status = ctrl, goal, status
        self.c.out_ctrl, self.c.out_goal, self.c.out_status = ctrl.data_ptr(), goal.data_ptr(), status.data_ptr()
        self.c.out_record_stride = 0

    def use_records(self, rec):
        """Write control points, sub-goal and status as ONE record per agent (dlscqp_device_batch.out_record_stride): `rec` is a
        float32 CUDA tensor [n_agents, M (n+1) 3 + 4] with contiguous rows (a slice of the table an exchange sends as it lies:
        NativeExchange.post_records).  out_ctrl / out_goal / out_status become views of it."""
        torch = self.torch
        cfg = self.opt.cfg
        E = cfg.M * (cfg.n + 1) * 3
        assert rec.dtype == torch.float32 and rec.shape == (self.n_agents, E + 4) and rec.stride(1) == 1 and rec.stride(0) == E + 4
        self._rec = rec
        self.out_ctrl = rec[:, :E].view(self.n_agents, cfg.M, cfg.n + 1, 3)
        self.out_goal = rec[:, E:E + 3]
        self.out_status = rec[:, E + 3:E + 4].view(torch.int32)[:, 0]
        p = rec.data_ptr()
        self.c.out_ctrl, self.c.out_goal, self.c.out_status = p, p + 4 * E, p + 4 * (E + 3)
        self.c.out_record_stride = E + 4

    def launch(self, do_goal=True, do_traj=True, stream=None, offset=0, count=0):
        """Asynchronous launch on `stream` (default: torch's current stream); `offset`/`count` restrict it to a
        range of launch slots (count = 0: the whole batch)."""
        self.c.do_goal, self.c.do_traj = int(do_goal), int(do_traj)
        self.c.launch_offset, self.c.launch_count = int(offset), int(count)
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        rc = self.opt.lib.dlscqp_plan_batch_device(self.opt.handle, C.byref(self.c), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_plan_batch_device")

    def results(self):
        self.torch.cuda.synchronize(self.device)
        out = dict(goal=self.out_goal.cpu().numpy(), goal_status=self.out_goal_status.cpu().numpy(),
                   ctrl=self.out_ctrl.cpu().numpy(), cost=self.out_cost.cpu().numpy(),
                   status=self.out_status.cpu().numpy(), slack_cost=self.out_slack.cpu().numpy())
        if self.out_x is not None:
            out["x"] = self.out_x.cpu().numpy()
        if self.out_iters is not None:
            out["iters"] = self.out_iters.cpu().numpy()
        return out


class LscGenerator:
    """DLSC-GC half-spaces of the local agents against their neighbours, generated in HBM from the table of
    previous plans (the all-gathered control points) and sub-goals; feeds DeviceBatch without a host round trip.
    Mirrors TrajPlanner::constructLSC -> generateDLSCGC (traj_planner.cpp:498-515, 603-666) for agent obstacles."""

    def __init__(self, opt, n_total, nbr_offsets, nbr_ids, radius, downwash, local_ids, device, dyn=None):
        import torch
        self.torch = torch
        self.opt = opt
        cfg = opt.cfg
        dev = torch.device("cuda", cfg.device if device is None else device)
        self.device = dev
        nbr_offsets = np.ascontiguousarray(nbr_offsets, dtype=np.int32)
        self.n_local = len(nbr_offsets) - 1
        self.per = cfg.M * (cfg.n + 1) - cfg.phi
        # dynamic obstacles (dict: radius, max_acc, downwash [D]; size_prediction, horizon) follow every agent's neighbours
        self.n_dyn = 0 if dyn is None else len(dyn["radius"])
        pairs = nbr_offsets.astype(np.int64) + self.n_dyn * np.arange(self.n_local + 1)
        self.n_rows = self.per * int(pairs[-1])
        self.row_offsets = self.per * pairs
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
        self.t = dict(nbr_offsets=up(nbr_offsets, np.int32), nbr_ids=up(nbr_ids if len(nbr_ids) else np.zeros(1), np.int32),
                      radius=up(radius, np.float64), downwash=up(downwash, np.float64),
                      local_ids=None if local_ids is None else up(local_ids, np.int32))
        pad = self.n_rows + 64
        self.soa = {k: torch.zeros(pad, dtype=torch.float32, device=dev) for k in ("nx", "ny", "nz", "px", "py", "pz")}
        self.soa["d"] = torch.zeros(pad, dtype=torch.float64, device=dev)
        self.soa["key"] = torch.zeros(pad, dtype=torch.int32, device=dev)
        g = abi.LscGen()
        g.n_local = self.n_local
        g.local_ids = None if self.t["local_ids"] is None else self.t["local_ids"].data_ptr()
        g.radius, g.downwash = self.t["radius"].data_ptr(), self.t["downwash"].data_ptr()
        g.nbr_offsets, g.nbr_ids = self.t["nbr_offsets"].data_ptr(), self.t["nbr_ids"].data_ptr()
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(g, k, self.soa[k].data_ptr())
        if self.n_dyn:
            self.t.update(dyn_radius=up(dyn["radius"], np.float64), dyn_max_acc=up(dyn["max_acc"], np.float64),
                          dyn_downwash=up(dyn["downwash"], np.float64))
            g.n_dyn = self.n_dyn
            g.dyn_size_prediction = int(dyn.get("size_prediction", 1))
            g.dyn_uncertainty_horizon = float(dyn.get("horizon", 1.0))
            g.dyn_radius, g.dyn_max_acc = self.t["dyn_radius"].data_ptr(), self.t["dyn_max_acc"].data_ptr()
            g.dyn_downwash = self.t["dyn_downwash"].data_ptr()
        g.max_neighbours = int(np.diff(nbr_offsets).max()) if self.n_local else 0  # the longest list (set_neighbours updates it)
        self.c = g
        self.n_total = n_total

    def set_neighbours(self, nbr_offsets, nbr_ids):
        """Replace the neighbour lists (broadcastMsgs hands every agent the agents within communication range, which changes
        from step to step: multi_sync_simulator.cpp:482-503).  The total must fit the capacity the generator was made with;
        row_offsets / n_rows follow.  Returns the new row_offsets (host)."""
        nbr_offsets = np.ascontiguousarray(nbr_offsets, dtype=np.int32)
        nbr_ids = np.ascontiguousarray(nbr_ids, dtype=np.int32)
        assert len(nbr_offsets) == self.n_local + 1 and len(nbr_ids) <= len(self.t["nbr_ids"])
        self.c.max_neighbours = int(np.diff(nbr_offsets).max()) if self.n_local else 0
        pairs = nbr_offsets.astype(np.int64) + self.n_dyn * np.arange(self.n_local + 1)
        assert self.per * int(pairs[-1]) + 64 <= len(self.soa["d"])
        self.t["nbr_offsets"].copy_(self.torch.from_numpy(nbr_offsets))
        if len(nbr_ids):
            self.t["nbr_ids"][:len(nbr_ids)].copy_(self.torch.from_numpy(nbr_ids))
        self.n_rows = self.per * int(pairs[-1])
        self.row_offsets = self.per * pairs
        return self.row_offsets

    def waypoint_trap(self, agents, sfcs=None, out_released=None, stream=None, launch=True):
        """TrajPlanner::checkWaypointTrap on the rows of the last generate(): `agents` the uint8 tensor of the local agents'
        records (DeviceBatch.t["agents"]), `sfcs` the corridor tensor [n_local, M, 6] or None, `out_released` an optional
        int32 tensor [n_local].  Asynchronous; a no-op without dynamic obstacles."""
        t = abi.WaypointTrap()
        t.agents = agents.data_ptr()
        t.sfc_boxes = None if sfcs is None else sfcs.data_ptr()
        t.out_released = None if out_released is None else out_released.data_ptr()
        self.trap_c = t
        if not launch:
            return
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        rc = self.opt.lib.dlscqp_waypoint_trap_device(self.opt.handle, C.byref(self.c), C.byref(t), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_waypoint_trap_device")

    def set_tail(self, tail_from, tail_segments):
        """dlscqp_lsc_gen.tail_from / tail_segments: neighbours at list positions >= tail_from contribute only their last
        tail_segments segments (the synthetic swarms: 4 x 27 + 12 rows).  Needs enable_pruning()."""
        self.c.tail_from, self.c.tail_segments = int(tail_from), int(tail_segments)

    def enable_pruning(self, sfcs=None):
        """Presolve inside the generator (dlscqp_lsc_gen.prune): half-spaces implied by the agent's corridor / world box are not
        written; `sfcs` [n_local, M, 6] is the corridor tensor of the local agents for the step (None without corridors).
        Afterwards `row_counts` ([n_local] int32 CUDA tensor) holds the rows per agent: give it to DeviceBatch.use_row_counts."""
        self.row_counts = self.torch.zeros(self.n_local, dtype=self.torch.int32, device=self.device)
        self.c.prune = 1
        self.c.prune_sfc_boxes = None if sfcs is None else sfcs.data_ptr()
        self.c.out_row_counts = self.row_counts.data_ptr()
        self._keep_prune = sfcs
        return self.row_counts

    def generate(self, goals, plans=None, position=None, velocity=None, stream=None, dyn_position=None, dyn_velocity=None,
                 observed_position=None, reset_threshold=0.0, launch=True):
        """goals [n_total,3] f32 and either plans [n_total,M,n+1,3] f32 (previous plans) or position/velocity
        (first replanning step); dyn_position / dyn_velocity [n_dyn,3] f32 when the generator has dynamic obstacles;
        observed_position [n_total,3] f32 + reset_threshold: TrajPlanner::checkObstacleDisturbance: CUDA tensors.  Asynchronous."""
        g = self.c
        if observed_position is not None:
            observed_position = observed_position.contiguous()
            g.observed_position, g.reset_threshold = observed_position.data_ptr(), float(reset_threshold)
            self._keep_obs = observed_position
        else:
            g.observed_position = None
        # raw pointers below: the layout must be the dense one
        goals = goals.contiguous()
        plans = None if plans is None else plans.contiguous()
        position = None if position is None else position.contiguous()
        velocity = None if velocity is None else velocity.contiguous()
        if self.n_dyn:
            dyn_position, dyn_velocity = dyn_position.contiguous(), dyn_velocity.contiguous()
            g.dyn_position, g.dyn_velocity = dyn_position.data_ptr(), dyn_velocity.data_ptr()
            self._keep_dyn = (dyn_position, dyn_velocity)
        g.first_plan = 1 if plans is None else 0
        g.plans = None if plans is None else plans.data_ptr()
        g.position = None if position is None else position.data_ptr()
        g.velocity = None if velocity is None else velocity.data_ptr()
        g.goals = goals.data_ptr()
        self._keep = (goals, plans, position, velocity)
        if not launch:  # (the structure is ready for BatchTrajOptimizer.fused_step)
            return
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        rc = self.opt.lib.dlscqp_generate_lsc_device(self.opt.handle, C.byref(g), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_generate_lsc_device")

    def rows_host(self):
        """Download the rows as AoS records (tests).  With pruning on, only the first row_counts[k] rows of agent k's block are
        meaningful (row_counts_host())."""
        self.torch.cuda.synchronize(self.device)
        R = self.n_rows
        soa = {k: v.cpu().numpy() for k, v in self.soa.items()}
        rows = np.zeros(R, abi.ROW_DTYPE)
        for c, k in enumerate(("nx", "ny", "nz")):
            rows["normal"][:, c] = soa[k][:R]
        for c, k in enumerate(("px", "py", "pz")):
            rows["point"][:, c] = soa[k][:R]
        rows["d"] = soa["d"][:R]
        key = soa["key"][:R].view(np.uint32)
        rows["m"], rows["i"], rows["flags"] = key & 0xFF, (key >> 8) & 0xF, (key >> 12) & 1
        sl = (key >> 16).astype(np.int64)
        rows["slack"] = np.where(sl == 0xFFFF, -1, sl)
        return self.row_offsets, rows


class NeighbourSearch:
    """broadcastMsgs' communication-range filter (multi_sync_simulator.cpp:482-503) on the device: every step the lists of
    a LscGenerator whose lists have a fixed pitch are rebuilt from the agents' positions, in HBM, with nothing sized or
    scanned on the host.  `counts` [n_local] int32 and `overflow` [1] int32 are CUDA tensors; with `row_counts=True` the
    rows per agent that the generator writes WITHOUT its presolve go to `row_counts` (with the presolve the generator
    writes its own)."""

    def __init__(self, opt, gen, pitch, device):
        import torch
        self.torch, self.opt, self.gen = torch, opt, gen
        dev = gen.device if device is None else torch.device("cuda", device)
        self.device = dev
        n = gen.n_local
        self.pitch = int(pitch)
        assert gen.t["nbr_ids"].numel() >= n * self.pitch
        off = gen.t["nbr_offsets"].cpu().numpy()
        assert np.array_equal(off, np.arange(n + 1) * self.pitch), "the generator's lists must have this pitch"
        self.counts = torch.zeros(n, dtype=torch.int32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self.row_counts = torch.zeros(n, dtype=torch.int32, device=dev)
        gen.c.nbr_counts = self.counts.data_ptr()
        c = abi.NeighbourSearch()
        c.n_total, c.n_local = gen.n_total, n
        c.local_ids = None if gen.t["local_ids"] is None else gen.t["local_ids"].data_ptr()
        c.pitch = self.pitch
        c.out_nbr_ids, c.out_nbr_counts = gen.t["nbr_ids"].data_ptr(), self.counts.data_ptr()
        c.out_row_counts = self.row_counts.data_ptr()
        c.n_dyn = gen.n_dyn
        c.out_overflow = self.overflow.data_ptr()
        self.c = c
        # dlscqp_neighbours_device: up to NBR_DIRECT_MAX agents in one launch, every agent a candidate of every other (no cells)
        import os
        self.direct = gen.n_total <= int(os.environ.get("DLSCQP_NBR_DIRECT_MAX", "256"))

    def search(self, position, stream=None, launch=True):
        """position: CUDA float32 tensor [n_total, >= 3] whose rows are `position.stride(0)` floats apart (a dense table or a
        view into the agent records).  Asynchronous."""
        assert position.dtype == self.torch.float32 and position.stride(1) == 1
        self.c.position, self.c.position_stride = position.data_ptr(), int(position.stride(0))
        self._keep = position
        if not launch:
            return
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        rc = self.opt.lib.dlscqp_neighbours_device(self.opt.handle, C.byref(self.c), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_neighbours_device")

    def lists_host(self):
        """(counts, ids [n_local, pitch]) on the host (tests)."""
        self.torch.cuda.synchronize(self.device)
        n = self.gen.n_local
        return self.counts.cpu().numpy(), self.gen.t["nbr_ids"][:n * self.pitch].cpu().numpy().reshape(n, self.pitch)


class KnnSearch:
    """dlscqp_neighbours_knn_device: every step the lists of a LscGenerator with lists of pitch k are rebuilt from the agents'
    positions: the k nearest agents of the agent's own block of `group` consecutive ids (a swarm).  Same interface as
    NeighbourSearch (search, counts, lists_host); `overflow` stays 0 (a list is never cut)."""

    def __init__(self, opt, gen, group, k, device):
        import torch
        self.torch, self.opt, self.gen = torch, opt, gen
        dev = gen.device if device is None else torch.device("cuda", device)
        self.device = dev
        n = gen.n_local
        assert gen.t["local_ids"] is None and gen.n_total == n
        self.pitch = int(k)
        assert np.array_equal(gen.t["nbr_offsets"].cpu().numpy(), np.arange(n + 1) * self.pitch), "the generator's lists must have pitch k"
        self.counts = torch.zeros(n, dtype=torch.int32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        gen.c.nbr_counts = self.counts.data_ptr()
        c = abi.NeighbourKnn()
        c.n_total, c.group, c.k, c.pitch = n, int(group), int(k), self.pitch
        c.out_nbr_ids, c.out_nbr_counts = gen.t["nbr_ids"].data_ptr(), self.counts.data_ptr()
        self.c = c

    def search(self, position, stream=None):
        assert position.dtype == self.torch.float32 and position.stride(1) == 1
        self.c.position, self.c.position_stride = position.data_ptr(), int(position.stride(0))
        self._keep = position
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        rc = self.opt.lib.dlscqp_neighbours_knn_device(self.opt.handle, C.byref(self.c), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_neighbours_knn_device")

    def lists_host(self):
        self.torch.cuda.synchronize(self.device)
        n = self.gen.n_local
        return self.counts.cpu().numpy(), self.gen.t["nbr_ids"][:n * self.pitch].cpu().numpy().reshape(n, self.pitch)


class SfcGenerator:
    """Corridor boxes of every agent, kept in HBM ([N, M, 6] float32) and updated in place once per replanning step;
    the tensor is what DeviceBatch takes as sfc boxes.  Mirrors CollisionConstraints::initializeSFC /
    constructSFCFromInitialTraj (collision_constraints.cpp:435-452, 502-536)."""

    def __init__(self, opt, boxes, resolution, world_min, world_max, radius, max_vel, grid_resolution, device):
        import torch
        self.torch, self.opt = torch, opt
        cfg = opt.cfg
        self.device = torch.device("cuda", cfg.device if device is None else device)
        b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        wmin, wmax = np.ascontiguousarray(world_min, dtype=np.float32), np.ascontiguousarray(world_max, dtype=np.float32)
        self.world = C.c_void_p()
        lib = opt.lib
        lib.dlscqp_world_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.dlscqp_world_destroy.argtypes = [C.c_void_p]
        lib.dlscqp_generate_sfc_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.SfcGen), C.c_void_p]
        opt._check(lib.dlscqp_world_create(opt.handle, b.ctypes.data, len(b), float(resolution), wmin.ctypes.data, wmax.ctypes.data,
                                           C.byref(self.world)), "dlscqp_world_create")
        N = len(radius)
        self.N = N
        # the per-radius-class tables now, from the host's radii: the launches then neither synchronise nor allocate
        lib.dlscqp_world_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        r64 = np.ascontiguousarray(radius, dtype=np.float64)
        if os.environ.get("DLSCQP_WORLD_PREPARE", "1") != "0":  # (0: left to the first launch, as rounds 3-4 did -- tests compare the two)
            opt._check(lib.dlscqp_world_prepare(self.world, r64.ctypes.data, N), "dlscqp_world_prepare")
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)  # noqa: E731
        self.radius, self.max_vel = up(radius, np.float64), up(max_vel, np.float64)
        self.sfcs = torch.zeros((N, cfg.M, 6), dtype=torch.float32, device=self.device)
        self.codes = torch.zeros(N, dtype=torch.int32, device=self.device)
        g = abi.SfcGen()
        g.n_agents, g.grid_resolution = N, float(grid_resolution)
        g.radius, g.max_vel = self.radius.data_ptr(), self.max_vel.data_ptr()
        g.sfcs, g.out_code = self.sfcs.data_ptr(), self.codes.data_ptr()
        self.c = g

    def _launch(self, stream):
        s = self.torch.cuda.current_stream(self.device) if stream is None else stream
        rc = self.opt.lib.dlscqp_generate_sfc_device(self.opt.handle, self.world, C.byref(self.c), C.c_void_p(s.cuda_stream))
        self.opt._check(rc, "dlscqp_generate_sfc_device")

    def initialize(self, position, stream=None):
        """position [N,3] f32 CUDA tensor: initializeSFC for every agent."""
        position = position.contiguous()  # raw pointers below: the layout must be the dense one
        self.c.initialize, self.c.position = 1, position.data_ptr()
        self._keep = (position,)
        self._launch(stream)

    def update(self, initial_traj, goals, waypoints, stream=None, launch=True):
        """initial_traj [N,M,n+1,3], goals, waypoints [N,3] f32 CUDA tensors: one constructSFCFromInitialTraj step."""
        initial_traj, goals, waypoints = initial_traj.contiguous(), goals.contiguous(), waypoints.contiguous()
        self.c.initialize = 0
        self.c.initial_traj, self.c.goals, self.c.waypoints = initial_traj.data_ptr(), goals.data_ptr(), waypoints.data_ptr()
        self._keep = (initial_traj, goals, waypoints)
        if not launch:
            return
        self._launch(stream)

    def close(self):
        if self.world:
            self.opt.lib.dlscqp_world_destroy(self.world)
            self.world = C.c_void_p()
