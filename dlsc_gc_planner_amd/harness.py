"""Closed-loop replanning harness on the device (first slice).

Reproduces the data flow of MultiSyncSimulator::run (/root/reference/src/multi_sync_simulator.cpp:70-141) for
agents in obstacle-free worlds: doStep (state := plan(t + dt)), broadcastMsgs (every agent sees every other
agent's previous plan and sub-goal), then for ALL agents at once: previous-plan shift -> DLSC-GC half-spaces
-> goal LP -> trajectory QP -> fail-safe.  Everything stays in HBM between steps: the half-spaces are generated
by dlscqp_generate_lsc_device, consumed by dlscqp_plan_batch_device, and the new plans become the next step's
plan table.  torch is plumbing only (state bookkeeping: a handful of elementwise ops per step).

Not reproduced (out of scope): the grid planner / PIBT that hands out waypoints in the
reference (N4) -- the waypoint here is the point `waypoint_step` metres ahead on the straight line to the
desired goal -- and static obstacles (N2).  Dynamic obstacles: `dyn` (constant-velocity obstacles, rows from
dlscqp_generate_lsc_device's dynamic branch, soft constraints with slack variables).

ResultCsv writes the reference's log/result_*.csv layout (multi_sync_simulator.cpp:735-850), so that its replay and
analysis scripts can read a run of this harness.
"""
import os
import time

import numpy as np

from . import abi
from .planner import BatchTrajOptimizer


def plan_state_at(plans, n, dt, tau):
    """Position, velocity, acceleration of every agent `tau` seconds into segment 0 of `plans` [N, M, n+1, 3]
    (Trajectory<point3d>::getStateAt, trajectory.cpp:110-199: float32 control-point differences, Bernstein weights).
    Works on numpy arrays and torch tensors alike."""
    from math import comb
    c = plans[:, 0]
    s = tau / dt

    def bern(ctrl, deg):
        w = [comb(deg, i) * s ** i * (1 - s) ** (deg - i) for i in range(deg + 1)]
        out = ctrl[:, 0] * np.float32(w[0])
        for i in range(1, deg + 1):
            out = out + ctrl[:, i] * np.float32(w[i])
        return out

    v = (c[:, 1:] - c[:, :-1]) * np.float32(n / dt)
    a = (v[:, 1:] - v[:, :-1]) * np.float32((n - 1) / dt)
    return bern(c, n), bern(v, n - 1), bern(a, n - 2)


class ResultCsv:
    """The reference's result file: one header, then per replanning step one line per recorded time (future_time =
    0, save_time_step, ... < dt), each line `id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time` for every agent followed by
    `obs_id,t,px,py,pz,size` for every dynamic obstacle.  Numbers as C++ iostreams print them (%g, 6 digits)."""

    def __init__(self, path, n_agents, n_obs=0, save_time_step=0.1):
        self.fh = open(path, "w")
        self.n_agents, self.n_obs, self.save_time_step = n_agents, n_obs, save_time_step
        cols = ["id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time"] * n_agents + ["obs_id,t,px,py,pz,size"] * n_obs
        self.fh.write(",".join(cols) + "\n")

    def write_step(self, t, plans, n, dt, planning_time, obs_state=None):
        """plans: numpy [N, M, n+1, 3] just planned at simulation time t; obs_state(tau) -> (pos [D,3], size [D])."""
        tau = 0.0
        while tau < dt - 1e-12:
            p, v, a = plan_state_at(plans, n, dt, tau)
            cells = []
            for q in range(self.n_agents):
                vals = [q, t + tau, *p[q], *v[q], *a[q], planning_time]
                cells.append(",".join("%g" % float(x) for x in vals))
            if self.n_obs:
                op, osz = obs_state(tau)
                for o in range(self.n_obs):
                    cells.append(",".join("%g" % float(x) for x in (o, t + tau, *op[o], osz[o])))
            self.fh.write(",".join(cells) + "\n")
            tau += self.save_time_step

    def close(self):
        self.fh.close()


class SpinObstacles:
    """The reference's spinning obstacles (SpinObstacle::getSpinObstacle, obstacle.hpp:113-149): obstacle o circles the axis
    (axis_position, axis_ori) through its start point with tangential speed `speed`; state(t) returns what the planner is
    told at time t -- position and velocity, float32 [D,3] (it then extrapolates with constant velocity)."""

    def __init__(self, obstacles):
        """obstacles: the mission file's "obstacles" entries of type "spin"."""
        o = [ob for ob in obstacles if ob["type"] == "spin"]
        self.axis = np.array([ob["axis_position"] for ob in o], np.float64)
        n = np.array([ob["axis_ori"] for ob in o], np.float64)
        self.n = n / np.linalg.norm(n, axis=1, keepdims=True)
        self.a = np.array([ob["start"] for ob in o], np.float64) - self.axis
        r = self.a - (self.a * self.n).sum(1, keepdims=True) * self.n
        self.w = np.array([ob["speed"] for ob in o], np.float64) / np.linalg.norm(r, axis=1)
        self.radius = np.array([ob["size"] for ob in o], np.float64)
        self.max_acc = np.array([ob["max_acc"] for ob in o], np.float64)
        self.downwash = np.array([ob["downwash"] if ob["downwash"] != 0 else 1.0 for ob in o], np.float64)

    @staticmethod
    def _cross(a, b):  # (np.cross spends 20 us per call on axis bookkeeping: this runs every replanning step)
        return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)

    @classmethod
    def _rotate(cls, v, n, theta):  # Rodrigues; the reference conjugates with the unit quaternion of the same rotation
        c, s = np.cos(theta)[:, None], np.sin(theta)[:, None]
        return v * c + cls._cross(n, v) * s + n * (n * v).sum(1, keepdims=True) * (1 - c)

    def state(self, t):
        p = self._rotate(self.a, self.n, self.w * t)
        v = self.w[:, None] * self._rotate(p, self.n, np.full(len(self.w), np.pi / 2))
        return (self.axis + p).astype(np.float32), v.astype(np.float32)

    def dyn(self, size_prediction=1, horizon=1.0, duration=-1.0):
        """The `dyn` argument of ClosedLoop for these obstacles."""
        p, v = self.state(0.0)
        return dict(position=p, velocity=v, radius=self.radius, max_acc=self.max_acc, downwash=self.downwash,
                    size_prediction=size_prediction, horizon=horizon, model=self, duration=duration)


class ClosedLoop:
    def __init__(self, cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, nominal_velocity=1.0,
                 waypoint_step=0.5, device=0, dyn=None, result_csv=None, world=None, neighbours=None, nbr_pitch=96,
                 corridor_margin=None, prune=False, order_hint=False, knn=None, tail=None, heavy_ranks=0):
        """dyn: None or dict(position [D,3], velocity [D,3], radius, max_acc, downwash [D], size_prediction, horizon) --
        obstacles that keep their velocity (obstacle.hpp: straight-line dynamic obstacles).  result_csv: path of a
        result file in the reference's layout.  world: None or dict(boxes [K,6] centre+size, resolution, grid_resolution)
        -- static obstacles: corridor boxes from the device generator (row N2) and waypoints from GridWaypoints.
        neighbours="device": the communication-range neighbour lists (broadcastMsgs, multi_sync_simulator.cpp:482-503) are
        rebuilt in HBM every step by dlscqp_neighbours_device, at most nbr_pitch per agent -- what a crowd of thousands needs
        (all-to-all lists grow with N^2).  corridor_margin (obstacle-free worlds with world_use_octomap set): one corridor
        box per segment, the bounding box of the segment's control points of the initial trajectory grown by that many metres
        (a stand-in for the boxes the octomap corridor generator would grow in a crowded maze).  prune: the generator's
        presolve.  order_hint: every launch in the order of the two previous steps' iteration counts, found by the plan kernel
        itself (dlscqp_device_batch.hint_iters).  neighbours="knn", knn=(group, k): the synthetic swarms' rule -- the k nearest
        agents of the agent's own block of `group` consecutive ids (dlscqp_neighbours_knn_device); tail=(tail_from, tail_segments):
        dlscqp_lsc_gen.tail_from (needs prune): BASELINE.json configs[1] is knn=(64, 5), tail=(4, 2): 120 rows per agent."""
        import torch
        self.torch = torch
        self.cfg = cfg
        self.opt = BatchTrajOptimizer(cfg)
        dev = torch.device("cuda", device)
        self.dev = dev
        N = len(start)
        self.N = N
        M, n = cfg.M, cfg.n
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        self.desired_goal = f32(goal)
        self.cur_goal = f32(start)  # agent_manager.cpp:9
        self.plans = torch.zeros((N, M, n + 1, 3), dtype=torch.float32, device=dev)
        self.init_buf = torch.zeros_like(self.plans)  # initial trajectory: the previous plan shifted by one segment
        self.graph = None
        self.keep_inputs = False  # tests: keep a copy of the agent records each launch read (step_inputs)
        self.radius, self.downwash = float(radius), float(downwash)
        self.waypoint_step = waypoint_step
        self.seq = 0
        # every agent is every other agent's obstacle (communication range -1 in testall_DLSCGC_empty.launch)
        ids = np.arange(N)
        self.nbr = None
        if neighbours is None and world is not None and world.get("pibt", False) == "reference" and cfg.communication_range > 0 and N > 1:
            # the reference's missions: broadcastMsgs' communication-range lists, rebuilt in HBM every step (a list per agent of
            # everybody else at most: the pitch never overflows) instead of on the host
            neighbours, nbr_pitch = "device", N - 1
        if neighbours == "device":
            assert cfg.communication_range > 0
            nbr_pitch = min(int(nbr_pitch), max(N - 1, 1))
            nbr_ids, nbr_offsets = np.zeros(N * nbr_pitch, np.int32), np.arange(N + 1) * nbr_pitch
        elif neighbours == "knn":
            nbr_pitch = int(knn[1])
            nbr_ids, nbr_offsets = np.zeros(N * nbr_pitch, np.int32), np.arange(N + 1) * nbr_pitch
        else:
            nbr_ids = np.concatenate([np.delete(ids, a) for a in range(N)]) if N > 1 else np.zeros(0, np.int32)
            nbr_offsets = np.arange(N + 1) * (N - 1)
        self.dyn = dyn
        self.D = 0 if dyn is None else len(dyn["radius"])
        self.dyn_model = None if dyn is None else dyn.get("model")
        if self.D:
            # position and velocity of the obstacles: two views of one buffer, refreshed by ONE transfer per step from pinned
            # memory that does not make the host wait for the step's kernels (step())
            self._dyn_dev = torch.stack([f32(dyn["position"]), f32(dyn["velocity"])])
            self.dyn_pos, self.dyn_vel = self._dyn_dev[0], self._dyn_dev[1]
            self._dyn_pin = torch.empty((2, self.D, 3), dtype=torch.float32).pin_memory()
            # the same states on the host, for the host-side waypoint planner (float32, the same arithmetic as on the device)
            self._dyn_host = [np.array(dyn["position"], np.float32).reshape(self.D, 3), np.array(dyn["velocity"], np.float32).reshape(self.D, 3)]
        self.gen = self.opt.make_lsc_generator(N, nbr_offsets, nbr_ids, np.full(N, radius), np.full(N, downwash), device=device,
                                               dyn=dyn)
        if neighbours == "device":
            self.nbr = self.opt.make_neighbour_search(self.gen, nbr_pitch)
        elif neighbours == "knn":
            self.nbr = self.opt.make_knn_search(self.gen, knn[0], knn[1])
            assert prune, "lists of the k nearest: the generator counts the rows it writes (prune=True)"
        self.world = world
        self.sfc = None
        self.box_margin = None
        self.order_hint = bool(order_hint)
        # heavy_ranks = R > 0 (with order_hint): every block's R most expensive agents -- launch slots [0, R n_blocks) of the class-mask
        # order -- are planned by a launch of their own on a second stream, beside the launch of the others: small enough
        # (<= 512 workgroups) to get the build with helper wavefronts, whose helpers price the rows in LDS.  An experiment, off by
        # default: measured on the flown crowd it LOSES (0.31 -> 0.35-0.37 ms per step) -- the previous step's iteration counts do
        # not say who the slowest agent of this step will be (tests/tools/crowd_tail.py), so the launch of the others is as long
        # as the whole launch was.
        self.heavy_ranks = int(heavy_ranks) if order_hint else 0
        self._heavy_stream = None
        self.pack_in_advance = False
        self.comm_neighbours = False
        self.slack_threshold = float(world.get("slack_threshold", 0.1)) if world is not None else 0.1  # param.slack_threshold
        if world is not None:
            wmin, wmax = np.array(cfg.world_min[:], np.float32), np.array(cfg.world_max[:], np.float32)
            self.sfc = self.opt.make_sfc_generator(world["boxes"], world["resolution"], wmin, wmax, np.full(N, radius),
                                                   np.full(N, max_vel), world["grid_resolution"], device=device)
            if world.get("pibt", False) == "reference":
                # the reference's own planner (row N4, csrc/grid_mapf.cpp) and its communication-range neighbour lists
                self.grid = MapfWaypoints(cfg, world["boxes"], world["resolution"], world["grid_resolution"], np.asarray(start, np.float32),
                                          np.asarray(goal, np.float32), radius, downwash, world.get("uncertainty_horizon", 1.0))
                self.comm_neighbours = cfg.communication_range > 0 and self.nbr is None  # host-side lists only without the device search
            elif world.get("pibt", False):
                self.grid = PibtWaypoints(world["boxes"], world["grid_resolution"], wmin, wmax, np.asarray(start, np.float64),
                                          np.asarray(goal, np.float64), radius)
            else:
                self.grid = GridWaypoints(world["boxes"], world["grid_resolution"], wmin, wmax, np.asarray(goal, np.float64), radius)
        self.csv = ResultCsv(result_csv, N, self.D) if result_csv else None
        self.sim_time = 0.0
        # agent records: static fields once, the five float3 fields are refreshed every step
        agents = np.zeros(N, abi.AGENT_DTYPE)
        agents["id"] = ids
        agents["n_obs"] = N - 1 + self.D
        agents["n_slack_obs"] = self.D
        agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = max_vel, max_acc, radius, nominal_velocity
        ro = self.gen.row_offsets
        sfc0 = None if self.sfc is None else np.zeros((N, M, 6), np.float32)
        self.batch = self.opt.make_device_batch(agents, ro, np.zeros(0, abi.ROW_DTYPE), sfc0, self.D, want_iters=True, device=device)
        if self.sfc is not None:
            self.batch.c.sfc_boxes = self.sfc.sfcs.data_ptr()  # the generator's tensor is the QP's SFC input
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(self.batch.c, k, self.gen.soa[k].data_ptr())
        self.ro32 = torch.from_numpy(ro.astype(np.int32)).to(dev)
        self.batch.c.row_offsets = self.ro32.data_ptr()
        # sized once to the generator's capacity and never shrunk: the launch's kernel choice, LDS layout and flag array
        # follow it, and a captured graph freezes them
        self.batch.c.max_rows_per_agent = int(np.diff(ro).max()) if N else 0
        if N > 1 and not self.comm_neighbours:
            # lists of a fixed pitch (everybody, or the device search's lists): the kernel computes where an agent's rows start
            self.batch.use_row_pitch(int(ro[1] - ro[0]))
        if corridor_margin is not None:
            assert world is None and cfg.world_use_octomap, "synthetic corridor boxes stand in for the octomap generator"
            self.box_margin = float(corridor_margin)
            self.box_t = torch.zeros((N, M, 6), dtype=torch.float32, device=dev)
            self.batch.sfc_t = self.box_t  # keep alive
            self.batch.c.sfc_boxes = self.box_t.data_ptr()
            self.box_lo = f32(np.array(cfg.world_min[:], np.float32))
            self.box_hi = f32(np.array(cfg.world_max[:], np.float32))
        self.row_counts = None
        if prune:
            self.row_counts = self.gen.enable_pruning(self.sfc.sfcs if self.sfc is not None else (self.box_t if self.box_margin is not None else None))
            if tail is not None:
                # the rows an agent can have at most: what picks the kernel instantiation and sizes its caches
                self.gen.set_tail(*tail)
                per = cfg.M * (cfg.n + 1)
                full = min(int(tail[0]), nbr_pitch)
                self.batch.c.max_rows_per_agent = full * (per - cfg.phi) + (nbr_pitch - full) * int(tail[1]) * (cfg.n + 1) + self.D * (per - cfg.phi)
        elif self.nbr is not None:
            self.row_counts = self.nbr.row_counts
        if self.row_counts is not None:
            self.batch.use_row_counts(self.row_counts)
        self.agent_f32 = self.batch.t["agents"].view(torch.float32).view(N, abi.AGENT_DTYPE.itemsize // 4)
        # position, velocity, acceleration and next_waypoint live in the agent records (views); dlscqp_advance_device
        # rewrites them after every launch
        a = self.agent_f32
        self.pos, self.vel, self.acc, self.wp = a[:, 0:3], a[:, 3:6], a[:, 6:9], a[:, 12:15]
        a[:, 0:3], a[:, 9:12], a[:, 12:15] = self.cur_goal, self.cur_goal, self.cur_goal
        if world is None:
            # waypoint: straight ahead towards the desired goal (stands in for the grid planner, see module doc); the
            # first one here, the following ones by dlscqp_advance_device
            to_goal = self.desired_goal - self.pos
            dist = to_goal.norm(dim=1, keepdim=True)
            self.wp.copy_(torch.where(dist <= waypoint_step, self.desired_goal,
                                      self.pos + to_goal / dist.clamp_min(1e-9) * waypoint_step))
        # (the boxes of a world without obstacles are written by the same kernel, for the step that follows: box_in_advance)
        self.box_in_advance = self.box_margin is not None and os.environ.get("DLSCQP_BOX_IN_ADVANCE", "1") != "0"
        bkw = dict(boxes=self.box_t, box_margin=self.box_margin, box_min=cfg.world_min, box_max=cfg.world_max) if self.box_in_advance else {}
        if world is not None and isinstance(self.grid, MapfWaypoints) and os.environ.get("DLSCQP_HOST_PACK", "1") != "0":
            # what the host-side waypoint planner reads (positions, sub-goals, previous plans, collision alerts) is written into
            # pinned host memory by the step's last kernel: no pack, no device-to-host copy, no unpack (step(): one stream
            # synchronisation, which the planner needs anyway)
            E = M * (cfg.n + 1) * 3
            self._adv_pack = torch.zeros(N * (6 + E + self.D), dtype=torch.float32).pin_memory()
            self._wp_pin = torch.empty((N, 3), dtype=torch.float32).pin_memory()
            bkw.update(host_pack=self._adv_pack, alert_obs=self.D, alert_threshold=self.slack_threshold)
            self.pack_in_advance = True
        self.advance = self.opt.make_advance(self.batch, self.plans, self.cur_goal, self.init_buf,
                                             desired_goal=self.desired_goal if world is None else None,
                                             waypoint_step=waypoint_step if world is None else 0.0,
                                             hold_on_lp_failure=neighbours == "knn", **bkw)  # the synthetic swarms' waypoint rule
        # ... and the plan kernel does that bookkeeping itself, in its epilogue: one launch less per step (DLSCQP_FUSED_ADVANCE=0: the
        # launch of its own, as rounds 2-5 had it)
        self.fused_advance = os.environ.get("DLSCQP_FUSED_ADVANCE", "1") != "0" and self.heavy_ranks == 0
        # ONE launch per steady-state step where the library has a fused build for the mission (dlscqp_fused_step_device: at most 256
        # agents, the reference's launch-file shapes); the first unsupported answer switches back to the chain of launches for good.
        # OFF unless DLSCQP_FUSED_STEP=1: measured in round 6 (tests/tools/fused_time.py, one MI355X) the fused step is bit-equal and
        # NOT faster -- empty10 0.0448 against 0.0440 ms p50, empty60 0.0746 against 0.0683, maze10 0.1486 against 0.1503, the same as
        # HIP graphs: a step's stages are chains of dependent trips to L2 / HBM whichever kernel they are in, and the gaps between
        # launches that are queued ahead are not what a step waits for
        self.fused_step = os.environ.get("DLSCQP_FUSED_STEP", "0") == "1" and self.fused_advance and N <= 256
        self.fused_steps_done = 0
        if self.fused_advance:
            self.batch.set_advance_after(self.advance)
        if world is not None and world.get("prune", True) and not prune:
            # presolve in the half-space generator: rows implied by this step's corridor boxes are never written
            self.row_counts = self.gen.enable_pruning(self.sfc.sfcs)
            self.batch.use_row_counts(self.row_counts)
        self.records_in = torch.zeros_like(self.batch.t["agents"])
        self.released = torch.zeros(N, dtype=torch.int32, device=dev)  # dynamic obstacles released by the waypoint trap, per agent
        self.history = []

    def close(self):
        if self.sfc is not None:
            self.sfc.close()
        if self.csv:
            self.csv.close()
        if hasattr(getattr(self, "grid", None), "close"):
            self.grid.close()  # (the host planner's handle; with DLSCQP_MAPF_PROFILE it prints where its time went)
        self.opt.close()

    def _waypoints_host(self):
        """Waypoints from the host-side grid planner stand-ins (worlds with static obstacles)."""
        torch, cfg = self.torch, self.cfg
        if isinstance(self.grid, MapfWaypoints):
            # MultiSyncSimulator::decentralizedMAPP: positions, sub-goals, previous plans in; next_waypoint updated in place
            # one device-to-host transfer for everything the planner reads (positions, sub-goals, waypoints, previous plans,
            # obstacle states, slack costs): packed on the device, copied into a pinned buffer
            N, D, E = self.N, self.D, cfg.M * (cfg.n + 1) * 3
            if self.pack_in_advance and self.seq > 0:
                # written by dlscqp_advance_device at the end of the previous step (dlscqp_advance.out_host_pack); the waypoints
                # are where the planner left them (nothing on the device changes them), the obstacle states came from the host
                torch.cuda.current_stream(self.dev).synchronize()
                h = self._adv_pack.numpy()
                o = np.cumsum([0, 3 * N, 3 * N, N * E, N * D])
                gdyn = None
                if D:
                    gdyn = dict(position=self._dyn_host[0], velocity=self._dyn_host[1], radius=self.dyn["radius"], max_acc=self.dyn["max_acc"])
                alert = (h[o[3]:o[4]].reshape(N, D) > 0.5).astype(np.uint8) if D else None
                pos = h[o[0]:o[1]].reshape(N, 3)
                self.mapf_rc = self.grid.step(pos, h[o[1]:o[2]].reshape(N, 3), self._wp_pin.numpy(), h[o[2]:o[3]].reshape(N, cfg.M, cfg.n + 1, 3),
                                              dyn=gdyn, alert=alert)
                if self.comm_neighbours:
                    self._set_comm_neighbours(pos)
                return self._wp_pin
            parts = [self.pos.reshape(-1), self.cur_goal.reshape(-1), self.wp.reshape(-1), self.plans.reshape(-1)]
            if D:
                # TrajOptimizer::solve's collision alert (traj_optimizer.cpp:85-106): obstacles whose slack cost in the last
                # QP exceeded param.slack_threshold (compared on the device, in double); the planner treats them as obstacles
                # of interest
                parts += [self.dyn_pos.reshape(-1), self.dyn_vel.reshape(-1),
                          (self.batch.out_slack[:, :D] > self.slack_threshold).to(torch.float32).reshape(-1)]
            packed = torch.cat(parts)
            if getattr(self, "_host_pack", None) is None or self._host_pack.numel() != packed.numel():
                self._host_pack = torch.empty(packed.numel(), dtype=torch.float32).pin_memory()
            self._host_pack.copy_(packed, non_blocking=True)
            torch.cuda.current_stream(self.dev).synchronize()
            h = self._host_pack.numpy()
            o = np.cumsum([0, 3 * N, 3 * N, 3 * N, N * E] + ([3 * D, 3 * D, N * D] if D else []))
            pos, cur_goal = h[o[0]:o[1]].reshape(N, 3).copy(), h[o[1]:o[2]].reshape(N, 3).copy()
            if getattr(self, "_wp_pin", None) is None:
                self._wp_pin = torch.empty((N, 3), dtype=torch.float32).pin_memory()
            wp = self._wp_pin.numpy()  # the planner updates the waypoints in place, in the buffer they are uploaded from
            wp[:] = h[o[2]:o[3]].reshape(N, 3)
            plans = h[o[3]:o[4]].reshape(N, cfg.M, cfg.n + 1, 3).copy()
            gdyn = None
            if D:
                gdyn = dict(position=h[o[4]:o[5]].reshape(D, 3).copy(), velocity=h[o[5]:o[6]].reshape(D, 3).copy(), radius=self.dyn["radius"],
                            max_acc=self.dyn["max_acc"])
            alert = (h[o[6]:o[7]].reshape(N, D) > 0.5).astype(np.uint8) if D and self.seq > 0 else None
            self.mapf_rc = self.grid.step(pos, cur_goal, wp, plans if self.seq > 0 else None, dyn=gdyn, alert=alert)
            if self.comm_neighbours:
                self._set_comm_neighbours(pos)
            return self._wp_pin
        if isinstance(self.grid, PibtWaypoints):
            anchors = torch.cat([self.pos[:, None, :], self.plans[:, :, 0, :], self.plans[:, -1:, -1, :]], dim=1) \
                if self.seq > 0 else self.pos[:, None, :]
            gdyn = None
            if self.D:
                gdyn = dict(position=self.dyn_pos.cpu().numpy(), velocity=self.dyn_vel.cpu().numpy(), radius=self.dyn["radius"],
                            max_acc=self.dyn["max_acc"], horizon=cfg.M * cfg.dt, uncertainty_horizon=self.dyn.get("horizon", 1.0))
            return torch.from_numpy(self.grid.step(self.cur_goal.cpu().numpy(), anchors.cpu().numpy(),
                                                   float(cfg.communication_range), dyn=gdyn))
        return torch.from_numpy(self.grid.next_waypoints(self.pos.cpu().numpy()))

    def _set_comm_neighbours(self, pos):
        """broadcastMsgs (multi_sync_simulator.cpp:482-503): agent qi is told about the agents within communication range
        (L-infinity) of it, by id; the half-space generator and the QP launch get this step's lists and row counts."""
        N, r = self.N, float(self.cfg.communication_range)
        d = np.abs(pos[:, None, :] - pos[None, :, :]).max(axis=2)
        keep = ~(d > np.float32(r))
        np.fill_diagonal(keep, False)
        if getattr(self, "_comm_keep", None) is not None and np.array_equal(keep, self._comm_keep):
            return  # the same lists as in the previous step: the generator and the launch already have them
        self._comm_keep = keep
        offs = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
        ids = np.concatenate([np.nonzero(keep[a])[0] for a in range(N)]) if N > 1 else np.zeros(0, np.int32)
        ro = self.gen.set_neighbours(offs, ids)
        self.ro32.copy_(self.torch.from_numpy(ro.astype(np.int32)))
        # max_rows_per_agent stays at the generator's capacity (set once in the constructor): see there
        self.n_neighbours = keep.sum(axis=1)

    # ---- per-stage timing in the reference's vocabulary (timer.hpp, sp_const.hpp:81-122; the columns of its summary file,
    # multi_sync_simulator.cpp:852-900): HIP events around the launches of a stage, host time for the waypoint planner
    STAGES = ("sfc_generation_time", "lsc_generation_time", "traj_optimization_time", "state_update_time")

    def enable_stage_timers(self, on=True):
        """Record a pair of events around every stage of the following steps (not with a captured graph); stage_times()
        returns what they measured.  traj_optimization_time = goal LP + QP (one fused launch here; the reference times
        goal_planning_time and traj_optimization_time apart); lsc_generation_time includes the neighbour search and the
        waypoint trap; mapf_time is host time of the reference's waypoint planner incl. its transfers."""
        self._stage_ev = [] if on else None
        self._mapf_ms = []

    def _stage(self, name):
        if getattr(self, "_stage_ev", None) is None:
            return
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(self.torch.cuda.current_stream(self.dev))
        self._stage_ev.append((name, ev))

    def stage_times(self):
        """dict stage -> array of ms per step (steps since enable_stage_timers).
        With the bookkeeping between two plans done by the plan kernel's own epilogue (fused_advance, the default:
        dlscqp_device_batch.advance_after) it is part of traj_optimization_time and state_update_time reads ~0 -- to compare the columns
        with the reference's summary file (multi_sync_simulator.cpp:863-871) stage by stage, run with DLSCQP_FUSED_ADVANCE=0."""
        self.torch.cuda.synchronize(self.dev)
        out = {k: [] for k in self.STAGES}
        ev = self._stage_ev
        cur = {}
        for i in range(len(ev) - 1):
            name, e0 = ev[i]
            if name == "end":
                continue
            cur[name] = cur.get(name, 0.0) + e0.elapsed_time(ev[i + 1][1])
            if ev[i + 1][0] == "end":
                for k in self.STAGES:
                    out[k].append(cur.get(k, 0.0))
                cur = {}
        res = {k: np.array(v) for k, v in out.items()}
        if self._mapf_ms:
            res["mapf_time"] = np.array(self._mapf_ms)
        return res

    def _body(self, first):
        """The device work of one step: half-spaces, corridors, goal LP + QP, then dlscqp_advance_device (fail-safe,
        sub-goal, state := plan(dt), the next step's agent records, waypoint in empty worlds, initial trajectory).  Three or
        four launches that read and write fixed buffers and never synchronise: the steady-state form (first = False) is
        what enable_graph() captures."""
        cfg = self.cfg
        dkw = dict(dyn_position=self.dyn_pos, dyn_velocity=self.dyn_vel) if self.D else {}
        if (self.fused_step and not first and not self.keep_inputs and getattr(self, "_stage_ev", None) is None
                and (self.box_margin is None or self.box_in_advance) and (self.nbr is None or hasattr(self.nbr, "direct"))):
            # the same structures, filled the same way, handed over in one call: neighbour lists, corridor boxes, half-spaces, waypoint
            # trap, goal LP + QP and the state update are stages of ONE kernel (dlscqp_fused_step_device)
            if self.sfc is not None:
                self.sfc.update(self.init_buf, self.cur_goal, self.wp, launch=False)
            if self.nbr is not None:
                self.nbr.search(self.pos, launch=False)
            self.gen.generate(self.cur_goal, plans=self.plans, launch=False, **dkw)
            trap = bool(self.D and self.world is not None)
            if trap:
                self.gen.waypoint_trap(self.batch.t["agents"], None if self.sfc is None else self.sfc.sfcs, self.released, launch=False)
            if self.opt.fused_step(self.batch, self.gen, nbr=self.nbr, sfc=self.sfc, trap=trap):
                self.fused_steps_done += 1
                return
            self.fused_step = False  # no fused build for this mission: the chain below, from now on
        # the corridor boxes do not depend on the half-spaces (constructLSC and constructSFC of TrajPlanner::planImpl are
        # independent), so they come first here: the half-space generator can then drop the rows that the step's boxes imply
        if first:
            pos = self.pos.contiguous()
            # "previous plan" of the first step: hovering at the start (zero velocity), so that its shift is itself
            self.plans.copy_(pos[:, None, None, :].expand(-1, cfg.M, cfg.n + 1, -1))
            self.init_buf.copy_(self.plans)
            if self.sfc is not None:  # TrajPlanner::constructSFC (traj_planner.cpp:517-520): initializeSFC once, then one shift per step
                self.sfc.initialize(pos)
        self._stage("sfc_generation_time")
        if self.sfc is not None:
            self.sfc.update(self.init_buf, self.cur_goal, self.wp)
        if self.box_margin is not None and (first or not self.box_in_advance):
            # one box per segment around the initial trajectory (the shifted previous plan), inside the world box (after the
            # first step dlscqp_advance_device has written them at the end of the step before: five small launches less)
            torch = self.torch
            lo = torch.maximum(self.init_buf.amin(dim=2) - self.box_margin, self.box_lo)
            hi = torch.minimum(self.init_buf.amax(dim=2) + self.box_margin, self.box_hi)
            self.box_t[:, :, 0:3].copy_(lo)
            self.box_t[:, :, 3:6].copy_(hi)
        self._stage("lsc_generation_time")
        if self.nbr is not None:
            self.nbr.search(self.pos)
        if first:
            self.gen.generate(self.cur_goal, position=self.pos.contiguous(), velocity=self.vel.contiguous(), **dkw)
        else:
            self.gen.generate(self.cur_goal, plans=self.plans, **dkw)
        if self.D and self.world is not None:
            # TrajPlanner::checkWaypointTrap (traj_planner.cpp:708-735): a waypoint cut off by neighbours / corridor / range
            # and within reach of a dynamic obstacle -> that obstacle's rows are dropped for this step
            self.gen.waypoint_trap(self.batch.t["agents"], None if self.sfc is None else self.sfc.sfcs, self.released)
        if self.keep_inputs:
            self.records_in.copy_(self.batch.t["agents"])
        self._stage("traj_optimization_time")
        if self.order_hint:
            # launch order found by the kernel itself from the class masks the previous step's launch left behind
            # (dlscqp_device_batch.hint_mask_*): three tables in rotation
            self.batch.rotate_hint_masks()
        if self.heavy_ranks > 0:
            torch = self.torch
            nb = (self.N + 63) // 64
            ordered = bool(self.batch.c.hint_mask_in)
            total = 64 * nb if ordered else self.N
            h = min(self.heavy_ranks * nb, total)
            cur = torch.cuda.current_stream(self.dev)
            if self._heavy_stream is None:
                self._heavy_stream = torch.cuda.Stream(self.dev)
            side = self._heavy_stream
            side.wait_stream(cur)
            self.batch.launch(True, True, stream=side, offset=0, count=h)
            if total > h:
                self.batch.launch(True, True, stream=cur, offset=h, count=total - h)
            cur.wait_stream(side)
        else:
            self.batch.launch(True, True)
        self._stage("state_update_time")
        # fail-safe: keep the shifted previous plan (traj_planner.cpp:776); sub-goal from the LP (traj_planner.cpp:494);
        # doStep of the next iteration: state = desired_traj.getStateAt(dt) = start of segment 1, float32 differences
        if not self.fused_advance:
            self.advance()
        self._stage("end")

    def enable_graph(self):
        """Capture the steady-state step in a HIP graph (torch.cuda.CUDAGraph): a ten-agent step is three or four small
        launches and launch-bound; replaying it as one graph removes the per-launch host cost.  Call after at
        least one step; result files and the host-side waypoint planners keep working (they run outside the graph)."""
        torch = self.torch
        assert self.seq >= 1, "the first step differs (constant-velocity prediction, corridor initialisation)"
        # (the launch-order hint rotates three mask tables from step to step: three graphs, one per phase of the rotation, replayed in turn)
        # (round 4 refused the device neighbour search through cells here: its captured step faulted on the second replay.  The
        # cause was the search's memset node -- the cell counts were not cleared again on replay, the slots ran past their cells'
        # runs and the scatter wrote out of bounds; the search now clears its counts itself, in its last kernel)
        torch.cuda.synchronize(self.dev)
        # (the graphs hold raw pointers: spill records in a buffer of the batch's own, not in the context's, which a later, larger
        # launch on the same context would free and allocate again under them)
        if getattr(self.batch, "_workspace", None) is None:
            self.batch.use_workspace()
        state = [self.batch.t["agents"], self.cur_goal, self.plans, self.init_buf]
        if self.sfc is not None:
            state.append(self.sfc.sfcs)  # the corridor boxes are shifted in place every step
        if self.box_margin is not None:
            state.append(self.box_t)
        if self.pack_in_advance:
            state.append(self._adv_pack)  # (pinned host memory: the warm-up's last kernel writes it)
        keep = [t.clone() for t in state]
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # warm-up on a side stream, as torch asks before a capture
            self._body(False)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        graphs = []
        for _ in range(3 if self.order_hint else 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._body(False)
            graphs.append(g)
        torch.cuda.synchronize(self.dev)
        for t, k in zip(state, keep):
            t.copy_(k)  # the warm-up advanced the state: put it back
        self.graph = graphs
        self._graph_k = 0

    def step(self):
        """One replanning step of every agent; returns (status [N] tensor, iterations tensor)."""
        torch, cfg = self.torch, self.cfg
        n, dt = cfg.n, cfg.dt
        first = self.seq < 1  # planner_seq < 2 in the reference counts from 1
        t_wall = time.perf_counter() if self.csv else 0.0
        if self.world is not None:
            t_mapf = time.perf_counter()
            self.wp.copy_(self._waypoints_host(), non_blocking=True)
            if getattr(self, "_stage_ev", None) is not None:
                self._mapf_ms.append((time.perf_counter() - t_mapf) * 1e3)
        if self.graph is not None and not first:
            self.graph[self._graph_k % len(self.graph)].replay()
            self._graph_k += 1
        else:
            self._body(first)
        if self.csv:
            torch.cuda.synchronize(self.dev)
            dyn_now = (self.dyn_pos.cpu().numpy(), self.dyn_vel.cpu().numpy()) if self.D else None
            if self.dyn_model is not None:
                obs_state = lambda tau: (self.dyn_model.state(self.sim_time + tau)[0], self.dyn["radius"])  # noqa: E731
            else:
                obs_state = lambda tau: (dyn_now[0] + dyn_now[1] * np.float32(tau), self.dyn["radius"])  # noqa: E731
            self.csv.write_step(self.sim_time, self.plans.cpu().numpy(), n, dt, time.perf_counter() - t_wall,
                                None if dyn_now is None else obs_state)
        self.sim_time += dt
        if self.D and 0 < self.dyn.get("duration", -1.0) < self.sim_time:
            self.dyn_model = None
            self.dyn_pos.fill_(1000.0)
            self.dyn_vel.zero_()
            self._dyn_host = [np.full((self.D, 3), 1000.0, np.float32), np.zeros((self.D, 3), np.float32)]
        elif self.dyn_model is not None:
            # (computed while the step's kernels run: nothing here waits for them)
            p, v = self.dyn_model.state(self.sim_time)
            pin = self._dyn_pin.numpy()
            pin[0], pin[1] = p, v
            self._dyn_host = [pin[0].copy(), pin[1].copy()]
            self._dyn_dev.copy_(self._dyn_pin, non_blocking=True)
        elif self.D:
            self.dyn_pos.add_(self.dyn_vel * np.float32(dt))
            self._dyn_host[0] = self._dyn_host[0] + self._dyn_host[1] * np.float32(dt)
        self.seq += 1
        return self.batch.out_status, self.batch.out_iters

    def step_inputs(self):
        """Host copies of what the last step's launch read -- (agents, row_offsets, rows, sfc or None, max_slack_obs) in the
        layout of BatchTrajOptimizer.plan -- for checking a closed-loop step against another solver (tests)."""
        self.torch.cuda.synchronize(self.dev)
        assert self.keep_inputs, "set keep_inputs before stepping: the records are rewritten after every launch"
        agents = self.records_in.cpu().numpy().view(abi.AGENT_DTYPE).reshape(-1).copy()
        ro, rows = self.gen.rows_host()
        if self.row_counts is not None:  # presolve or lists of fixed pitch: only the first row_counts[k] rows of every block
            cnt = self.row_counts.cpu().numpy()
            rows = np.concatenate([rows[ro[k]:ro[k] + cnt[k]] for k in range(len(cnt))]) if len(cnt) else rows[:0]
            ro = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        sfc = None if self.sfc is None else self.sfc.sfcs.cpu().numpy().copy()
        if self.box_margin is not None:
            sfc = self.box_t.cpu().numpy().copy()
        return agents, ro, rows, sfc, self.D

    def min_obstacle_separation(self):
        """Smallest agent-to-dynamic-obstacle distance in the pair's downwash metric divided by the sum of the radii (the
        reference's safety_ratio_obs, multi_sync_simulator.cpp:686-712)."""
        torch = self.torch
        r_o = torch.tensor(self.dyn["radius"], dtype=torch.float32, device=self.dev)
        dw_o = torch.tensor(self.dyn["downwash"], dtype=torch.float32, device=self.dev)
        dw = (r_o * dw_o + self.radius * self.downwash) / (self.radius + r_o)
        d = self.pos[:, None, :] - self.dyn_pos[None, :, :]
        d = torch.stack([d[..., 0], d[..., 1], d[..., 2] / dw[None, :]], dim=-1)
        return float((d.norm(dim=2) / (self.radius + r_o)[None, :]).min())

    def min_separation(self):
        """Smallest pairwise distance in the downwash metric divided by the collision distance (>= 1: no collision),
        the reference's safety_ratio_agent (multi_sync_simulator.cpp:652-690)."""
        torch = self.torch
        d = self.pos[:, None, :] - self.pos[None, :, :]
        d = d * torch.tensor([1.0, 1.0, 1.0 / self.downwash], device=self.dev)
        dist = d.norm(dim=2) + torch.eye(self.N, device=self.dev) * 1e9
        return float(dist.min() / (2 * self.radius))


class ShardedLoop:
    """BASELINE.json configs[2]: an obstacle-free mission flown in closed loop with its agents sharded over the ranks of a
    torch.distributed group by contiguous id range (sharding.shard_range).  Every rank holds the table of everybody's
    previous plans and sub-goals; per replanning step it generates the half-spaces of ITS agents from that table, plans them
    (goal LP + QP), updates their states, and ONE all-gather (sharding.PlanExchange: RCCL over xGMI) of plans, sub-goals and
    verdicts rebuilds the table -- the reference's broadcastMsgs (multi_sync_simulator.cpp:468-514) ahead of its plan loop
    (:516-524).  In these missions every agent neighbours every other (communication range -1): every agent is a halo
    agent and the exchange is on the critical path by necessity.  With world size 1 (one-rank group) the same code runs
    with a one-rank all-gather."""

    def __init__(self, cfg, start, goal, dist, rank, world, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0,
                 nominal_velocity=1.0, waypoint_step=0.5, device=0):
        import torch
        from . import sharding
        self.torch, self.dist, self.rank, self.world, self.cfg = torch, dist, rank, world, cfg
        self.opt = BatchTrajOptimizer(cfg)
        dev = torch.device("cuda", device)
        self.dev = dev
        N = len(start)
        self.N = N
        M, n = cfg.M, cfg.n
        lo, hi = sharding.shard_range(N, rank, world)
        self.lo, self.hi = lo, hi
        nl = hi - lo
        self.pad = max(sharding.shard_sizes(N, world))
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        start, goal = np.asarray(start, np.float32), np.asarray(goal, np.float32)
        # the table every rank holds
        self.plans_all = f32(np.broadcast_to(start[:, None, None, :], (N, M, n + 1, 3)).copy())  # hovering at the start
        self.goals_all = f32(start)
        self.pos_all, self.vel_all = f32(start), torch.zeros((N, 3), dtype=torch.float32, device=dev)
        ids = np.arange(N)
        own = np.arange(lo, hi)
        nbr_ids = np.concatenate([np.delete(ids, a) for a in own]) if N > 1 else np.zeros(0, np.int32)
        self.gen = self.opt.make_lsc_generator(N, np.arange(nl + 1) * (N - 1), nbr_ids, np.full(N, radius), np.full(N, downwash),
                                               local_ids=own.astype(np.int32), device=device)
        agents = np.zeros(nl, abi.AGENT_DTYPE)
        agents["id"] = own
        agents["n_obs"] = N - 1
        agents["max_vel"], agents["max_acc"], agents["radius"], agents["nominal_velocity"] = max_vel, max_acc, radius, nominal_velocity
        for k in ("position", "current_goal_point", "next_waypoint"):
            agents[k] = start[lo:hi]
        ro = self.gen.row_offsets
        self.batch = self.opt.make_device_batch(agents, ro, np.zeros(0, abi.ROW_DTYPE), None, 0, want_iters=True, device=device)
        for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
            setattr(self.batch.c, k, self.gen.soa[k].data_ptr())
        self.ro32 = torch.from_numpy(ro.astype(np.int32)).to(dev)
        self.batch.c.row_offsets = self.ro32.data_ptr()
        self.batch.c.max_rows_per_agent = int(np.diff(ro).max()) if nl else 0
        self.plans = self.plans_all[lo:hi].clone()
        self.init_buf = self.plans.clone()
        self.cur_goal = f32(start[lo:hi])
        self.desired_goal = f32(goal[lo:hi])
        a = self.batch.t["agents"].view(torch.float32).view(nl, abi.AGENT_DTYPE.itemsize // 4)
        self.pos, self.wp = a[:, 0:3], a[:, 12:15]
        to_goal = self.desired_goal - self.pos
        dist_ = to_goal.norm(dim=1, keepdim=True)
        self.wp.copy_(torch.where(dist_ <= waypoint_step, self.desired_goal, self.pos + to_goal / dist_.clamp_min(1e-9) * waypoint_step))
        self.advance = self.opt.make_advance(self.batch, self.plans, self.cur_goal, self.init_buf, desired_goal=self.desired_goal,
                                             waypoint_step=waypoint_step)
        E = M * (n + 1) * 3
        self.E = E
        self.xchg = sharding.make_exchange(self.opt, dist, rank, world, self.pad, dev)  # RCCL from the library itself; gloo: torch.distributed
        self.send_plans = torch.zeros((self.pad, E), dtype=torch.float32, device=dev)
        self.send_goal = torch.zeros((self.pad, 3), dtype=torch.float32, device=dev)
        self.send_status = torch.zeros((self.pad, 1), dtype=torch.int32, device=dev)
        rows = np.concatenate([r * self.pad + np.arange(sharding.shard_range(N, r, world)[1] - sharding.shard_range(N, r, world)[0])
                               for r in range(world)])
        self.valid_rows = torch.from_numpy(rows.astype(np.int64)).to(dev)
        self.seq = 0

    def step(self):
        """One replanning step of this rank's agents and the exchange that follows it; asynchronous."""
        torch = self.torch
        nl = self.hi - self.lo
        if self.seq == 0:
            self.gen.generate(self.goals_all, position=self.pos_all, velocity=self.vel_all)
        else:
            self.gen.generate(self.goals_all, plans=self.plans_all)
        self.batch.launch(True, True)
        self.advance()
        self.send_plans[:nl].copy_(self.plans.view(nl, -1))
        self.send_goal[:nl].copy_(self.cur_goal)
        self.send_status[:nl].copy_(self.batch.out_status.view(nl, 1))
        self.xchg.post(self.send_plans, self.send_goal, self.send_status)
        table = self.xchg.table()  # every agent of the next step reads other ranks' plans: wait here
        if self.world > 1 or self.pad * self.world != self.N:
            table = table.index_select(0, self.valid_rows)
        self.plans_all.view(self.N, -1).copy_(table[:, :self.E])
        self.goals_all.copy_(table[:, self.E:self.E + 3])
        self.seq += 1

    def close(self):
        self.opt.close()


class MapfWaypoints:
    """The reference's decentralised waypoint planner (MultiSyncSimulator::decentralizedMAPP: communication groups, grid
    map, PIBT, waypoint update rules; multi_sync_simulator.cpp:308-466, grid_based_planner.cpp, src/mapf/pibt.cpp) through
    the C ABI (dlscqp_mapf_*, csrc/grid_mapf.cpp).  Host only: numpy arrays in, next_waypoint updated in place."""

    def __init__(self, cfg, boxes, world_resolution, grid_resolution, start, goal, radius, downwash, uncertainty_horizon=1.0):
        import ctypes as C
        self.C = C
        self.lib = abi.load_library()
        mc = abi.MapfConfig()
        mc.world_dimension, mc.M, mc.world_use_octomap = cfg.world_dimension, cfg.M, int(cfg.world_use_octomap)
        mc.dt, mc.grid_resolution, mc.world_resolution, mc.world_z_2d = cfg.dt, grid_resolution, world_resolution, cfg.world_z_2d
        mc.communication_range, mc.agent_radius, mc.agent_downwash = cfg.communication_range, radius, downwash
        mc.obs_uncertainty_horizon = uncertainty_horizon
        for k in range(3):
            mc.world_min[k], mc.world_max[k] = cfg.world_min[k], cfg.world_max[k]
        b = np.ascontiguousarray(boxes if boxes is not None else np.zeros((0, 6)), dtype=np.float64).reshape(-1, 6)
        self.h = C.c_void_p()
        rc = self.lib.dlscqp_mapf_create(C.byref(mc), b.ctypes.data if len(b) else None, len(b), C.byref(self.h))
        if rc != abi.SUCCESS:
            raise ValueError("dlscqp_mapf_create failed (%d)" % rc)
        self.n1 = cfg.n + 1
        self.start = np.ascontiguousarray(start, dtype=np.float32)
        self.goal = np.ascontiguousarray(goal, dtype=np.float32)
        self.N = len(self.start)
        self.desired = np.zeros((self.N, 3), np.float32)
        self.group = np.zeros(self.N, np.int32)

    def close(self):
        if self.h:
            self.lib.dlscqp_mapf_destroy(self.h)
            self.h = self.C.c_void_p()

    def step(self, position, current_goal, next_waypoint, plans=None, dyn=None, alert=None):
        """position, current_goal [N,3] float32; next_waypoint [N,3] float32, updated IN PLACE; plans [N,M,n+1,3] float32
        (the plans of the previous step) or None before the first plan; dyn: None or dict(position, velocity [D,3] float32,
        radius, max_acc [D]).  Returns 0, or 1 if the planner failed for some group."""
        a = abi.MapfStepArgs()
        keep = [np.ascontiguousarray(position, dtype=np.float32), np.ascontiguousarray(current_goal, dtype=np.float32)]
        assert next_waypoint.dtype == np.float32 and next_waypoint.flags.c_contiguous
        a.n_agents, a.n1 = self.N, self.n1
        a.position, a.current_goal = keep[0].ctypes.data, keep[1].ctypes.data
        a.start_point, a.desired_goal = self.start.ctypes.data, self.goal.ctypes.data
        a.next_waypoint = next_waypoint.ctypes.data
        if plans is not None:
            keep.append(np.ascontiguousarray(plans, dtype=np.float32))
            a.plans = keep[-1].ctypes.data
        if dyn is not None and len(dyn["radius"]):
            dp, dv = np.ascontiguousarray(dyn["position"], np.float32), np.ascontiguousarray(dyn["velocity"], np.float32)
            dr, da = np.ascontiguousarray(dyn["radius"], np.float64), np.ascontiguousarray(dyn["max_acc"], np.float64)
            keep += [dp, dv, dr, da]
            a.n_obs = len(dr)
            a.obs_position, a.obs_velocity, a.obs_radius, a.obs_max_acc = dp.ctypes.data, dv.ctypes.data, dr.ctypes.data, da.ctypes.data
            if alert is not None:
                keep.append(np.ascontiguousarray(alert, np.uint8))
                a.alert = keep[-1].ctypes.data
        a.out_desired, a.out_group = self.desired.ctypes.data, self.group.ctypes.data
        rc = self.lib.dlscqp_mapf_step(self.h, self.C.byref(a))
        if rc < 0:
            raise ValueError("dlscqp_mapf_step failed (%d)" % rc)
        return rc


class GridWaypoints:
    """Host-side stand-in for the reference's grid-based waypoint planner (grid_based_planner.cpp; SURVEY row N4 stays on
    the host): a 2-D grid of `grid_resolution` cells over the world, a cell is blocked if an obstacle box comes within the
    agent radius of its centre; per agent a breadth-first distance field from its goal cell.  next_waypoints() returns, for
    every agent, the centre of the neighbouring cell (4-connected) with the smallest distance to the goal -- the goal
    itself once the agent is in the goal cell.  No conflict resolution between agents (that is PIBT's job in the reference)."""

    def __init__(self, boxes, grid_resolution, world_min, world_max, goals, radius):
        from collections import deque
        g = float(grid_resolution)
        self.g = g
        self.i0 = np.ceil(np.asarray(world_min[:2], np.float64) / g).astype(int)
        self.n = np.floor(np.asarray(world_max[:2], np.float64) / g).astype(int) - self.i0 + 1
        b = np.asarray(boxes, np.float64).reshape(-1, 6)
        cx = (self.i0[0] + np.arange(self.n[0])) * g
        cy = (self.i0[1] + np.arange(self.n[1])) * g
        X, Y = np.meshgrid(cx, cy, indexing="ij")
        free = np.ones(X.shape, bool)
        for c in b:
            free &= ~((np.abs(X - c[0]) < 0.5 * c[3] + radius) & (np.abs(Y - c[1]) < 0.5 * c[4] + radius))
        self.free = free
        self.goals = np.asarray(goals, np.float64)
        self.dist = []
        for gl in self.goals:
            d = np.full(free.shape, -1, int)
            c0 = self._cell(gl)
            q = deque([c0])
            d[c0] = 0
            while q:
                i, j = q.popleft()
                for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    a, bb = i + di, j + dj
                    if 0 <= a < self.n[0] and 0 <= bb < self.n[1] and free[a, bb] and d[a, bb] < 0:
                        d[a, bb] = d[i, j] + 1
                        q.append((a, bb))
            self.dist.append(d)

    def _cell(self, p):
        c = np.rint(np.asarray(p[:2], np.float64) / self.g).astype(int) - self.i0
        return (int(np.clip(c[0], 0, self.n[0] - 1)), int(np.clip(c[1], 0, self.n[1] - 1)))

    def next_waypoints(self, pos):
        out = np.array(pos, np.float32)
        for a, p in enumerate(pos):
            i, j = self._cell(p)
            d = self.dist[a]
            if d[i, j] == 0:
                out[a, :2] = self.goals[a, :2]
                continue
            best, bc = None, (i, j)
            for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                x, y = i + di, j + dj
                if 0 <= x < self.n[0] and 0 <= y < self.n[1] and d[x, y] >= 0 and (best is None or d[x, y] < best):
                    best, bc = d[x, y], (x, y)
            out[a, 0] = (self.i0[0] + bc[0]) * self.g
            out[a, 1] = (self.i0[1] + bc[1]) * self.g
        return out


class PibtWaypoints(GridWaypoints):
    """GridWaypoints plus the conflict resolution of the reference's planner: waypoints are grid nodes, no two agents hold
    the same node or swap nodes, and who yields is decided by priority inheritance with backtracking (PIBT, Okumura et al.;
    reference: src/mapf/pibt.cpp:119-190, one step of it per replanning step, multi_sync_simulator.cpp:308-466).  An agent's
    waypoint advances only when its sub-goal has reached the waypoint it holds; the others keep theirs.  Priorities: steps
    since the agent last stood on its goal node, ties by initial path length, then id (pibt.cpp:14-21 without the obstacle
    term).  Host-side and simplified (SURVEY row N4 stays on the host): not a restatement of the reference's update rules."""

    def __init__(self, boxes, grid_resolution, world_min, world_max, start, goals, radius):
        super().__init__(boxes, grid_resolution, world_min, world_max, goals, radius)
        self.radius = float(radius)
        self.node = [self._cell(p) for p in np.asarray(start, np.float64)]
        self.goal_node = [self._cell(g) for g in self.goals]
        self.elapsed = [0] * len(self.node)
        self.init_d = [int(self.dist[a][self.node[a]]) for a in range(len(self.node))]

    def _pos(self, node):
        return ((self.i0[0] + node[0]) * self.g, (self.i0[1] + node[1]) * self.g)

    def step(self, sub_goals, anchors=None, comm_range=-1.0, reach=0.02, dyn=None):
        """sub_goals [N,3]: current_goal_point of every agent; anchors [N,K,3]: current position and the segment start / end
        points of the current plan -- with a communication range the next waypoint must stay within range/2 (L-infinity) of
        all of them (multi_sync_simulator.cpp:384-407; the QP has the matching rows, traj_optimizer.cpp:490-513).
        dyn: None or dict(position, velocity [D,3], radius, max_acc [D], horizon, uncertainty_horizon): nodes inside the
        region a dynamic obstacle can reach within the planning horizon are "warning" nodes (Obstacle::isCollided,
        obstacle.hpp:26-37; grid_based_planner.cpp:140-154): an agent holding one may move on at once, and free nodes are
        preferred to warning nodes before anything else.  Returns the waypoints [N,3] float32."""
        N = len(self.node)
        sub_goals = np.asarray(sub_goals, np.float64)
        if dyn is not None and len(dyn["radius"]):
            h = float(dyn["horizon"])
            ts = np.arange(0.0, h + 1e-12, min(0.1 * h, 0.1))
            grow = 0.5 * np.asarray(dyn["max_acc"], np.float64)[:, None] * np.minimum(ts, dyn["uncertainty_horizon"])[None, :] ** 2
            reach_r = self.radius + np.asarray(dyn["radius"], np.float64)[:, None] + grow  # [D,T]
            track = np.asarray(dyn["position"], np.float64)[:, None, :] + np.asarray(dyn["velocity"], np.float64)[:, None, :] * ts[None, :, None]

            def warn(a, u):
                pt = np.array([*self._pos(u), sub_goals[a, 2]])
                return bool((np.linalg.norm(track - pt, axis=2) < reach_r).any())
        else:
            def warn(a, u):
                return False
        # an agent resting on its goal node does not start a move, but can be pushed aside by priority inheritance
        threatened = [warn(a, self.node[a]) for a in range(N)]
        movable = [threatened[a] or
                   np.hypot(*(sub_goals[a, :2] - (self.goals[a, :2] if self.node[a] == self.goal_node[a] else self._pos(self.node[a])))) <= reach
                   for a in range(N)]
        if comm_range > 0 and anchors is not None:
            anchors = np.asarray(anchors, np.float64)
            for a in range(N):
                if movable[a]:  # any neighbouring node is at most one cell further away than the node held now
                    far = np.abs(anchors[a, :, :2] - np.array(self._pos(self.node[a]))).max()
                    movable[a] = far + self.g <= 0.5 * comm_range - 1e-5
        threatened = [t and m for t, m in zip(threatened, movable)]
        now = {self.node[a]: a for a in range(N)}
        nxt, v_next = {}, [None] * N
        for a in range(N):  # agents that are still on their way to their waypoint keep it
            if not movable[a]:
                v_next[a] = self.node[a]
                nxt[self.node[a]] = a

        def choose(a):
            i, j = self.node[a]
            cand = [(i + 1, j), (i - 1, j), (i, j + 1), (i, j - 1), (i, j)]
            best, key = None, None
            for u in cand:
                if not (0 <= u[0] < self.n[0] and 0 <= u[1] < self.n[1]) or self.dist[a][u] < 0:
                    continue
                if u in nxt:
                    continue  # vertex conflict
                b = now.get(u)
                if b is not None and b != a and v_next[b] == self.node[a]:
                    continue  # swap conflict
                gx, gy = self._pos(self.goal_node[a])
                ux, uy = self._pos(u)
                k = (1 if warn(a, u) else 0, int(self.dist[a][u]), 1 if (b is not None and b != a) else 0, np.hypot(ux - gx, uy - gy))
                if key is None or k < key:
                    best, key = u, k
            return best

        def pibt(a):
            v = choose(a)
            while v is not None:
                nxt[v] = a
                v_next[a] = v
                b = now.get(v)
                if b is not None and b != a and v_next[b] is None:
                    if not pibt(b):  # the agent in the way could not move: try another node
                        del nxt[v]
                        v_next[a] = None
                        v = choose_excluding(a, v)
                        continue
                return True
            v_next[a] = self.node[a]
            nxt[self.node[a]] = a
            return False

        banned = {}

        def choose_excluding(a, v):
            banned.setdefault(a, set()).add(v)
            saved = dict(nxt)
            for u in banned[a]:
                nxt.setdefault(u, -1)
            r = choose(a)
            for u in list(nxt):
                if nxt[u] == -1:
                    del nxt[u]
            nxt.update({k2: v2 for k2, v2 in saved.items() if k2 not in nxt})
            return r

        order = sorted(range(N), key=lambda a: (-self.elapsed[a], self.init_d[a], a))
        for a in order:
            if v_next[a] is None and (self.node[a] != self.goal_node[a] or threatened[a]):
                pibt(a)
        for a in range(N):  # resting agents that nobody pushed
            if v_next[a] is None:
                v_next[a] = self.node[a]
        out = np.array(sub_goals, np.float32)
        for a in range(N):
            self.node[a] = v_next[a]
            self.elapsed[a] = 0 if self.node[a] == self.goal_node[a] else self.elapsed[a] + 1
            if self.node[a] == self.goal_node[a]:
                out[a, :2] = self.goals[a, :2]
            else:
                out[a, :2] = self._pos(self.node[a])
        return out
