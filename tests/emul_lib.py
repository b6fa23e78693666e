"""ctypes access to tests/emul/libdlscqp_emul.so -- TEST INFRASTRUCTURE ONLY.

The lane-serial host build of the kernel source (tests/emul/emul.cpp).  It lets the CPU-only
test tier check the device algorithm against the oracle.  Never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from dlsc_gc_planner_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")])
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_emul.so"))
    vp = C.c_void_p
    L.emul_plan_batch.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                  vp, vp, vp, vp, vp, vp, vp, vp]
    L.emul_plan_batch.restype = C.c_int
    L.emul_table_info.argtypes = [C.POINTER(abi.Config), vp, vp, vp, vp, C.c_int, C.c_int]
    L.emul_table_info.restype = C.c_int
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def conflict_doubles(cfg, max_slack_obs=0):
    f = table_info(cfg, max_slack_obs, 0)["f"]
    return 4 + 2 * (cfg.world_dimension * f + cfg.M * max_slack_obs + 1)


def write_lp(cfg, agent, rows, path, sfc=None, max_slack_obs=0, goal=None):
    """csrc/model_rows.hpp: the agent's model as a CPLEX-LP file (the product's dlscqp_write_lp without a device context)"""
    agent = np.ascontiguousarray(agent, dtype=abi.AGENT_DTYPE).reshape(1)
    rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
    sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
    goal = None if goal is None else np.ascontiguousarray(goal, dtype=np.float32)
    rc = lib().emul_write_lp(C.byref(cfg), _ptr(agent), _ptr(rows) if len(rows) else None, len(rows), _ptr(sfc), max_slack_obs, _ptr(goal), str(path).encode())
    if rc != 0:
        raise ValueError("emul_write_lp failed (%d)" % rc)


def conflict_decode(cfg, agent, rows, record, sfc=None, max_slack_obs=0, goal=None):
    """csrc/model_rows.hpp: one out_conflict record -> (members, n_con, gap), like BatchTrajOptimizer.conflict_decode"""
    agent = np.ascontiguousarray(agent, dtype=abi.AGENT_DTYPE).reshape(1)
    rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
    sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
    goal = None if goal is None else np.ascontiguousarray(goal, dtype=np.float32)
    record = np.ascontiguousarray(record, dtype=np.float64)
    cap = max(int(record[0]), 1)
    out = (abi.ConflictMember * cap)()
    n, ncon, gap = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
    L = lib()
    L.emul_conflict_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(abi.ConflictMember), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    rc = L.emul_conflict_decode(C.byref(cfg), _ptr(agent), _ptr(rows) if len(rows) else None, len(rows), _ptr(sfc), max_slack_obs, _ptr(goal),
                                _ptr(record), out, cap, C.byref(n), C.byref(ncon), C.byref(gap))
    if rc != 0:
        raise ValueError("emul_conflict_decode failed (%d)" % rc)
    mem = [dict(ref_index=int(m.ref_index), kind=abi.MODEL_KINDS[m.kind], var=[int(m.var[t]) for t in range(m.n)],
                coef=[float(m.coef[t]) for t in range(m.n)], lo=float(m.lo), hi=float(m.hi), lam=float(m.lam)) for m in out[:n.value]]
    return mem, int(ncon.value), float(gap.value)


def plan_batch(cfg, agents, row_offsets, rows, sfc=None, max_slack_obs=0, do_goal=True, do_traj=True, want_conflict=False):
    L = lib()
    agents = np.ascontiguousarray(agents, dtype=abi.AGENT_DTYPE)
    rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
    row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int64)
    sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
    N = len(agents)
    M, n, dim = cfg.M, cfg.n, cfg.world_dimension
    nx = dim * M * (n + 1) + M * max_slack_obs
    out = dict(goal=np.zeros((N, 3), np.float32), goal_status=np.zeros(N, np.int32),
               ctrl=np.zeros((N, M, n + 1, 3), np.float32), x=np.zeros((N, nx)), cost=np.zeros(N),
               slack_cost=np.zeros((N, max(max_slack_obs, 1))), status=np.full(N, -1, np.int32),
               iters=np.zeros((N, 4), np.int32))
    if want_conflict:  # dlscqp_device_batch.out_conflict: certificates of the INFEASIBLE verdicts
        out["conflict"] = np.zeros((N, conflict_doubles(cfg, max_slack_obs)))
        L.emul_set_conflict.argtypes = [C.c_void_p, C.c_int]
        L.emul_set_conflict(_ptr(out["conflict"]), out["conflict"].shape[1])
    try:
        rc = _emul_plan(L, cfg, N, agents, row_offsets, rows, sfc, max_slack_obs, do_goal, do_traj, out)
    finally:
        if want_conflict:
            L.emul_set_conflict(None, 0)
    if rc != 0:
        raise ValueError("emul_plan_batch: invalid configuration")
    return out


def _emul_plan(L, cfg, N, agents, row_offsets, rows, sfc, max_slack_obs, do_goal, do_traj, out):
    rc = L.emul_plan_batch(C.byref(cfg), N, _ptr(agents), _ptr(row_offsets), _ptr(rows), _ptr(sfc),
                           max_slack_obs, int(do_goal), int(do_traj), _ptr(out["goal"]),
                           _ptr(out["goal_status"]), _ptr(out["ctrl"]), _ptr(out["x"]), _ptr(out["cost"]),
                           _ptr(out["slack_cost"]) if max_slack_obs else None, _ptr(out["status"]), _ptr(out["iters"]))
    return rc


def neighbours(cfg, position, pitch, local_ids=None, n_dyn=0):
    """The CPU tier's stand-in for dlscqp_neighbours_device: (counts [n_local], ids [n_local, pitch], row_counts, overflow)."""
    L = lib()
    L.emul_neighbours.argtypes = [C.POINTER(abi.Config), C.POINTER(abi.NeighbourSearch)]
    L.emul_neighbours.restype = C.c_int
    pos = np.ascontiguousarray(position, dtype=np.float32)
    loc = None if local_ids is None else np.ascontiguousarray(local_ids, dtype=np.int32)
    n_local = len(pos) if loc is None else len(loc)
    ids = np.zeros((n_local, pitch), np.int32)
    counts, rows, over = np.zeros(n_local, np.int32), np.zeros(n_local, np.int32), np.zeros(1, np.int32)
    ns = abi.NeighbourSearch()
    ns.n_total, ns.n_local = len(pos), n_local
    ns.local_ids = None if loc is None else loc.ctypes.data
    ns.position, ns.position_stride, ns.pitch = pos.ctypes.data, pos.shape[1], pitch
    ns.out_nbr_ids, ns.out_nbr_counts, ns.out_row_counts = ids.ctypes.data, counts.ctypes.data, rows.ctypes.data
    ns.n_dyn, ns.out_overflow = n_dyn, over.ctypes.data
    assert L.emul_neighbours(C.byref(cfg), C.byref(ns)) == 0
    return counts, ids, rows, int(over[0])


def table_info(cfg, max_slack_obs=0, kmax=0):
    L = lib()
    v = [C.c_int() for _ in range(4)]
    rc = L.emul_table_info(C.byref(cfg), *[C.byref(t) for t in v], max_slack_obs, kmax)
    if rc != 0:
        raise ValueError("invalid configuration")
    return dict(f=v[0].value, n_slots=v[1].value, nb=v[2].value, lds_bytes=v[3].value)


def generate_lsc(cfg, plans, goals, radius, downwash, nbr_offsets, nbr_ids, local_ids=None, position=None, velocity=None,
                 dyn=None, observed=None, reset_threshold=0.0, trap=None, prune=None, nbr_counts=None):
    """prune: None or dict(sfc=[n_local][M][6] or None): the generator's presolve; on return prune["counts"] holds the rows per
    agent and the returned rows are the kept ones, concatenated (row_offsets accordingly)."""
    """Row N1 through the host build of lsc_core.hpp.  Returns (row_offsets, rows as ROW_DTYPE).
    dyn: None or dict(position [D,3], velocity [D,3], radius [D], max_acc [D], downwash [D], size_prediction, horizon)."""
    L = lib()
    L.emul_generate_lsc.argtypes = [C.POINTER(abi.Config), C.POINTER(abi.LscGen)]
    L.emul_generate_lsc.restype = C.c_int
    L.emul_waypoint_trap.argtypes = [C.POINTER(abi.Config), C.POINTER(abi.LscGen), C.POINTER(abi.WaypointTrap)]
    L.emul_waypoint_trap.restype = C.c_int

    def call(g):
        rc = L.emul_generate_lsc(C.byref(cfg), C.byref(g))
        if rc == 0 and trap is not None:  # TrajPlanner::checkWaypointTrap on the rows just generated
            t = abi.WaypointTrap()
            ag = np.ascontiguousarray(trap["agents"], dtype=abi.AGENT_DTYPE)
            sfc = None if trap.get("sfc") is None else np.ascontiguousarray(trap["sfc"], dtype=np.float32)
            rel = np.zeros(len(ag), np.int32)
            t.agents, t.sfc_boxes, t.out_released = ag.ctypes.data, None if sfc is None else sfc.ctypes.data, rel.ctypes.data
            rc = L.emul_waypoint_trap(C.byref(cfg), C.byref(g), C.byref(t))
            trap["released"] = rel
        return rc

    return _run_lsc_gen(cfg, call, plans, goals, radius, downwash, nbr_offsets, nbr_ids, local_ids, position, velocity, dyn,
                        observed, reset_threshold, prune, nbr_counts)


def _run_lsc_gen(cfg, call, plans, goals, radius, downwash, nbr_offsets, nbr_ids, local_ids, position, velocity, dyn=None,
                 observed=None, reset_threshold=0.0, prune=None, nbr_counts=None):
    first = plans is None
    keep = []

    def arr(a, dt):
        if a is None:
            return None, None
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b, b.ctypes.data

    g = abi.LscGen()
    nbr_offsets = np.ascontiguousarray(nbr_offsets, dtype=np.int32)
    nbr_ids_a = np.ascontiguousarray(nbr_ids, dtype=np.int32)
    n_local = len(nbr_offsets) - 1
    g.n_local, g.first_plan = n_local, 1 if first else 0
    _, g.local_ids = arr(local_ids, np.int32)
    _, g.plans = arr(plans, np.float32)
    _, g.position = arr(position, np.float32)
    _, g.velocity = arr(velocity, np.float32)
    _, g.goals = arr(goals, np.float32)
    if observed is not None:
        _, g.observed_position = arr(observed, np.float32)
        g.reset_threshold = float(reset_threshold)
    _, g.radius = arr(radius, np.float64)
    _, g.downwash = arr(downwash, np.float64)
    g.nbr_offsets, g.nbr_ids = nbr_offsets.ctypes.data, nbr_ids_a.ctypes.data
    if nbr_counts is not None:  # lists of fixed pitch (dlscqp_neighbours_device)
        _, g.nbr_counts = arr(nbr_counts, np.int32)
    per = cfg.M * (cfg.n + 1) - cfg.phi
    n_dyn = 0 if dyn is None else len(dyn["radius"])
    if n_dyn:
        g.n_dyn = n_dyn
        g.dyn_size_prediction = int(dyn.get("size_prediction", 1))
        g.dyn_uncertainty_horizon = float(dyn.get("horizon", 1.0))
        _, g.dyn_position = arr(dyn["position"], np.float32)
        _, g.dyn_velocity = arr(dyn["velocity"], np.float32)
        _, g.dyn_radius = arr(dyn["radius"], np.float64)
        _, g.dyn_max_acc = arr(dyn["max_acc"], np.float64)
        _, g.dyn_downwash = arr(dyn["downwash"], np.float64)
    pairs = nbr_offsets.astype(np.int64) + n_dyn * np.arange(n_local + 1)
    R = per * int(pairs[-1])
    out = {k: np.zeros(R + 1, np.float32) for k in ("nx", "ny", "nz", "px", "py", "pz")}
    out["d"] = np.zeros(R + 1)
    out["key"] = np.zeros(R + 1, np.uint32)
    for k, v in out.items():
        setattr(g, k, v.ctypes.data)
    counts = None
    if prune is not None:
        counts = np.zeros(n_local, np.int32)
        g.prune, g.out_row_counts = 1, counts.ctypes.data
        if prune.get("sfc") is not None:
            _, g.prune_sfc_boxes = arr(prune["sfc"], np.float32)
    rc = call(g)
    assert rc == 0
    ro, rows = per * pairs, soa_to_rows(out, R)
    if counts is not None:
        prune["counts"] = counts
        rows = np.concatenate([rows[ro[k]:ro[k] + counts[k]] for k in range(n_local)]) if n_local else rows[:0]
        ro = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return ro, rows


def soa_to_rows(soa, R):
    rows = np.zeros(R, abi.ROW_DTYPE)
    rows["normal"][:, 0], rows["normal"][:, 1], rows["normal"][:, 2] = soa["nx"][:R], soa["ny"][:R], soa["nz"][:R]
    rows["point"][:, 0], rows["point"][:, 1], rows["point"][:, 2] = soa["px"][:R], soa["py"][:R], soa["pz"][:R]
    rows["d"] = soa["d"][:R]
    key = soa["key"][:R]
    rows["m"] = key & 0xFF
    rows["i"] = (key >> 8) & 0xF
    rows["flags"] = (key >> 12) & 1
    sl = (key >> 16).astype(np.int64)
    rows["slack"] = np.where(sl == 0xFFFF, -1, sl)
    return rows


def generate_sfc(cfg, world, radius, max_vel, grid_resolution, sfcs=None, position=None, initial_traj=None, goals=None, waypoints=None):
    """Row N2 through the host build of sfc_core.hpp + sfc_world.cpp.  world: dict(boxes, resolution, world_min, world_max).
    position given: initializeSFC; else constructSFCFromInitialTraj on `sfcs` [N][M][6].  Returns (sfcs, codes)."""
    L = lib()
    L.emul_generate_sfc.argtypes = [C.POINTER(abi.Config), C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(abi.SfcGen)]
    L.emul_generate_sfc.restype = C.c_int
    N = len(radius)
    g = abi.SfcGen()
    keep = []

    def arr(a, dt):
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b.ctypes.data

    g.n_agents, g.initialize, g.grid_resolution = N, 1 if position is not None else 0, grid_resolution
    g.radius, g.max_vel = arr(radius, np.float64), arr(max_vel, np.float64)
    out = np.zeros((N, cfg.M, 6), np.float32) if sfcs is None else np.ascontiguousarray(sfcs, dtype=np.float32).copy()
    codes = np.zeros(N, np.int32)
    g.sfcs, g.out_code = out.ctypes.data, codes.ctypes.data
    if position is not None:
        g.position = arr(position, np.float32)
    else:
        g.initial_traj, g.goals, g.waypoints = arr(initial_traj, np.float32), arr(goals, np.float32), arr(waypoints, np.float32)
    boxes = np.ascontiguousarray(world["boxes"], dtype=np.float64).reshape(-1, 6)
    wmin = np.ascontiguousarray(world["world_min"], dtype=np.float32)
    wmax = np.ascontiguousarray(world["world_max"], dtype=np.float32)
    rc = L.emul_generate_sfc(C.byref(cfg), boxes.ctypes.data, len(boxes), world["resolution"], wmin.ctypes.data, wmax.ctypes.data, C.byref(g))
    assert rc == 0
    return out, codes
