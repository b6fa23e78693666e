"""Random agent-QP batches for parity tests (test infrastructure).

Unlike dlsc_gc_planner_amd.workloads (closed-loop, realistic) these are adversarial mixes that
exercise every row family of populatebyrow (traj_optimizer.cpp:225-527): SFC boxes,
communication range, static and dynamic obstacles with slack variables, 2-D and 3-D, ragged
row counts, zero normals, rows on the untouchable initial control points, and infeasible sets.
"""
import numpy as np

from dlsc_gc_planner_amd import abi


def random_batch(cfg, n_agents, seed, max_rows=60, n_dyn=0, infeasible_frac=0.1, tight=0.08, n_static_max=3, keep=0.4):
    rng = np.random.default_rng(seed)
    dim, M, n = cfg.world_dimension, cfg.M, cfg.n
    wmin = np.array(cfg.world_min[:])
    wmax = np.array(cfg.world_max[:])
    agents = np.zeros(n_agents, abi.AGENT_DTYPE)
    rows = []
    offsets = [0]
    sfc = np.zeros((n_agents, M, 6), np.float32) if cfg.world_use_octomap else None
    for a in range(n_agents):
        pos = rng.uniform(wmin + 1.0, wmax - 1.0)
        if dim == 2:
            pos[2] = cfg.world_z_2d
        vel = rng.uniform(-0.5, 0.5, 3)
        acc = rng.uniform(-0.6, 0.6, 3)
        if dim == 2:
            vel[2] = acc[2] = 0
        if rng.random() < 0.3:
            vel[:] = 0
            acc[:] = 0
        if cfg.communication_range > 0:
            # the waypoint must stay within half the communication range of every segment end
            # (traj_optimizer.cpp:505-512); the grid planner hands out waypoints <= 0.5 m away
            goal = pos + rng.uniform(-0.9, 0.9, 3)
            wp = goal + rng.uniform(-0.3, 0.3, 3)
        else:
            goal = pos + rng.uniform(-2.5, 2.5, 3)
            wp = goal + rng.uniform(-0.8, 0.8, 3)
        if dim == 2:
            goal[2] = wp[2] = cfg.world_z_2d
        if rng.random() < 0.15:
            wp = goal.copy()
        ag = agents[a]
        ag["position"], ag["velocity"], ag["acceleration"] = pos, vel, acc
        ag["current_goal_point"], ag["next_waypoint"] = goal, wp
        ag["id"] = a
        ag["max_vel"], ag["max_acc"], ag["radius"], ag["nominal_velocity"] = 1.0, 2.0, 0.15, 1.0
        n_static = int(rng.integers(0, n_static_max + 1))
        nd = int(rng.integers(0, n_dyn + 1)) if n_dyn > 0 else 0
        ag["n_obs"] = n_static + nd
        ag["n_slack_obs"] = nd
        # rough straight-line guess of where the control points will be
        horizon = M * cfg.dt
        guess = lambda m, i: pos + vel * min((m + i / n) * cfg.dt, horizon) * 0.5  # noqa: E731
        agent_rows = []
        infeasible = rng.random() < infeasible_frac
        for oi in range(n_static + nd):
            dyn = oi >= n_static
            slack = oi - n_static if dyn else -1
            opos = pos + rng.uniform(-2.0, 2.0, 3)
            if dim == 2:
                opos[2] = cfg.world_z_2d
            for m in range(M):
                nrm = pos - opos + rng.normal(0, 0.2, 3)
                if dim == 2:
                    nrm[2] = 0
                nn = np.linalg.norm(nrm)
                nrm = nrm / nn if nn > 1e-6 else np.array([1.0, 0, 0])
                if dim == 3:
                    nrm[2] /= 2.0  # downwash
                if rng.random() < 0.05:
                    nrm[:] = 0  # zero normal: ignored row (traj_optimizer.cpp:422-424)
                for i in range(n + 1):
                    if len(agent_rows) >= max_rows:
                        break
                    if rng.random() >= keep:
                        continue
                    r = np.zeros(1, abi.ROW_DTYPE)[0]
                    r["normal"] = nrm
                    r["point"] = opos
                    g = guess(m, i)
                    margin = float(np.dot(np.float32(nrm).astype(float), g - np.float32(opos).astype(float)))
                    r["d"] = margin - rng.uniform(-tight, 1.0)  # sometimes the guess violates it
                    r["m"], r["i"], r["obs"], r["slack"] = m, i, oi, slack
                    r["flags"] = abi.ROW_DYNAMIC if dyn else 0
                    agent_rows.append(r)
        if infeasible and len(agent_rows) >= 2:
            r0, r1 = agent_rows[-1].copy(), agent_rows[-2].copy()
            nrm = np.array([0.8, 0.6, 0.0], np.float32)
            r0["normal"], r1["normal"] = nrm, -nrm
            r0["point"] = r1["point"] = pos
            r0["d"] = r1["d"] = 0.2
            r0["m"] = r1["m"] = M - 1
            r0["i"] = r1["i"] = 1
            r0["slack"] = r1["slack"] = -1
            r0["flags"] = r1["flags"] = 0
            agent_rows[-1], agent_rows[-2] = r0, r1
        rows.extend(agent_rows)
        offsets.append(len(rows))
        if sfc is not None:
            for m in range(M):
                c = guess(m, n / 2)
                half = rng.uniform(1.0, 3.0, 3)
                sfc[a, m, :3] = c - half
                sfc[a, m, 3:] = c + half
                if rng.random() < 0.02:
                    sfc[a, m, 3] = sfc[a, m, 0] - 0.1  # empty box: infeasible
    rows_arr = np.array(rows, dtype=abi.ROW_DTYPE) if rows else np.zeros(0, abi.ROW_DTYPE)
    return dict(agents=agents, row_offsets=np.array(offsets, np.int64), rows=rows_arr, sfc=sfc,
                max_slack_obs=int(agents["n_slack_obs"].max()) if n_agents else 0)


CONFIGS = {
    # name: (config kwargs, generator kwargs)
    "empty3d_M5": (dict(dim=3, M=5), dict(max_rows=80)),
    "maze2d_M10_sfc_comm": (dict(dim=2, M=10, communication_range=3.0, world_use_octomap=True,
                                 world_min=(-2.0, -0.3, 0.0), world_max=(6.0, 4.3, 2.5)), dict(max_rows=120)),
    "forest3d_M10_dyn": (dict(dim=3, M=10, communication_range=3.0, world_use_octomap=True,
                              slack_collision_weight=100.0), dict(max_rows=150, n_dyn=2)),
    # the reference's forest10_spin4 launch shape: four slack obstacles at M = 10 -> 124 reduced dimensions, a 203 KB working
    # set: beyond the LDS of a CU, planned by the kernel variant that keeps its working set in HBM
    "forest3d_M10_dyn4": (dict(dim=3, M=10, communication_range=3.0, world_use_octomap=True,
                               slack_collision_weight=100.0), dict(max_rows=150, n_dyn=4)),
    "forest3d_M5_dyn": (dict(dim=3, M=5, communication_range=3.0, world_use_octomap=True,
                             slack_collision_weight=100.0), dict(max_rows=90, n_dyn=3)),
    "lsc2d_M4_nostop": (dict(dim=2, M=4, planner_mode=abi.PLANNER_BVC), dict(max_rows=40)),
    "rsfc3d_M5_allslack": (dict(dim=3, M=5, planner_mode=abi.PLANNER_RECIPROCALRSFC,
                                slack_mode=abi.SLACK_COLLISIONCONSTRAINT), dict(max_rows=60, n_dyn=2)),
    # more than 256 rows per agent: rows beyond the kernel's register cache are read from HBM on every scan
    "crowd3d_M5_400rows": (dict(dim=3, M=5), dict(max_rows=400, n_static_max=22, keep=0.75, tight=-0.15, infeasible_frac=0.05)),
    # M = 12: runtime-shape kernel, 102 reduced dimensions (two registers per vector), > 640 structured slots
    "long3d_M12_comm": (dict(dim=3, M=12, communication_range=3.0), dict(max_rows=100, tight=0.05)),
    # phi_n > 1: Q_base also carries the acceleration (and velocity) Gram terms (traj_optimizer.cpp:172-187); the kernel's
    # weak-duality infeasibility test, whose objective bound covers the jerk term only, must stay off (KParams::wd_cut)
    "empty3d_M5_phin2": (dict(dim=3, M=5, phi_n=2), dict(max_rows=80, infeasible_frac=0.3)),
    "maze2d_M10_phin3": (dict(dim=2, M=10, phi_n=3, communication_range=3.0, world_use_octomap=True,
                              world_min=(-2.0, -0.3, 0.0), world_max=(6.0, 4.3, 2.5)), dict(max_rows=120, infeasible_frac=0.3)),
    # ---- round 4: OFF the launch files' parameter values.  Every entry above runs control_input_weight 0.01, dt 0.2,
    # terminal_weight 1 (abi.make_config); the reference's CODE defaults are 1 / 0.2 / 1 (src/param.cpp:71-80), and the weights
    # set the conditioning of the reduced Hessian, i.e. what the kernel's dependency thresholds (qp_types.hpp: DEP_*) were
    # calibrated on.  The same row families at the corners of the parameter box:
    "empty3d_M5_wc1": (dict(dim=3, M=5, control_input_weight=1.0), dict(max_rows=80)),
    "empty3d_M5_wc1e-4": (dict(dim=3, M=5, control_input_weight=1e-4), dict(max_rows=80)),
    "empty3d_M5_dt01_wt20": (dict(dim=3, M=5, dt=0.1, terminal_weight=20.0), dict(max_rows=80, infeasible_frac=0.2)),
    "maze2d_M10_sfc_comm_dt04": (dict(dim=2, M=10, dt=0.4, communication_range=3.0, world_use_octomap=True,
                                      world_min=(-2.0, -0.3, 0.0), world_max=(6.0, 4.3, 2.5)), dict(max_rows=120)),
    "maze2d_M10_sfc_comm_wc1_wt20": (dict(dim=2, M=10, control_input_weight=1.0, terminal_weight=20.0, communication_range=3.0,
                                          world_use_octomap=True, world_min=(-2.0, -0.3, 0.0), world_max=(6.0, 4.3, 2.5)), dict(max_rows=120)),
    "forest3d_M5_dyn_ws1": (dict(dim=3, M=5, communication_range=3.0, world_use_octomap=True, slack_collision_weight=1.0),
                            dict(max_rows=90, n_dyn=3)),
    "forest3d_M10_dyn_wc1e-4_dt01": (dict(dim=3, M=10, control_input_weight=1e-4, dt=0.1, communication_range=3.0, world_use_octomap=True,
                                          slack_collision_weight=100.0), dict(max_rows=150, n_dyn=2)),
    "long3d_M12_comm_wc1": (dict(dim=3, M=12, control_input_weight=1.0, communication_range=3.0), dict(max_rows=100, tight=0.05)),
}
# the entries above that leave the launch files' parameter values (tests that want only those)
OFF_DEFAULT = [k for k in CONFIGS if any(t in k for t in ("_wc1", "_dt0", "_wt20", "_ws1"))]
