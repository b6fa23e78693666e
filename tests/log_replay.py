"""Replay of the reference's committed run with the CPU oracle -- TEST INFRASTRUCTURE ONLY.

The only artefact of the reference that pins results of its CPLEX path is one recorded run,
log/result_1742185870.978562_DLSCGC_10agents.csv (maze10_dense/maze10_1.json, launch/simulation.launch: 2-D, M = 10,
ten agents, 171 replanning steps, states printed every 0.1 s with six significant digits).  This module re-enacts that
run step by step with the oracle's restatements of every stage of TrajPlanner::planImpl (traj_planner.cpp:108-133):

    obstaclePrediction + initialTrajPlanning   shift of the previous plans            (traj_planner.cpp:290-327, 409-433)
    constructLSC -> generateDLSCGC             oracle/dlsc_lsc_oracle.c               (:603-666)
    constructSFC -> generateSFC                oracle/dlsc_sfc_oracle.c               (:692-706)
    goalPlanning -> GoalOptimizer::solve       oracle/dlsc_oracle.c                   (:491-496)
    trajOptimization -> TrajOptimizer::solve   oracle/dlsc_oracle.c                   (:737-748)
    AgentManager::doStep                       state := plan(dt), float32             (agent_manager.cpp:34-57)

so that the inputs of every one of the 1 710 goal LPs and trajectory QPs of the run are reconstructed, and the plan that
comes out can be compared with what CPLEX produced: the log's position / velocity / acceleration at t + 0.1 and t + 0.2.
The one input the log does not hold is each agent's next_waypoint (the grid planner's output, multi_sync_simulator.cpp:
308-466); tests/golden/mine_log.py infers it per agent and step from the log itself (the waypoint is a node of the 0.5 m
grid next to the previous one), and tests/golden/kat_replay.npz stores the inferred sequence.  The replay takes the
waypoints from there (or from a waypoint planner under test).
"""
import numpy as np

from dlsc_gc_planner_amd import abi
import oracle_lib as O

# launch/simulation.launch + missions/maze10_dense/maze10_1.json
LAUNCH = dict(dim=2, M=10, n=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, slack_collision_weight=1.0,
              communication_range=3.0, world_min=(-2.0, -0.3, 0.0), world_max=(6.0, 4.3, 2.5), world_z_2d=1.0,
              world_use_octomap=True)
QUAD = dict(max_vel=1.0, max_acc=2.0, radius=0.15, downwash=2.0, nominal_velocity=1.0)
GRID_RES, WORLD_RES = 0.5, 0.1


def make_cfg(**over):
    kw = dict(LAUNCH)
    kw.update(over)
    return abi.make_config(**kw)


def linf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float32) - np.asarray(b, np.float32))))


class Replay:
    """State of the ten agents between two replanning steps, as the reference holds it."""

    def __init__(self, world, start, cfg=None):
        self.cfg = cfg or make_cfg()
        self.N = len(start)
        c = self.cfg
        self.pos = np.array(start, np.float32)
        self.pos[:, 2] = np.float32(c.world_z_2d)  # agent_manager.cpp:59-61
        self.vel = np.zeros((self.N, 3), np.float32)
        self.acc = np.zeros((self.N, 3), np.float32)
        self.goal = self.pos.copy()      # current_goal_point (agent_manager.cpp:9)
        self.wp = self.pos.copy()        # next_waypoint      (agent_manager.cpp:10)
        self.plan = None                 # desired_traj of every agent, [N][M][n+1][3] float32
        self.sfc = np.zeros((self.N, c.M, 6), np.float32)
        self.seq = 0                     # planner_seq of the NEXT plan, starting at 0
        self.world = O.SfcWorld(world["boxes"], world["world_resolution"], c.world_min[:], c.world_max[:], QUAD["radius"],
                                world["grid_resolution"], QUAD["max_vel"], c.dt)
        self.radius = np.full(self.N, QUAD["radius"])
        self.downwash = np.full(self.N, QUAD["downwash"])

    def close(self):
        self.world.close()

    # -- the stages before the optimizers ------------------------------------------------------------------
    def neighbours(self, i):
        """broadcastMsgs (multi_sync_simulator.cpp:482-503): the other agents within communication range, by id."""
        r = self.cfg.communication_range
        return [j for j in range(self.N) if j != i and not (r > 0 and linf(self.pos[i], self.pos[j]) > r)]

    def initial_trajs(self):
        """initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol (traj_planner.cpp:409-421, 290-314)."""
        c = self.cfg
        if self.seq == 0:  # planner_seq < 2: constant-velocity trajectories
            return np.stack([O.const_vel_plan(self.pos[i], self.vel[i], c.M, c.n, c.dt) for i in range(self.N)])
        return np.stack([O.shift_plan(self.plan[i]) for i in range(self.N)])

    def step_inputs(self, i, init, wp_i):
        """Everything TrajPlanner hands to the two optimizers for agent i, if its next_waypoint is wp_i."""
        c = self.cfg
        nb = self.neighbours(i)
        _, rows = O.generate_dlscgc(c, init, self.goal, self.radius, self.downwash, [0, len(nb)], nb, local_ids=[i])
        if self.seq == 0:
            ok, boxes = self.world.initialize(self.pos[i], c.M)  # initializeSFC, collision_constraints.cpp:435-452
            code = 0 if ok else -1
        else:
            code, boxes = self.world.construct(init[i], self.goal[i], wp_i, self.sfc[i])  # :502-536
        ag = np.zeros(1, abi.AGENT_DTYPE)
        ag["position"][0], ag["velocity"][0], ag["acceleration"][0] = self.pos[i], self.vel[i], self.acc[i]
        ag["current_goal_point"][0], ag["next_waypoint"][0] = self.goal[i], wp_i
        ag["id"], ag["n_obs"], ag["n_slack_obs"] = i, len(nb), 0
        ag["max_vel"], ag["max_acc"], ag["radius"], ag["nominal_velocity"] = QUAD["max_vel"], QUAD["max_acc"], QUAD["radius"], QUAD["nominal_velocity"]
        return dict(agent=ag, rows=rows, sfc=boxes, sfc_code=code, neighbours=nb)

    @staticmethod
    def solve(cfg, inp):
        """goalPlanning + trajOptimization with the oracle: returns (goal, goal_status, ctrl, status, x)."""
        g = O.goal_solve(cfg, inp["agent"], inp["rows"], inp["sfc"])
        ag = inp["agent"].copy()
        if g["status"] != abi.INFEASIBLE:
            ag["current_goal_point"][0] = g["goal"]
        r = O.traj_solve(cfg, ag, inp["rows"], inp["sfc"])
        return dict(goal=ag["current_goal_point"][0].copy(), goal_status=g["status"], goal_t=g["t"], ctrl=r["ctrl"], status=r["status"],
                    x=r["x"], cost=r["cost"], agent=ag)

    def commit(self, wps, sols, inputs):
        """All agents have planned: waypoints, sub-goals, corridors and plans become the state of the next step;
        AgentManager::doStep moves every agent to its plan's state at dt."""
        c = self.cfg
        new_plan = np.zeros((self.N, c.M, c.n + 1, 3), np.float32)
        for i in range(self.N):
            self.wp[i] = wps[i]
            self.goal[i] = sols[i]["goal"]
            self.sfc[i] = inputs[i]["sfc"]
            new_plan[i] = sols[i]["ctrl"]
            s = O.state_at(sols[i]["ctrl"], c.dt, c.dt)
            self.pos[i], self.vel[i], self.acc[i] = s[0], s[1], s[2]
            self.pos[i, 2] = np.float32(c.world_z_2d)
        self.plan = new_plan
        self.seq += 1


def log_states(log, k, i):
    """Expected (pos, vel, acc) at t_k + 0.1 and t_k + 0.2 of agent i from the log array [rows][N][9]; the second is None
    for the last step."""
    a = log[2 * k + 1, i].reshape(3, 3)
    b = log[2 * k + 2, i].reshape(3, 3) if 2 * k + 2 < len(log) else None
    return a, b


def ulp32(v):
    return float(np.spacing(np.float32(abs(v))))


def print_unit(v):
    """One unit of the last of the six significant digits iostreams print for v."""
    a = abs(float(v))
    return 1e-6 if a < 1e-300 else 10.0 ** (np.floor(np.log10(a)) - 5)


def errors(ctrl, dt, exp01, exp02):
    """|plan sample - log| per quantity (position, velocity, acceleration), each in units of its tolerance, worst over
    the axes and the two samples.  Tolerance of a logged number v of an agent whose coordinates are at most p metres:
        position      1.5 print units of v + 1e-6 m            (CPLEX's answers are good to about 1e-6 m: the log's own
                                                                mirror-symmetric agents differ by that much)
        velocity      1.5 print units + n/dt * (4 ulp32(p) + 1e-6 m / 5)
        acceleration  1.5 print units + n(n-1)/dt^2 * (4 ulp32(p) + 1e-6 m / 25)
    because the reference stores control points as float32 and differentiates them in float32 (trajectory.cpp:182-199):
    the factors n/dt = 25 and n(n-1)/dt^2 = 500 multiply the rounding of the positions, and two float64 solutions that
    differ in the ninth digit round to different float32 neighbours."""
    out = np.zeros(3)
    for tau, exp in ((0.1, exp01), (0.2, exp02)):
        if exp is None:
            continue
        s = O.state_at(ctrl, dt, tau).astype(np.float64)
        for k in range(3):
            u = ulp32(max(abs(exp[0][k]), 0.5))
            tol = (1.5 * print_unit(exp[0][k]) + 1e-6,
                   1.5 * print_unit(exp[1][k]) + 25.0 * (4 * u + 2e-7),
                   1.5 * print_unit(exp[2][k]) + 500.0 * (4 * u + 4e-8))
            for q in range(3):
                out[q] = max(out[q], abs(s[q][k] - exp[q][k]) / tol[q])
    return out


def mismatch(ctrl, dt, exp01, exp02):
    """<= 1: the plan reproduces the log's two samples within the log's precision (see errors())."""
    return float(errors(ctrl, dt, exp01, exp02).max())


def load_kat():
    """tests/golden/kat_replay.npz (written by tests/golden/mine_log.py): the log's numbers and the inferred waypoints."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_replay.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_world():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_maze1.json")))


_RUN_CACHE = {}


def make_mapf(cfg=None):
    """The product's host-side waypoint planner (row N4, csrc/grid_mapf.cpp) set up for the logged mission."""
    from dlsc_gc_planner_amd.harness import MapfWaypoints
    world = load_world()
    return MapfWaypoints(cfg or make_cfg(), world["boxes"], world["world_resolution"], world["grid_resolution"],
                         np.array(world["start"], np.float32), np.array(world["goal"], np.float32), QUAD["radius"], QUAD["downwash"])


def run(anchor=True, steps=None, waypoint_fn=None, own_waypoints=False):
    """Replay the logged run with the oracle and the stored waypoints (or waypoint_fn(sim, k) -> [N][3]).  anchor: every
    step starts from the state the log prints for it (row 2k) instead of this replay's own, which drifts away from the
    reference's by the accumulated ~1e-6 differences between CPLEX's and the oracle's solutions.
    Returns dict(agents, row_offsets, rows, sfc: the inputs of all K*N agent-QPs as ONE batch in step-major order;
    goal_in: sub-goal before the LP; sols: oracle results per QP; score [K][N]; comp [K][N][3]; K, N)."""
    key = (anchor, steps, own_waypoints)
    if waypoint_fn is None and key in _RUN_CACHE:
        return _RUN_CACHE[key]
    kat = load_kat()
    log = kat["log"]
    cfg = make_cfg()
    world = load_world()
    sim = Replay(world, world["start"], cfg)
    K = len(kat["waypoints"]) if steps is None else steps
    N = sim.N
    agents, rows, offs, sfcs, sols = [], [], [0], [], []
    score = np.zeros((K, N))
    comp = np.zeros((K, N, 3))
    # the waypoint planner under test runs alongside: own_waypoints = False -> the replay follows the mined waypoints and
    # only counts where the planner disagrees; True -> the replay follows the planner (nothing but t = 0 comes from the log)
    mapf = make_mapf(cfg) if waypoint_fn is None else None
    wp_diff = np.zeros((K, N), bool)
    for k in range(K):
        if anchor and k > 0:
            for i in range(N):
                st = log[2 * k, i].reshape(3, 3)
                sim.pos[i], sim.vel[i], sim.acc[i] = st[0], st[1], st[2]
        if waypoint_fn is not None:
            wps = waypoint_fn(sim, k)
        else:
            own = sim.wp.copy()
            mapf.step(sim.pos, sim.goal, own, sim.plan)  # MultiSyncSimulator::decentralizedMAPP
            wp_diff[k] = np.abs(own - kat["waypoints"][k]).max(axis=1) > 1e-6
            wps = own if own_waypoints else kat["waypoints"][k]
        init = sim.initial_trajs()
        ins, ss = [], []
        for i in range(N):
            inp = sim.step_inputs(i, init, np.asarray(wps[i], np.float32))
            sol = Replay.solve(cfg, inp)
            e1, e2 = log_states(log, k, i)
            comp[k, i] = errors(sol["ctrl"], cfg.dt, e1, e2) if sol["status"] == abi.OK else np.inf
            score[k, i] = comp[k, i].max()
            agents.append(inp["agent"])
            rows.append(inp["rows"])
            offs.append(offs[-1] + len(inp["rows"]))
            sfcs.append(inp["sfc"])
            ins.append(inp)
            ss.append(sol)
        sols.extend(ss)
        sim.commit(wps, ss, ins)
    sim.close()
    if mapf is not None:
        mapf.close()
    out = dict(wp_diff=wp_diff, cfg=cfg, agents=np.concatenate(agents), row_offsets=np.array(offs, np.int64),
               rows=np.concatenate(rows) if rows else np.zeros(0, abi.ROW_DTYPE), sfc=np.stack(sfcs), sols=sols, score=score,
               comp=comp, K=K, N=N, log=log, kat=kat)
    if waypoint_fn is None:
        _RUN_CACHE[key] = out
    return out


def score_batch(run_out, ctrl, status):
    """Scores [K][N] of a solver's answers (ctrl [K*N][M][n+1][3] float32, status) for the replay's inputs against the log."""
    K, N, log, cfg = run_out["K"], run_out["N"], run_out["log"], run_out["cfg"]
    sc = np.full((K, N), np.inf)
    for k in range(K):
        for i in range(N):
            q = k * N + i
            if status[q] == abi.OK:
                e1, e2 = log_states(log, k, i)
                sc[k, i] = mismatch(ctrl[q], cfg.dt, e1, e2)
    return sc
