#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD CONTAINER only).

    python tests/golden/make_golden.py

Writes next to this file:
  missions_empty.json  start / goal / vehicle numbers of missions/empty10/..._1.json and missions/empty60/..._1.json
  world_maze1.json, world_forest1.json
                     obstacle boxes of world/maze_icra2023/dense/maze1.csv and world/forest_tro2022/forest1.csv with the
                     start / goal / vehicle / obstacle numbers of the first mission file that the reference pairs with
                     them (maze10_dense/maze10_1.json; forest10_spin4_100/forest_spin4_100_1.json) and the launch
                     parameters of simulation.launch / testall_DLSCGC_3D.launch
  kat_log.json       known-answer values copied (as numbers) out of the reference's own run log
                     /root/reference/log/result_1742185870.978562_DLSCGC_10agents.csv -- the only
                     artifact of the reference that pins results of the trajectory QP -- together
                     with the mission/launch parameters of that run (maze10_dense/maze10_1.json,
                     launch/simulation.launch).
  golden_<name>.npz  inputs of agent-QP batches (raw structured arrays) and the answers of an
                     independent solver: scipy's bundled HiGHS on the numpy model twin
                     (tests/model_twin.py) for the QP, scipy.optimize.linprog(highs) for the goal LP.
                     Every stored QP answer passed the solver-independent KKT certificate of
                     tests/kkt.py (x_source says which solver produced it).
Neither CPLEX nor any code of the reference is involved; nothing here is read on the GPU box
except the written data files.
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dlsc_gc_planner_amd import abi, workloads  # noqa: E402
import highs_ref  # noqa: E402
import kkt  # noqa: E402
import model_twin  # noqa: E402
import oracle_lib  # noqa: E402
import randqp  # noqa: E402

REF_LOG = "/root/reference/log/result_1742185870.978562_DLSCGC_10agents.csv"
REF_SUMMARY = "/root/reference/log/summary_DLSCGC_10agents.csv"


def make_kat():
    rows = list(csv.reader(open(REF_LOG)))
    data = [[float(v) for v in r[:120]] for r in rows[1:4]]
    out = {
        "source": "log/result_1742185870.978562_DLSCGC_10agents.csv data rows 1-3 (t = 0, 0.1, 0.2)",
        "columns": "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time per agent (multi_sync_simulator.cpp:741-762)",
        "mission": "missions/maze10_dense/maze10_1.json, launch/simulation.launch",
        "config": {"dim": 2, "M": 10, "n": 5, "dt": 0.2, "control_input_weight": 0.01, "terminal_weight": 1.0,
                   "communication_range": 3.0, "world_min": [-2.0, -0.3, 0.0], "world_max": [6.0, 4.3, 2.5],
                   "world_z_2d": 1.0, "max_vel": 1.0, "max_acc": 2.0, "radius": 0.15, "nominal_velocity": 1.0,
                   "grid_resolution": 0.5},
        "agents": [],
    }
    for a in range(10):
        c = a * 12
        out["agents"].append({
            "id": int(data[0][c]),
            "start": data[0][c + 2:c + 5],
            "t0.1": {"pos": data[1][c + 2:c + 5], "vel": data[1][c + 5:c + 8], "acc": data[1][c + 8:c + 11]},
            "t0.2": {"pos": data[2][c + 2:c + 5], "vel": data[2][c + 5:c + 8], "acc": data[2][c + 8:c + 11]},
        })
    srow = list(csv.reader(open(REF_SUMMARY)))
    out["summary"] = dict(zip(srow[0], srow[1]))
    json.dump(out, open(os.path.join(HERE, "kat_log.json"), "w"), indent=1)
    print("kat_log.json written")


def make_missions():
    """Start/goal/vehicle numbers of two of the reference's mission files (inputs for the closed-loop test)."""
    out = {}
    for name, path in (("empty10_1", "/root/reference/missions/empty10/multi_random_10agents_1.json"),
                       ("empty60_1", "/root/reference/missions/empty60/multi_random_60agents_1.json")):
        m = json.load(open(path))
        q = m["quadrotors"]["crazyflie"]
        out[name] = dict(source="missions/" + path.split("/missions/")[1], world=m["world"][0]["dimension"],
                         start=[a["start"] for a in m["agents"]], goal=[a["goal"] for a in m["agents"]],
                         max_vel=q["max_vel"][0], max_acc=q["max_acc"][0], radius=q["radius"], downwash=q["downwash"],
                         nominal_velocity=q["nominal_velocity"])
    json.dump(out, open(os.path.join(HERE, "missions_empty.json"), "w"))
    print("missions_empty.json written")


def make_worlds():
    """Obstacle boxes + mission numbers of the two obstacle worlds the closed-loop tests fly (data only)."""
    ref = "/root/reference/"
    for name, world, mission, extra in (
            ("world_maze1.json", "world/maze_icra2023/dense/maze1.csv", "missions/maze10_dense/maze10_1.json", {}),
            ("world_forest1.json", "world/forest_tro2022/forest1.csv", "missions/forest10_spin4_100/forest_spin4_100_1.json",
             # launch/testall_DLSCGC_3D.launch
             dict(dim=3, M=10, dt=0.2, communication_range=3.0, slack_collision_weight=100.0, uncertainty_horizon=0.7,
                  obs_duration=60.0, goal_threshold=0.6))):
        boxes = [[float(x) for x in row] for row in csv.reader(open(ref + world)) if row]
        m = json.load(open(ref + mission))
        q = m["quadrotors"]["crazyflie"]
        out = dict(source="%s (rows: centre x,y,z, size x,y,z) and %s of the reference" % (world, mission),
                   world_resolution=0.1, grid_resolution=0.5, world=m["world"][0]["dimension"], boxes=boxes,
                   start=[a["start"] for a in m["agents"]], goal=[a["goal"] for a in m["agents"]],
                   radius=q["radius"], max_vel=q["max_vel"][0], max_acc=q["max_acc"][0])
        if m.get("obstacles"):
            out["downwash"] = q["downwash"]
            out["obstacles"] = m["obstacles"]
        out.update(extra)
        json.dump(out, open(os.path.join(HERE, name), "w"))
        print(name, "written:", len(boxes), "boxes,", len(out["start"]), "agents,", len(m.get("obstacles", [])), "obstacles")


def solve_batch_highs(cfg, batch):
    """Independent answers: goal LP by linprog(highs), QP by HiGHS on the numpy twin model."""
    from scipy.optimize import linprog
    agents, ro, rows, sfc = batch["agents"], batch["row_offsets"], batch["rows"], batch["sfc"]
    N = len(agents)
    dim, M, n = cfg.world_dimension, cfg.M, cfg.n
    mso = batch["max_slack_obs"]
    nx = dim * M * (n + 1) + M * mso
    goal_t = np.full(N, np.nan)
    goal_feasible = np.ones(N, bool)
    x = np.zeros((N, nx))
    qp_status = np.zeros(N, np.int32)
    qp_obj = np.zeros(N)
    x_source = np.zeros(N, np.int32)  # 0: HiGHS (KKT-certified), 1: oracle, KKT-certified
    for a in range(N):
        ag = agents[a:a + 1].copy()
        r = rows[ro[a]:ro[a + 1]]
        box = None if sfc is None else sfc[a]
        lp = model_twin.goal_lp_rows(cfg, ag, r, box)
        if lp is None:
            goal_t[a] = 0.0
        else:
            A, b = lp
            if len(A):
                res = linprog([1.0], A_ub=-A.reshape(-1, 1), b_ub=b, bounds=[(0.0, 1.0 + 1e-5)], method="highs")
            else:
                res = linprog([1.0], bounds=[(0.0, 1.0 + 1e-5)], method="highs")
            if res.status == 0:
                goal_t[a] = res.x[0]
            else:
                goal_feasible[a] = False
        # the QP is posed with the oracle's goal (bit-exact float32 chain of goal_optimizer.cpp:51)
        og = oracle_lib.goal_solve(cfg, ag, r, box)
        if og["status"] == abi.INFEASIBLE:
            qp_status[a] = -1  # never reached in the reference
            continue
        ag["current_goal_point"][0] = og["goal"]
        md = model_twin.traj_model(cfg, ag, r, box)
        st, xs, obj = highs_ref.solve_model(md)
        if st == "Optimal":
            cert = kkt.certificate(md, xs)
            if max(cert["primal"], cert["stationarity"]) > 1e-7:
                # HiGHS' QP code occasionally stops at a point that misses the continuity equalities by
                # ~1e-4 m on these ill-conditioned models.  Such an answer is replaced by the oracle's,
                # but only after the solver-independent KKT certificate (tests/kkt.py) has accepted it.
                osol = oracle_lib.traj_solve(cfg, ag, r, box)
                c2 = kkt.certificate(md, osol["x"])
                assert osol["status"] == abi.OK and max(c2["primal"], c2["stationarity"]) < 1e-9, (a, c2)
                xs, obj = osol["x"], osol["cost"]
                x_source[a] = 1
            x[a, :md["nv"]] = xs
            qp_obj[a] = obj
        elif st == "Infeasible":
            qp_status[a] = abi.INFEASIBLE
        else:
            raise RuntimeError("HiGHS status %s for agent %d" % (st, a))
    return dict(goal_t=goal_t, goal_feasible=goal_feasible, x=x, qp_status=qp_status, qp_obj=qp_obj, x_source=x_source)


def save(name, cfg_kwargs, cfg, batch, ans):
    path = os.path.join(HERE, "golden_%s.npz" % name)
    np.savez_compressed(path, cfg_kwargs=json.dumps(cfg_kwargs), agents=batch["agents"], row_offsets=batch["row_offsets"],
                        rows=batch["rows"], sfc=np.zeros(0, np.float32) if batch["sfc"] is None else batch["sfc"],
                        max_slack_obs=batch["max_slack_obs"], **ans)
    print(name, "agents", len(batch["agents"]), "rows", len(batch["rows"]), "qp status", np.bincount(ans["qp_status"] + 1),
          "goal infeasible", int((~ans["goal_feasible"]).sum()), "certified-oracle answers", int(ans["x_source"].sum()), "->", os.path.getsize(path) // 1024, "KiB")


def main():
    only = sys.argv[1:]  # names of randqp configurations: (re)generate just those golden batches
    if only:
        for name in only:
            ck, gk = randqp.CONFIGS[name]
            cfg = abi.make_config(**ck)
            batch = randqp.random_batch(cfg, 24, seed=4242, **gk)
            save(name, ck, cfg, batch, solve_batch_highs(cfg, batch))
        return
    make_kat()
    make_missions()
    make_worlds()
    # S64 (BASELINE.json configs[1]) closed loop driven by the oracle
    cfg = workloads.s64_config()

    def solver(agents, ro, rows, sfc):
        return oracle_lib.plan_batch(cfg, agents, ro, rows, sfc, 0, True)

    wl = workloads.generate(cfg, solver, n_swarms=1, swarm_size=64, seed=2000)
    batch = dict(agents=wl["agents"], row_offsets=wl["row_offsets"], rows=wl["rows"], sfc=None, max_slack_obs=0)
    save("s64", dict(workload="s64"), cfg, batch, solve_batch_highs(cfg, batch))
    for name, (ck, gk) in randqp.CONFIGS.items():
        cfg = abi.make_config(**ck)
        batch = randqp.random_batch(cfg, 24, seed=4242, **gk)
        save(name, ck, cfg, batch, solve_batch_highs(cfg, batch))


if __name__ == "__main__":
    main()
