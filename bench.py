#!/usr/bin/env python3
"""bench.py -- agent-QP solves/s of the batched DLSC-GC trajectory optimizer on MI355X.

One "step" = one replanning pass of the hot path (goal LP + trajectory QP, one fused kernel
launch) over a batch of synthetic agents already resident in HBM, followed -- when more than one
GPU takes part -- by ONE RCCL all-gather of the solved control points (the reference's
broadcastMsgs, multi_sync_simulator.cpp:468-514).  Workload: BASELINE.json configs[1]
("synthetic 64-agent batch, 5 segments x degree 5, 120 DLSC inequalities each"), `--swarms`
independent 64-agent swarms per GPU per step (weak scaling: every rank owns its own swarms).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 5

Rank 0 prints one JSON line.  `value` = agent-QPs of all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6


def algorithmic_bytes_per_qp(dim, M, n, K, use_sfc, n_dyn):
    """bytes one agent-QP has to move at least."""
    b = 8 * (3 * dim + dim + dim + 2 * dim + 5)
    if use_sfc:
        b += 4 * 6 * M
    b += K * 36
    b += 4 * dim * M * (n + 1)
    b += 8 * (1 + n_dyn) + 4
    return b


def cpu_baseline(cfg, wl, target_seconds=12.0):
    """The CPU oracle (a port: dense model + generic dual active-set, oracle/dlsc_oracle.c) timed on this host on a
    bounded sample of the same workload: one thread first, then one thread per host core over disjoint agent ranges
    (the oracle is re-entrant and ctypes releases the GIL).  `value` is the all-core rate."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    agents, ro, rows = wl["agents"], wl["row_offsets"], wl["rows"]
    N = len(agents)

    def solve(lo, hi):
        oracle_lib.plan_batch(cfg, agents[lo:hi], ro[lo:hi + 1] - ro[lo], rows[ro[lo]:ro[hi]], None, 0, True)

    probe = min(N, 256)
    t0 = time.perf_counter()
    solve(0, probe)
    one = probe / (time.perf_counter() - t0)
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    for fq, fp in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:  # a container's CPU quota, if any, is what the threads can really use
            if fp is None:
                quota, period = open(fq).read().split()
            else:
                quota, period = open(fq).read().strip(), open(fp).read().strip()
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError):
            pass
    # time-bounded: every thread walks its own range of the batch in chunks of 16 agents until the deadline
    per = max(16, N // cores)
    deadline = time.perf_counter() + target_seconds
    done = [0] * cores

    def work(c):
        lo0 = (c * per) % max(N - per + 1, 1)
        pos = 0
        while time.perf_counter() < deadline:
            lo = lo0 + pos
            hi = min(lo + 16, lo0 + per, N)
            solve(lo, hi)
            done[c] += hi - lo
            pos = (pos + 16) % per

    with ThreadPoolExecutor(cores) as pool:
        t0 = time.perf_counter()
        list(pool.map(work, range(cores)))
        dt = time.perf_counter() - t0
    n = sum(done)
    return {"value": n / dt, "unit": "QP/s", "cores": cores, "kind": "port",
            "sample": "%d agent-QPs (goal LP + QP each) of the same batch in %.1f s: %d threads, each walking its own range of "
                      "the batch" % (n, dt, cores),
            "one_core_qp_per_s": one,
            "ref_published_qp_per_s": 89.4,
            "ref_published_note": "reference log: 11.18 ms/QP mean, CPLEX 22.1.1 Threads=6, 2-D M=10, unknown CPU"}


def closed_loop_latency(device):
    """p50 / p99 wall time of one replanning step of the reference's empty10 / empty60 missions (BASELINE.json
    configs[0] and [2]) flown by the device-resident closed loop: half-space generation + goal LP + QP + state update
    for all agents of the mission, synchronised after every step (dlsc_gc_planner_amd/harness.py): three launches per step
    (half-spaces, LP + QP, dlscqp_advance_device); also timed replayed as one HIP graph."""
    import torch
    from dlsc_gc_planner_amd import abi
    from dlsc_gc_planner_amd.harness import ClosedLoop
    path = os.path.join(ROOT, "tests", "golden", "missions_empty.json")
    if not os.path.exists(path):
        return None
    missions = json.load(open(path))
    out = {}
    for name in ("empty10_1", "empty60_1"):
        m = missions[name]
        w = m["world"]
        cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                              world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]), device=device)
        sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"],
                         downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"],
                         nominal_velocity=m["nominal_velocity"], device=device)
        ts, fails = {"eager": [], "graph": []}, 0
        for k in range(110):
            if k == 55:
                sim.enable_graph()  # the same three launches replayed as one HIP graph
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            status, _ = sim.step()
            torch.cuda.synchronize()
            if k % 55 >= 5:
                ts["graph" if k >= 55 else "eager"].append((time.perf_counter() - t0) * 1e3)
            fails += int((status != abi.OK).sum())
        sim.close()
        out[name] = {"agents": len(m["start"]), "lsc_rows_per_agent": (len(m["start"]) - 1) * 27, "steps": len(ts["eager"]),
                     "p50_ms": float(np.percentile(ts["eager"], 50)), "p99_ms": float(np.percentile(ts["eager"], 99)),
                     "p50_ms_as_one_hip_graph": float(np.percentile(ts["graph"], 50)), "qp_failsafes": fails}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--swarms", type=int, default=512, help="64-agent swarms per GPU per step")
    ap.add_argument("--seed", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-swarm host-to-host latency loop")
    ap.add_argument("--no-order-hint", action="store_true",
                    help="launch agents in id order instead of most-expensive-first (previous step's iteration counts)")
    ap.add_argument("--exchange", choices=("pipelined", "serial"), default="pipelined",
                    help="N>1: 'pipelined' posts the all-gather of step k on a communication stream so that it runs while "
                         "step k+1 plans the agents whose neighbours are all local (every agent of this workload: shards "
                         "are whole swarms); 'serial' runs launch and all-gather back to back on one stream")
    ap.add_argument("--force-dist", action="store_true", help="(test) create the process group even for one rank")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic_latest.json"))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trajectory optimizer has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL's kernels on a high-priority stream: with the pipelined exchange they share the GPU with the plan kernel of
        # the next step and should get the first free workgroup slots
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=opts)

    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

    cfg = workloads.s64_config(device=local_rank)
    opt = BatchTrajOptimizer(cfg)

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    t_gen = time.perf_counter()
    wl = workloads.generate(cfg, solver, n_swarms=args.swarms, swarm_size=64, seed=args.seed + 1000 * rank)
    t_gen = time.perf_counter() - t_gen
    N = len(wl["agents"])
    K = wl["K"]
    db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
    shape1 = tuple(db.out_ctrl.shape[1:])

    for _ in range(args.warmup):
        db.launch(True, True)
    torch.cuda.synchronize(dev)
    if not args.no_order_hint:
        # closed-loop planners replan the same agents every 0.2 s and an agent's difficulty changes slowly:
        # launch the agents that needed most active-set iterations in the previous step first
        prev_iters = db.out_iters.cpu().numpy()[:, 0]
        db.set_order(np.argsort(-prev_iters, kind="stable").astype(np.int32))
    pipelined = False
    if use_dist:
        from dlsc_gc_planner_amd import sharding
        # Which agents of step k+1 need the gathered table of step k?  Those with a neighbour on another rank.  Every
        # rank owns whole swarms here (an agent's neighbours are the other 63 agents of its swarm), so there are none:
        # the exchange of step k may run while step k+1 is planned.
        lo = rank * N
        members = lo + (np.arange(N) // 64 * 64)[:, None] + np.arange(64)[None, :]  # [N, 64] ids of the agent's swarm
        nbr_ids = members[members != (lo + np.arange(N))[:, None]]                 # ... without the agent itself
        halo = sharding.halo_agents(np.arange(N + 1) * 63, nbr_ids, lo, lo + N)
        pipelined = args.exchange == "pipelined" and len(halo) == 0
        # per agent: control points, sub-goal, status (376 B at S64) -- the broadcast payload of the reference
        xchg = sharding.PlanExchange(dist, world, N, (int(np.prod(shape1)), 3, 1), dev)

    def step():
        if not use_dist:
            db.launch(True, True)
        elif pipelined:
            # outputs double-buffered: the exchange of step k (packing included) runs on the communication stream while
            # step k+1 writes the other set; it is complete before the step after the next is posted
            db.flip_outputs()
            xchg.before_overwrite()
            db.launch(True, True)
            xchg.post(db.out_ctrl, db.out_goal, db.out_status, pack_on_comm=True)
        else:
            db.launch(True, True)
            xchg.post(db.out_ctrl, db.out_goal, db.out_status)
            xchg.table()  # the next launch waits for the exchange

    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    opt.set_timing(args.steps)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if use_dist:
        # the gathered table must hold this rank's own block
        ctrl_all, goal_all, status_all = xchg.unpack(xchg.table(), (torch.float32, torch.float32, torch.int32))
        torch.cuda.synchronize(dev)
        own = slice(rank * N, (rank + 1) * N)
        assert torch.equal(ctrl_all[own], db.out_ctrl.view(N, -1)) and torch.equal(goal_all[own], db.out_goal)
        assert torch.equal(status_all[own].flatten(), db.out_status)
    kms = opt.kernel_times_ms(args.steps)
    res = db.results()

    if rank == 0:
        ok = res["status"] == abi.OK
        dim, M, n = cfg.world_dimension, cfg.M, cfg.n
        bytes_qp = algorithmic_bytes_per_qp(dim, M, n, K, False, 0)
        avg_kernel_s = float(np.mean(kms)) * 1e-3 if len(kms) else float("nan")
        achieved = N * bytes_qp / avg_kernel_s / 1e9
        traffic = None
        try:
            with open(args.traffic_json) as fh:
                tj = json.load(fh)
            if tj.get("agents_per_launch") == N:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            pass
        it = res["iters"]
        R_rows, r_dim = K + 2 * 207, dim * (M * (n + 1) - 17)
        # n_W grows from 0 to `active` over the adds: sum over iterations ~ iterations * active / 2
        flops_qp = float((it[:, 0] * (6.0 * R_rows + 2.0 * r_dim * r_dim) + 6.0 * r_dim * it[:, 0] * it[:, 3] / 2.0).mean()
                         + 6.0 * R_rows)  # + the scan that finds no violated row
        fp64 = {"flops_per_qp_model": flops_qp, "achieved_tflops": N * flops_qp / avg_kernel_s / 1e12,
                "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac": N * flops_qp / avg_kernel_s / 1e12 / FP64_VALU_PEAK_TFLOPS}
        # single-swarm replanning latency through the host-buffer entry (H2D + solve + D2H)
        a1, r1 = wl["agents"][:64], wl["rows"][:wl["row_offsets"][64]]
        ro1 = wl["row_offsets"][:65]
        lat = []
        for _ in range(0 if args.no_latency else 60):
            t1 = time.perf_counter()
            opt.plan(a1, ro1, r1, None, 0, True, True)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = np.array(lat[10:]) if lat else np.array([float("nan")])
        closed = None
        if not args.no_latency and world == 1:
            closed = closed_loop_latency(local_rank)
        line = {
            "metric": "agent-QP solves/sec (goal LP + trajectory QP per agent per replanning step)",
            "value": world * N * args.steps / elapsed,
            "unit": "QP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "S64: %d independent 64-agent swarms per GPU per step (BASELINE.json configs[1]: "
                            "3-D, 5 segments x degree 5, %d DLSC rows per agent)" % (args.swarms, K),
                "agents_per_gpu": N, "segments": M, "degree": n, "lsc_rows_per_agent": K,
                "vars_per_agent": dim * M * (n + 1), "ineq_rows_per_agent": K + 2 * 207,
                "exchange": ("none (1 GPU)" if not use_dist else
                             "one RCCL all-gather of control points, sub-goal and status per step, posted on a communication stream: runs "
                             "while the next step plans the agents without remote neighbours (all of them here)"
                             if pipelined else "one RCCL all-gather of control points, sub-goal and status per step, after the launch"),
                "launch_order": "id" if args.no_order_hint else "previous step's iteration count, descending",
                "solved_fraction": float(ok.mean()), "mean_active_set_iterations": float(it[:, 0].mean()),
                "max_active_set_iterations": int(it[:, 0].max()), "seed": args.seed,
                "workload_generation_s": t_gen,
            },
            "p50_replan_ms_one_swarm_host_to_host": float(np.percentile(lat, 50)),
            "p99_replan_ms_one_swarm_host_to_host": float(np.percentile(lat, 99)),
            "closed_loop_replan_ms": closed,
            # SURVEY.md 8(d) flop model: per active-set iteration 6 R (row residuals) + 2 r^2 (two r x r sweeps) + 6 r n_W
            # (factor update at working-set size n_W), R = inequality rows, r = reduced dimension
            "fp64_valu": fp64,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "dlscqp_plan_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_qp": bytes_qp, "qp_per_launch": N,
                "note": "latency/FP64-VALU bound by construction (SURVEY.md 8d): ~5 KB of traffic per QP",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, wl)
        print(json.dumps(line))
    opt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
