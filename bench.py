#!/usr/bin/env python3
"""bench.py -- agent-QP solves/s of the batched DLSC-GC trajectory optimizer on MI355X.

One "step" = one replanning pass of the hot path (goal LP + trajectory QP, one fused kernel launch) over a batch of
synthetic agents already resident in HBM, followed -- when more than one GPU takes part -- by ONE RCCL all-gather of the
solved control points (the reference's broadcastMsgs, multi_sync_simulator.cpp:468-514).  Workload: BASELINE.json
configs[1] ("synthetic 64-agent batch, 5 segments x degree 5, 120 DLSC inequalities each"), `--swarms` independent
64-agent swarms per GPU per step (weak scaling: every rank owns its own swarms).

    python bench.py                      # 1 GPU
    python bench.py --gpus 8             # starts 8 ranks itself (python -m torch.distributed.run ... bench.py --gpus 8)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8

What is timed, and how the line avoids flattering itself:
  * the timed loop rotates through `--batches` (4) DIFFERENT batches: the same agents at consecutive replanning steps of
    the closed-loop workload generator, 0.2 s apart;
  * agents are launched most-expensive-first by the iteration counts of the PREVIOUS batch (what a closed loop knows when
    it plans a step), and the order is found where a closed loop finds it: on the device, INSIDE the timed loop, by the plan
    kernel itself from the class masks the previous launch of the same agents left behind (round 5; rounds 1-3: sort kernels
    after every step, `ms_per_step_sorted_on_device`); the same loop in plain id order (`ms_per_step_no_order_hint`) and with
    orders prepared on the host before the loop (`ms_per_step_host_order`) are timed right after and reported beside it;
  * a "hard" batch made only of agents with at least five active rows at their solution is timed too (`hard_batch`);
  * a bracket of `--steps` steps shorter than 50 ms is repeated back to back and the MEDIAN repeat is the line's value
    (`timing`): the line does not depend on `--steps` (round 6);
  * `--swarms` defaults to 1 024 since round 6 (a step cannot be shorter than its slowest agent's 35-46 iterations, which 512
    swarms ran into once the kernel held three wavefronts per SIMD); the 512-swarm loop is reported as `at_512_swarms`.
Rank 0 prints one JSON line.  `value` = agent-QPs of all ranks / max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--swarms", type=int, default=1024,
                    help="64-agent swarms per GPU per step.  A step is never shorter than its slowest agent (30-46 active-set iterations: ~0.2 ms); "
                         "1 024 swarms are what one MI355X needs to be busy for longer than that (round 6: three wavefronts per SIMD).  The same "
                         "loop over the first 512 swarms -- rounds 1-5's batch -- is reported as at_512_swarms")
    ap.add_argument("--batches", type=int, default=4, help="distinct consecutive replanning steps rotated through the timed loop")
    ap.add_argument("--groups", type=int, default=0,
                    help="independent groups the swarms of a GPU are planned in, each on its own stream with its own sequence of replanning steps "
                         "(1 = one launch of all agents per step; default: 4 on one GPU, 2 with the plan exchange of N > 1 -- measured with a "
                         "one-rank process group: 0.31 ms per step with 2 groups, 0.37 with 1, 0.41-0.52 with 4, where the host's calls "
                         "for four groups and the exchange no longer fit the step)")
    ap.add_argument("--seed", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-latency", dest="no_extras", action="store_true",
                    help="skip the closed-loop missions, the stress batch, the hard batch and the host-to-host latency loop")
    ap.add_argument("--exchange", choices=("pipelined", "serial"), default="pipelined",
                    help="N>1, headline loop: 'pipelined' posts the all-gather of step k on a communication stream so that it "
                         "runs while step k+1 plans the agents whose neighbours are all local; 'serial' runs launch and all-gather "
                         "back to back.  The other mode is timed after the headline loop and reported beside it")
    ap.add_argument("--headline-only", action="store_true",
                    help="(profiling) stop after the headline loop: the last --steps launches of the process are the timed ones")
    ap.add_argument("--force-dist", action="store_true", help="(test) create the process group even for one rank")
    ap.add_argument("--lockstep-exchange", action="store_true", help="(measurement) rounds 3-5: a launch waits for the collective of the step before last instead of for the one that read the table it overwrites")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic_latest.json"))
    return ap.parse_args()


def spawn_ranks(n):
    """`bench.py --gpus N` started as ONE process: start N ranks as children (before anything has touched the GPU -- a
    process that has initialised HIP must never exec) and hand their exit code on."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def algorithmic_bytes_per_qp(dim, M, n, K, use_sfc, n_dyn):
    """bytes one agent-QP has to move at least."""
    b = 8 * (3 * dim + dim + dim + 2 * dim + 5)
    if use_sfc:
        b += 4 * 6 * M
    b += K * 36
    b += 4 * dim * M * (n + 1)
    b += 8 * (1 + n_dyn) + 4
    return b


def scaling_model(step_ms_one_rank, n_agents_per_rank):
    """What a 2 / 4 / 8-GPU run of this bench should show, so that a measured SCALE file can be read against it (no node with more than
    one GPU was available to measure it): weak scaling, every rank plans its own agents; ONE all-gather of 376 B per agent per step whose
    transfer runs beside the next step's launches (pipelined: no agent of S64 reads a remote plan):
        step(n) = max(step(1) + post, transfer(n)),   transfer(n) = latency + (n - 1) N 376 B / ring bandwidth"""
    post, lat, ring = 0.02, 0.03, 300e9
    ms = [max(step_ms_one_rank + (0.0 if g == 1 else post), 0.0 if g == 1 else lat + (g - 1) * n_agents_per_rank * 376 / ring * 1e3) for g in (1, 2, 4, 8)]
    return {"n_gpus": [1, 2, 4, 8], "ms_per_step": ms, "qp_per_s": [g * n_agents_per_rank / t * 1e3 for g, t in zip((1, 2, 4, 8), ms)],
            "speedup_over_one_gpu": [g * ms[0] / t for g, t in zip((1, 2, 4, 8), ms)],
            "assumptions": {"step_ms_one_rank": step_ms_one_rank, "post_ms": post, "exchange_latency_ms": lat, "ring_GBps": ring / 1e9, "bytes_per_agent": 376,
                            "note": "post_ms: what the exchange costs a step on one rank before any byte crosses xGMI (--force-dist, its two groups: "
                                    "0.3585 ms, against the same 1 024 swarms in four groups without a process group: 0.3396; "
                                    "profiles/r06_force_dist_trace.md); ring_GBps: bus bandwidth of RCCL's ring all-gather over xGMI (7 links x ~153 GB/s "
                                    "per GPU, point to point: a ring is bound by one link each way; DESIGN.md section 5); the transfer is hidden "
                                    "behind the next step's launches while it is shorter than a step"}}


def host_cores():
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
    return cores


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, wl, seconds=12.0):
    """CPU numbers on this host, same batch (rank 0, N = 1 only):
      value             a CPU-SHAPED SOLVER (kind "port"): tests/native/native_qp.cpp -- the kernel's mathematics (host tables, whitened
                        dual active-set method in range-space form) written a second time as a scalar program: plain loops over an
                        agent's own rows, variables and active rows, g++ -O3 -march=x86-64-v3, one native thread per core the
                        container may use, each walking its own range of the batch for most of `seconds`; parity-checked against
                        the oracle by tests/test_native.py;
      lane_serial_*     the kernel's own SOURCE on the CPU: the lane-serial build of csrc/gi_core.hpp (tests/emul/libdlscqp_cpu.so:
                        every lane loop 64 trips whatever the agent needs) -- what rounds 2-4 reported as `value`;
      oracle_qp_per_s   the checker (oracle/dlsc_oracle.c: dense model in the reference's row order, generic
                        Goldfarb-Idnani with a fresh QR per QP), one core, for orientation only.
    All are this repository's code: CPLEX, which the reference calls here, is not available on either machine."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import native_lib
    import oracle_lib
    from dlsc_gc_planner_amd import abi
    agents = np.ascontiguousarray(wl["agents"])
    ro = np.ascontiguousarray(wl["row_offsets"], dtype=np.int64)
    rows = np.ascontiguousarray(wl["rows"])
    N = len(agents)
    cores = host_cores()
    nat_one = native_lib.bench(cfg, agents, ro, rows, None, 0, 1, min(2.0, seconds))
    nat_all = native_lib.bench(cfg, agents, ro, rows, None, 0, cores, max(2.0, seconds - 5.0))
    lib_path = os.path.join(ROOT, "tests", "emul", "libdlscqp_cpu.so")
    L = C.CDLL(lib_path)
    vp = C.c_void_p
    L.emul_throughput.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double)]
    res = {}
    for label, nt, secs in (("one", 1, 1.0), ("all", cores, 2.0)):
        n, el = C.c_int64(), C.c_double()
        rc = L.emul_throughput(C.byref(cfg), N, agents.ctypes.data, ro.ctypes.data, rows.ctypes.data, None, 0, nt, secs, C.byref(n), C.byref(el))
        if rc != 0:
            raise RuntimeError("emul_throughput failed (%d)" % rc)
        res[label] = (n.value, el.value)
    probe = min(N, 256)
    t0 = time.perf_counter()
    oracle_lib.plan_batch(cfg, agents[:probe], ro[:probe + 1], rows[:ro[probe]], None, 0, True)
    oracle_rate = probe / (time.perf_counter() - t0)
    return {"value": nat_all[0], "unit": "QP/s", "cores": cores, "kind": "port", "port_kind": "native scalar solver",
            "native_qp_per_s": nat_all[0], "native_one_core_qp_per_s": nat_one[0], "nproc": os.cpu_count(), "cpu_model": cpu_model(),
            "port_note": "tests/native/native_qp.cpp: a scalar C++ program -- loops over the agent's own rows, variables and active rows -- "
                         "with the kernel's mathematics (equalities eliminated and Hessian whitened by the host tables of csrc/qp_tables.cpp, "
                         "Goldfarb-Idnani dual active-set method in range-space form), written independently of the kernel's source and "
                         "checked against the oracle; one agent per thread",
            "sample": "%d agent-QPs (goal LP + QP each) of the timed batch in %.1f s, g++ -O3 -march=x86-64-v3, %d native threads, each "
                      "walking its own range of the batch" % (nat_all[1], nat_all[2], cores),
            "one_core_qp_per_s": nat_one[0],
            "lane_serial_qp_per_s": res["all"][0] / res["all"][1], "lane_serial_one_core_qp_per_s": res["one"][0] / res["one"][1],
            "lane_serial_note": "the kernel's own source run lane by lane on the host (tests/emul/libdlscqp_cpu.so: every lane loop 64 trips, "
                                "every reduction 64 elements): the device algorithm on a CPU, what rounds 2-4 reported as `value`",
            "oracle_qp_per_s_one_core": oracle_rate,
            "oracle_note": "oracle/dlsc_oracle.c (the parity checker: dense, generic) on %d agent-QPs, one core" % probe,
            "ref_published_qp_per_s": 89.4,
            "ref_published_note": "reference log: 11.18 ms/QP mean, CPLEX 22.1.1 Threads=6, 2-D M=10, unknown CPU"}


def cpu_port_step_ms(cfg, inputs, seconds=1.0):
    """The same replanning step on the host: the kernel's own algorithm (lane-serial -O3 build of csrc/gi_core.hpp,
    tests/emul/libdlscqp_cpu.so) on the inputs the GPU step read (read back from HBM), one native thread per granted core
    (never more threads than agents), each walking its own range of the agents.  Returns (ms for all agents once, threads)."""
    import ctypes as C
    import numpy as np
    from dlsc_gc_planner_amd import abi
    agents, ro, rows, sfc, D = inputs
    agents = np.ascontiguousarray(agents)
    ro = np.ascontiguousarray(ro, dtype=np.int64)
    rows = np.ascontiguousarray(rows)
    sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
    N = len(agents)
    nt = max(1, min(host_cores(), N))
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_cpu.so"))
    vp = C.c_void_p
    L.emul_throughput.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double)]
    n, el = C.c_int64(), C.c_double()
    rc = L.emul_throughput(C.byref(cfg), N, agents.ctypes.data, ro.ctypes.data, rows.ctypes.data, None if sfc is None else sfc.ctypes.data,
                           int(D), nt, float(seconds), C.byref(n), C.byref(el))
    if rc != 0:
        raise RuntimeError("emul_throughput failed (%d)" % rc)
    return el.value / n.value * N * 1e3, nt


def closed_loop_latency(device):
    """Wall time of one replanning step of four of the reference's missions flown by the device-resident closed loop
    (dlsc_gc_planner_amd/harness.py), synchronised after every step:
      empty10_1, empty60_1   BASELINE.json configs[0] and [2]: half-spaces + goal LP + QP + state update, three launches per
                             step, also replayed as one HIP graph;
      maze10_dense           the mission of the reference's committed log (2-D, M = 10, corridors, communication range 3):
                             waypoints from the host-side planner of row N4 (the reference's PIBT), four launches per step;
      forest10_spin4_100     BASELINE.json configs[3] (3-D, M = 10, four spinning obstacles with slack variables)."""
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi
    from dlsc_gc_planner_amd.harness import ClosedLoop, SpinObstacles
    g = os.path.join(ROOT, "tests", "golden")
    out = {}
    missions = json.load(open(os.path.join(g, "missions_empty.json")))

    def fly(sim, steps, graph_at=None, skip=5, port=True):
        """Wall time per step (synchronised), then -- for the eager steps -- the stages in the reference's columns (HIP
        events; host time of the waypoint planner) and the same step on the host (the kernel's algorithm, all granted cores,
        the inputs of one mid-flight step read back from HBM)."""
        ts, fails = {"eager": [], "graph": []}, 0
        sim.keep_inputs = True
        sim.enable_stage_timers()
        inputs = None
        for k in range(steps):
            if graph_at is not None and k == graph_at:
                stages = sim.stage_times()
                sim.enable_stage_timers(False)
                sim.enable_graph()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            status, _ = sim.step()
            torch.cuda.synchronize()
            phase = "graph" if graph_at is not None and k >= graph_at else "eager"
            if (k - (graph_at or 0) if phase == "graph" else k) >= skip:
                ts[phase].append((time.perf_counter() - t0) * 1e3)
            fails += int((status != abi.OK).sum())
            if k == min(steps - 1, 30):
                inputs = sim.step_inputs()
        if graph_at is None:
            stages = sim.stage_times()
        extra = {"stage_p50_ms": {k2: float(np.percentile(v[skip:], 50)) for k2, v in stages.items() if len(v) > skip},
                 "stage_names": "the reference's timers (sp_const.hpp:81-122, summary columns multi_sync_simulator.cpp:863-871): "
                                "traj_optimization_time = goal LP + QP in one launch; mapf_time = host waypoint planner with its transfers"}
        try:
            if not port:
                raise RuntimeError("not asked for")
            ms, nt = cpu_port_step_ms(sim.cfg, inputs)
            extra["cpu_port_ms"] = ms
            extra["cpu_port_threads"] = nt
            extra["cpu_port_note"] = "goal LP + QP of one mid-flight step on the host (lane-serial -O3 build of the kernel source), inputs read back from HBM"
            gpu_ms = extra["stage_p50_ms"].get("traj_optimization_time")
            if gpu_ms:
                extra["lp_qp_gpu_ms_over_cpu_port_ms"] = gpu_ms / ms
                extra["verdict"] = ("%d host threads plan this step's %d LP + QPs %.1fx faster than the GPU launch does (one wavefront per "
                                    "agent leaves the chip empty at this size; the launch alone is ~10 us)" % (nt, len(inputs[0]), gpu_ms / ms)
                                    if ms < gpu_ms else "the GPU launch plans this step's LP + QPs %.1fx faster than %d host threads" % (ms / gpu_ms, nt))
        except Exception as e:  # the checker library is optional for the bench line
            extra["cpu_port_ms"] = None
            extra["cpu_port_note"] = "unavailable: %s" % e
        sim.close()
        return ts, fails, extra

    for name in ("empty10_1", "empty60_1"):
        m = missions[name]
        w = m["world"]
        cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                              world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]), device=device)
        sim = ClosedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), radius=m["radius"],
                         downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"],
                         nominal_velocity=m["nominal_velocity"], device=device)
        ts, fails, extra = fly(sim, 110, graph_at=55)
        out[name] = {"agents": len(m["start"]), "lsc_rows_per_agent": (len(m["start"]) - 1) * 27, "steps": len(ts["eager"]),
                     "p50_ms": float(np.percentile(ts["eager"], 50)), "p99_ms": float(np.percentile(ts["eager"], 99)),
                     "p50_ms_as_one_hip_graph": float(np.percentile(ts["graph"], 50)), "qp_failsafes": fails}
        out[name].update(extra)
    # the mission of the reference's log: 2-D, M = 10, corridors, communication range 3, the reference's waypoint planner
    w = json.load(open(os.path.join(g, "world_maze1.json")))
    cfg = abi.make_config(dim=2, M=10, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=3.0,
                          world_use_octomap=True, world_min=tuple(w["world"][:3]), world_max=tuple(w["world"][3:]), world_z_2d=1.0,
                          device=device)
    world = dict(boxes=w["boxes"], resolution=w["world_resolution"], grid_resolution=w["grid_resolution"], pibt="reference")
    sim = ClosedLoop(cfg, np.array(w["start"], np.float32), np.array(w["goal"], np.float32), radius=w["radius"], downwash=2.0,
                     max_vel=w["max_vel"], max_acc=w["max_acc"], nominal_velocity=1.0, waypoint_step=w["grid_resolution"],
                     world=world, device=device)
    def maze_sim():
        return ClosedLoop(cfg, np.array(w["start"], np.float32), np.array(w["goal"], np.float32), radius=w["radius"], downwash=2.0,
                          max_vel=w["max_vel"], max_acc=w["max_acc"], nominal_velocity=1.0, waypoint_step=w["grid_resolution"],
                          world=world, device=device)
    ts, fails, extra = fly(sim, 171)
    ts_g = fly(maze_sim(), 171, graph_at=5, port=False)[0]  # the same flight again, the device part of every step as one graph
    out["maze10_dense_1"] = {"agents": 10, "steps": len(ts["eager"]), "p50_ms": float(np.percentile(ts["eager"], 50)),
                             "p99_ms": float(np.percentile(ts["eager"], 99)), "qp_failsafes": fails,
                             "p50_ms_as_one_hip_graph": float(np.percentile(ts_g["graph"], 50)),
                             "as_one_hip_graph": "the same flight flown again with the device part of the step (neighbour lists, corridors, half-spaces, "
                                                 "LP + QP, state update: five launches) replayed as one graph from step 5 on; the waypoint planner stays "
                                                 "on the host in front of it",
                             "includes": "host-side waypoint planner (PIBT) and the per-step neighbour lists",
                             "reference_log_planning_time_ms_per_agent": 12.17,
                             "reference_log_stage_ms_per_agent": {"lsc_generation_time": 0.145, "sfc_generation_time": 0.542,
                                                                  "traj_optimization_time": 11.18,
                                                                  "note": "log/summary_DLSCGC_10agents.csv: per agent and step, CPLEX 22.1.1, unknown CPU; the "
                                                                          "figures above are per STEP of all ten agents"}}
    out["maze10_dense_1"].update(extra)
    # BASELINE.json configs[3]
    f = json.load(open(os.path.join(g, "world_forest1.json")))
    cfg = abi.make_config(dim=f["dim"], M=f["M"], dt=f["dt"], control_input_weight=0.01, terminal_weight=1.0,
                          slack_collision_weight=f["slack_collision_weight"], communication_range=f["communication_range"],
                          world_use_octomap=True, world_min=tuple(f["world"][:3]), world_max=tuple(f["world"][3:]), device=device)
    spin = SpinObstacles(f["obstacles"])
    world = dict(boxes=f["boxes"], resolution=f["world_resolution"], grid_resolution=f["grid_resolution"], pibt="reference",
                 uncertainty_horizon=f["uncertainty_horizon"])
    sim = ClosedLoop(cfg, np.array(f["start"], np.float32), np.array(f["goal"], np.float32), radius=f["radius"], downwash=f["downwash"],
                     max_vel=f["max_vel"], max_acc=f["max_acc"], nominal_velocity=1.0, waypoint_step=f["grid_resolution"],
                     world=world, dyn=spin.dyn(horizon=f["uncertainty_horizon"], duration=f["obs_duration"]), device=device)
    ts, fails, extra = fly(sim, 80)
    sim_g = ClosedLoop(cfg, np.array(f["start"], np.float32), np.array(f["goal"], np.float32), radius=f["radius"], downwash=f["downwash"],
                       max_vel=f["max_vel"], max_acc=f["max_acc"], nominal_velocity=1.0, waypoint_step=f["grid_resolution"],
                       world=world, dyn=SpinObstacles(f["obstacles"]).dyn(horizon=f["uncertainty_horizon"], duration=f["obs_duration"]), device=device)
    ts_g = fly(sim_g, 80, graph_at=5, port=False)[0]
    out["forest10_spin4_100_1"] = {"agents": len(f["start"]), "dynamic_obstacles": len(f["obstacles"]), "steps": len(ts["eager"]),
                                   "p50_ms": float(np.percentile(ts["eager"], 50)), "p99_ms": float(np.percentile(ts["eager"], 99)),
                                   "qp_failsafes": fails, "p50_ms_as_one_hip_graph": float(np.percentile(ts_g["graph"], 50)),
                                   "note": "220 variables, up to 124 active rows: the working set (96 KB) fits the LDS of a CU since the solver "
                                           "keeps no orthogonal factor (DESIGN.md 6f); includes the host-side waypoint planner"}
    out["forest10_spin4_100_1"].update(extra)
    out["x2048_moving_crowd"] = moving_crowd(device)
    return out


def moving_crowd(device, steps=60, warm=20, heavy_ranks=0):
    """BASELINE.json configs[4] FLOWN: 2 048 agents in one 2-D arena, every agent swapping places with a partner 3.4 m away
    (workloads.swapping_pairs); every step, all in HBM: communication-range neighbour search (dlscqp_neighbours_device),
    corridor boxes, DLSC-GC half-spaces with the generator's presolve, goal LP + QP in the order of the previous step's
    iteration counts, state update.  No host work and no synchronisation inside a step."""
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.harness import ClosedLoop
    cfg = workloads.x2048_config(device=device)
    N = 2048
    start, goal = workloads.swapping_pairs(cfg, N)
    sim = ClosedLoop(cfg, start, goal, radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=0.5, device=device,
                     neighbours="device", nbr_pitch=256, corridor_margin=0.5, prune=True, order_hint=True, heavy_ranks=heavy_ranks)
    sim.keep_inputs = True
    for _ in range(warm):
        sim.step()
    torch.cuda.synchronize()
    # un-synchronised: the steps queue up on the stream like a real loop's
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    sim.enable_stage_timers()
    its, act, fails, rows = [], [], 0, []
    for _ in range(20):
        sim.step()
        r = sim.batch.results()
        its.append(r["iters"][:, 0].copy())
        act.append(r["iters"][:, 3].mean())
        fails += int((r["status"] != abi.OK).sum())
        rows.append(float(sim.row_counts.float().mean()))
    stages = sim.stage_times()
    inputs = sim.step_inputs()
    its = np.stack(its)
    out = {"agents": N, "steps_timed": steps, "ms_per_step": ms, "agent_qp_per_s": N / ms * 1e3,
           "mean_active_set_iterations": float(its.mean()), "max_active_set_iterations": int(its.max()),
           "zero_iteration_fraction": float((its == 0).mean()), "mean_active_rows": float(np.mean(act)),
           "rows_kept_per_agent": float(np.mean(rows)), "neighbours_per_agent": float(sim.nbr.counts.float().mean()),
           "neighbour_list_overflow": int(sim.nbr.overflow.cpu()[0]), "qp_failsafes_in_20_steps": fails,
           "min_separation_ratio": sim.min_separation(),
           "stage_p50_ms": {k: float(np.percentile(v, 50)) for k, v in stages.items()},
           "includes": "neighbour search + corridor boxes + half-space generation (presolve) + goal LP + QP + launch order + state update"}
    # roofline line of the half-space generator (dlscqp_lsc_kernel, with the neighbour search in front of it): it reads one
    # 736-byte record per neighbour (its previous plan and sub-goal) and writes 36 bytes per row that survives the presolve
    nb, kept = float(sim.nbr.counts.float().sum()), float(sim.row_counts.float().sum())
    lsc_ms = float(np.percentile(stages["lsc_generation_time"], 50))
    lsc_bytes = 736.0 * nb + 36.0 * kept
    out["lsc_generation_roofline"] = {"bound": "valu_issue", "algorithmic_bytes_per_step": lsc_bytes, "ms": lsc_ms,
                                      "achieved": lsc_bytes / (lsc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": lsc_bytes / (lsc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "736 B x neighbours read + 36 B x rows written; the kernel is arithmetic (hull search per "
                                              "neighbour and segment), not HBM bound: profiles/r04_other_kernels_pmc.md"}
    try:
        cms, nt = cpu_port_step_ms(cfg, inputs, seconds=2.0)
        out["cpu_port_ms"], out["cpu_port_threads"] = cms, nt
        out["cpu_port_note"] = "goal LP + QP only (no neighbour search, no half-space generation) on the host, inputs read back from HBM"
    except Exception as e:
        out["cpu_port_ms"], out["cpu_port_note"] = None, "unavailable: %s" % e
    # the same step as HIP graphs (three: the phases of the launch-order masks' rotation), neighbour search through cells included
    try:
        sim.enable_stage_timers(False)
        sim.keep_inputs = False
        sim.enable_graph()
        for _ in range(6):
            sim.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sim.step()
        torch.cuda.synchronize()
        out["ms_per_step_as_hip_graphs"] = (time.perf_counter() - t0) / steps * 1e3
    except Exception as e:  # noqa: BLE001
        out["ms_per_step_as_hip_graphs"] = None
        out["hip_graphs_note"] = "unavailable: %s" % e
    sim.close()
    return out


def overlapping_streams(dev, G):
    """G streams whose work runs side by side: streams are mapped to a few hardware queues, and two streams on one queue run one after
    the other (SwarmLoop.pick_streams measures that with the loop's own batches; here with torch's spin kernel: a pair that takes
    as long as the two spins in a row shares a queue)."""
    import torch
    cands = [torch.cuda.Stream(dev) for _ in range(max(8, 2 * G) + 4)]
    if G == 1 or not hasattr(torch.cuda, "_sleep"):
        return cands[:G]

    def span(sa, sb):
        best = float("inf")
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            with torch.cuda.stream(sa):
                torch.cuda._sleep(1000000)
            with torch.cuda.stream(sb):
                torch.cuda._sleep(1000000)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return best
    serial = span(cands[0], cands[0])
    kept, rest = [cands[0]], []
    for s_ in cands[1:]:
        if len(kept) < G and all(span(k_, s_) < 0.75 * serial for k_ in kept):
            kept.append(s_)
        else:
            rest.append(s_)
    return kept + rest[:G - len(kept)] if len(kept) < G else kept


def s64_closed_loop(device, n_swarms=512, groups=1, first=3, steps=8, flights=4, graph=False):
    """BASELINE.json configs[1] at the headline's scale with the WHOLE replanning step on the device (traj_planner.cpp:108-133 per
    agent: prediction of the neighbours from their previous plans, generateDLSCGC :603-666, goal LP, QP; then doStep): the
    `n_swarms` swarms in `groups` closed loops (harness.ClosedLoop), each on its own stream; per step and loop four launches,
    nothing on the host in between -- neighbour lists (the workload's rule: the 5 nearest agents of the swarm,
    dlscqp_neighbours_knn_device), half-spaces (dlscqp_lsc_kernel with the presolve: 4 x 27 + 12 rows per agent at most), goal
    LP + QP in the kernel's own launch order (iteration counts of the two previous steps), state update (dlscqp_advance_device).
    Timed: replanning steps [first, first + steps) of `flights` fresh flights, queued without synchronising -- the window in
    which the swarms cross (the batches the headline rotates through are replanning steps 3..6 of the same swarms; after ~40
    steps everybody has arrived and a step is trivial).  Statistics and stage times: a further flight of one loop, same window,
    synchronised per step."""
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.harness import ClosedLoop
    cfg = workloads.s64_config(device=device)
    start, goal = workloads.swarm_missions(cfg, n_swarms=n_swarms, swarm_size=64, seed=2000)
    G = groups
    while n_swarms % G:
        G -= 1
    per = (n_swarms // G) * 64
    dev = torch.device("cuda", device)
    streams = overlapping_streams(dev, G)

    def take_off():
        sims = []
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                sims.append(ClosedLoop(cfg, start[g * per:(g + 1) * per], goal[g * per:(g + 1) * per], radius=0.15, downwash=2.0, max_vel=1.0,
                                       max_acc=2.0, waypoint_step=5.0, device=device, neighbours="knn", knn=(64, 5), tail=(4, 2), prune=True,
                                       order_hint=True))
        return sims

    def advance(sims, k):
        t0 = time.perf_counter()
        for _ in range(k):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    sims[g].step()
        host = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, host
    el = host = 0.0
    for _ in range(flights):
        sims = take_off()
        advance(sims, first)
        if graph:  # the steady-state step of every loop as HIP graphs (three per loop: the phases of the launch-order masks' rotation)
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    sims[g].enable_graph()
            torch.cuda.synchronize(dev)
        e_, h_ = advance(sims, steps)
        el, host = el + e_, host + h_
        for s_ in sims:
            s_.close()
    ms = el / (flights * steps) * 1e3
    # statistics and stage times of the same window: one more flight of the first loop, synchronised per step
    sim = take_off()[0] if G == 1 else ClosedLoop(cfg, start[:per], goal[:per], radius=0.15, downwash=2.0, max_vel=1.0, max_acc=2.0, waypoint_step=5.0,
                                                   device=device, neighbours="knn", knn=(64, 5), tail=(4, 2), prune=True, order_hint=True)
    for _ in range(first):
        sim.step()
    its, rows, fails, lp_fails = [], [], 0, 0
    sim.enable_stage_timers()
    for _ in range(steps):
        sim.step()
        r = sim.batch.results()
        its.append(r["iters"][:, 0].copy())
        fails += int((r["status"] != abi.OK).sum())
        lp_fails += int((r["goal_status"] == abi.INFEASIBLE).sum())
        rows.append(float(sim.row_counts.float().mean()))
    stages = sim.stage_times()
    its = np.stack(its)
    N = n_swarms * 64
    plan_ms = float(np.percentile(stages["traj_optimization_time"], 50))
    lsc_ms = float(np.percentile(stages["lsc_generation_time"], 50))
    nb = float(sim.nbr.counts.float().sum())
    kept = float(sim.row_counts.float().sum())
    rows_mean = float(np.mean(rows))
    bytes_qp = 104 + 36 * rows_mean + 360 + 24
    out = {"agents": N, "swarms": n_swarms, "loops": G, "agents_per_loop": per, "replanning_steps_timed": [first, first + steps], "flights": flights,
           "ms_per_step": ms, "agent_qp_per_s": N / ms * 1e3, "host_ms_to_queue_a_step": host / (flights * steps) * 1e3,
           "launches_per_step_and_loop": 4,
           "mean_active_set_iterations": float(its.mean()), "max_active_set_iterations": int(its.max()),
           "zero_iteration_fraction": float((its == 0).mean()), "rows_kept_per_agent": rows_mean, "rows_generated_per_agent": 120,
           "qp_failsafe_fraction": fails / float(steps * per), "goal_lp_infeasible_fraction": lp_fails / float(steps * per),
           "stage_p50_ms_one_loop_alone": {k: float(np.percentile(v, 50)) for k, v in stages.items()},
           "stage_names": "the reference's timers (sp_const.hpp:81-122): lsc_generation_time = neighbour lists + half-spaces, "
                          "traj_optimization_time = goal LP + QP (one launch), state_update_time = doStep; one loop of %d agents "
                          "running alone, events around the stages" % per,
           "includes": "neighbour lists + half-space generation (presolve) + goal LP + QP + state update, all in HBM",
           # the dominant kernel of THIS step is still the plan kernel; the generator reads one 736-byte record per neighbour and
           # writes 36 bytes per row it keeps
           "roofline_plan_kernel": {"bound": "valu_issue", "algorithmic_bytes_per_qp": bytes_qp, "ms": plan_ms,
                                    "achieved": per * bytes_qp / (plan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": per * bytes_qp / (plan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "roofline_half_space_generator": {"bound": "valu_issue", "algorithmic_bytes_per_step": 736.0 * nb + 36.0 * kept, "ms": lsc_ms,
                                             "achieved": (736.0 * nb + 36.0 * kept) / (lsc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": (736.0 * nb + 36.0 * kept) / (lsc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "note": "736 B x neighbours read + 36 B x rows written (one loop alone; includes the neighbour search)"}}
    sim.close()
    return out


class StressStep:
    """BASELINE.json configs[4] (2 048 agents, 2-D, M = 10, ~2 500 DLSC-GC rows per agent generated on the device) as a
    STRONG-scaling step: the agents are sharded by contiguous id range; every rank generates the half-spaces of its own
    agents from the table of everybody's previous plans, plans them, and one all-gather rebuilds the table.  Interior
    agents (all neighbours on this rank) are planned before the exchange of the previous step is awaited, halo agents
    after it.  The batch is stationary (the generators read the workload's plan table every step);
    the payload that travels and the copies into the gathered table are the real ones."""

    def __init__(self, device_index, rank, world, dist, prune=True, order_hint=True):
        import numpy as np
        import torch
        from dlsc_gc_planner_amd import abi, sharding, workloads
        from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self.order_hint = order_hint
        cfg = workloads.x2048_config(device=device_index)
        self.cfg = cfg
        w = workloads.x2048(cfg)
        N = len(w["agents"])
        self.N = N
        lo, hi = sharding.shard_range(N, rank, world)
        self.lo, self.hi = lo, hi
        off, ids = w["nbr_offsets"].astype(np.int64), w["nbr_ids"]
        loc_off, loc_ids = (off[lo:hi + 1] - off[lo]), ids[off[lo]:off[hi]]
        halo = sharding.halo_agents(loc_off, loc_ids, lo, hi) if world > 1 else []
        hs = set(halo)
        interior = [a for a in range(hi - lo) if a not in hs]
        self.n_interior, self.n_halo = len(interior), len(halo)
        self.opt = BatchTrajOptimizer(cfg)
        dev = torch.device("cuda", device_index)
        self.dev = dev
        self.plans = torch.from_numpy(w["plans"]).to(dev)
        self.goals = torch.from_numpy(w["goals"]).to(dev)
        self.parts = []
        for sel in (interior, halo):
            if not sel:
                self.parts.append(None)
                continue
            sel = np.asarray(sel)
            cnt = (loc_off[sel + 1] - loc_off[sel])
            o = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            nb = np.concatenate([loc_ids[loc_off[a]:loc_off[a + 1]] for a in sel]).astype(np.int32)
            gen = self.opt.make_lsc_generator(N, o, nb, w["radius"], w["downwash"], local_ids=(lo + sel).astype(np.int32), device=device_index)
            db = self.opt.make_device_batch(w["agents"][lo + sel], gen.row_offsets, np.zeros(0, abi.ROW_DTYPE), w["sfc"][lo + sel], 0,
                                            want_iters=True, device=device_index)
            for k in ("nx", "ny", "nz", "px", "py", "pz", "d", "key"):
                setattr(db.c, k, gen.soa[k].data_ptr())
            ro32 = torch.from_numpy(gen.row_offsets.astype(np.int32)).to(dev)
            db.c.row_offsets = ro32.data_ptr()
            db.c.max_rows_per_agent = int(np.diff(gen.row_offsets).max())
            if prune:  # the generator's presolve: rows implied by the agent's corridor boxes are never written
                db.use_row_counts(gen.enable_pruning(db.t["sfc"].view(torch.float32).view(len(sel), cfg.M, 6)))
            self.parts.append((gen, db, ro32, torch.from_numpy(sel).to(dev)))
        self.rows_per_agent = float(np.diff(off).mean() * (cfg.M * (cfg.n + 1) - cfg.phi))
        self.prune = prune
        n_loc = hi - lo
        self.n_loc = n_loc
        E = cfg.M * (cfg.n + 1) * 3
        self.own_ctrl = torch.zeros((n_loc, E), dtype=torch.float32, device=dev)
        self.own_goal = torch.zeros((n_loc, 3), dtype=torch.float32, device=dev)
        self.own_status = torch.zeros((n_loc, 1), dtype=torch.int32, device=dev)
        self.table = torch.zeros((N, E + 4), dtype=torch.float32, device=dev)  # the gathered plans as the next step would read them
        self.xchg = None
        if dist is not None:
            pad = max(sharding.shard_sizes(N, world))
            assert pad == n_loc, "2 048 agents divide evenly over 1, 2, 4, 8 ranks"
            self.xchg = sharding.make_exchange(self.opt, dist, rank, world, n_loc, dev)

    def _plan(self, part):
        gen, db, _, sel = part
        gen.generate(self.goals, plans=self.plans)
        if self.order_hint:
            db.rotate_hint_masks()  # launch order from the class masks of the previous step's launch, found by the plan kernel itself
        db.launch(True, True)
        self.own_ctrl.index_copy_(0, sel, db.out_ctrl.view(len(sel), -1))
        self.own_goal.index_copy_(0, sel, db.out_goal)
        self.own_status.index_copy_(0, sel, db.out_status.view(-1, 1))

    def step(self, pipelined):
        inner, halo = self.parts
        if self.xchg is None:
            self._plan(inner)
            return
        if pipelined:
            if inner is not None:
                self._plan(inner)          # runs while the exchange of the previous step is still in flight
            if self.xchg.last is not None:
                self.table.copy_(self.xchg.table())  # wait for it; the halo agents read the complete table
            if halo is not None:
                self._plan(halo)
            self.xchg.post(self.own_ctrl, self.own_goal, self.own_status)
        else:
            if inner is not None:
                self._plan(inner)
            if halo is not None:
                self._plan(halo)
            self.xchg.post(self.own_ctrl, self.own_goal, self.own_status)
            self.table.copy_(self.xchg.table())

    def time(self, steps, pipelined):
        torch, dist = self.torch, self.dist
        for _ in range(3):
            self.step(pipelined)
        torch.cuda.synchronize(self.dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(pipelined)
        torch.cuda.synchronize(self.dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(self.dev)
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el / steps * 1e3

    def rows_kept_per_agent(self):
        import numpy as np
        c = [p[0].row_counts.cpu().numpy() for p in self.parts if p is not None and getattr(p[0], "row_counts", None) is not None]
        return float(np.concatenate(c).mean()) if c else self.rows_per_agent

    def solved_fraction(self):
        import numpy as np
        from dlsc_gc_planner_amd import abi
        st = np.concatenate([p[1].results()["status"] for p in self.parts if p is not None])
        return float((st == abi.OK).mean())

    def close(self):
        self.opt.close()


def over_ranks(dist, dev, values):
    """{name: {"min", "max", "mean"} over the ranks} of one float per rank and name (one small all-gather)."""
    import torch
    names = sorted(values)
    t = torch.tensor([float(values[k]) for k in names], dtype=torch.float64, device=dev)
    if dist is not None:
        world = dist.get_world_size()
        out = torch.empty(world * len(names), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, t)
        m = out.view(world, len(names)).cpu().numpy()
    else:
        m = t.cpu().numpy()[None, :]
    return {k: {"min": float(m[:, i].min()), "max": float(m[:, i].max()), "mean": float(m[:, i].mean())} for i, k in enumerate(names)}


def exchange_breakdown(xchg, skip=0):
    """Mean ms per exchange of a PlanExchange that was timed (sharding.PlanExchange.enable_timing)."""
    import numpy as np
    t = xchg.times()
    return {"pack_ms": float(np.mean(t["pack"][skip:])) if len(t.get("pack", [])) > skip else float("nan"),
            "all_gather_ms": float(np.mean(t["all_gather"][skip:])) if len(t.get("all_gather", [])) > skip else float("nan"),
            "output_ready_to_gathered_ms": float(np.mean(t["lag"][skip:])) if len(t.get("lag", [])) > skip else float("nan"),
            "wait_for_table_ms": float(np.mean(t["wait"][skip:])) if len(t.get("wait", [])) > skip else 0.0}


def sharded_mission(dist, rank, world, device, steps=50, warm=10):
    """BASELINE.json configs[2]: the empty60 mission flown in closed loop with its 60 agents sharded over the ranks and the
    RCCL all-gather of plans, sub-goals and verdicts on the critical path of every step (every agent neighbours every other
    one: harness.ShardedLoop).  ms per step = max over the ranks of (wall time of `steps` steps) / steps, steps queued
    without synchronising; the breakdown is per rank (HIP events on the compute and the communication stream)."""
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi
    from dlsc_gc_planner_amd.harness import ShardedLoop
    m = json.load(open(os.path.join(ROOT, "tests", "golden", "missions_empty.json")))["empty60_1"]
    w = m["world"]
    cfg = abi.make_config(dim=3, M=5, dt=0.2, control_input_weight=0.01, terminal_weight=1.0, communication_range=-1.0,
                          world_use_octomap=False, world_min=tuple(w[:3]), world_max=tuple(w[3:]), device=device)
    dev = torch.device("cuda", device)
    sim = ShardedLoop(cfg, np.array(m["start"], np.float32), np.array(m["goal"], np.float32), dist, rank, world, radius=m["radius"],
                      downwash=m["downwash"], max_vel=m["max_vel"], max_acc=m["max_acc"], nominal_velocity=m["nominal_velocity"], device=device)
    for _ in range(warm):
        sim.step()
    torch.cuda.synchronize(dev)
    sim.xchg.enable_timing()
    sim.opt.set_timing(steps)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    tt = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    kms = sim.opt.kernel_times_ms(steps)
    per_rank = exchange_breakdown(sim.xchg)
    per_rank["plan_kernel_ms"] = float(np.mean(kms)) if len(kms) else float("nan")
    st = sim.batch.results()["status"]
    fails = torch.tensor([int((st != abi.OK).sum())], dtype=torch.int64, device=dev)
    dist.all_reduce(fails)
    moved = float((sim.plans_all[:, 0, 0] - torch.from_numpy(np.array(m["start"], np.float32)).to(dev)).norm(dim=1).mean())
    out = {"mission": "empty60_1", "agents": 60, "agents_this_rank": sim.hi - sim.lo, "ranks": world, "steps_timed": steps,
           "ms_per_step": float(tt.item()) / steps * 1e3, "qp_failsafes_last_step_all_ranks": int(fails.item()),
           "mean_distance_flown_m": moved,
           "per_rank_ms": over_ranks(dist, dev, per_rank),
           "per_rank_note": "plan_kernel_ms: HIP events around the LP + QP launch; pack / all_gather / output_ready_to_gathered: events "
                            "on the compute and communication streams of sharding.PlanExchange; wait_for_table: how long the compute "
                            "stream stood still for the gathered table; min / max / mean over the ranks",
           "exchange": "one all-gather of [agents, 94] float32 (plans, sub-goal, verdict) per step, on the critical path: every agent "
                       "reads every other agent's previous plan"}
    sim.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST HOOK (tests/test_gpu_bench.py): DLSCQP_BENCH_GLOO_ONE_DEVICE=1 runs the ranks of an N > 1 launch on ONE device over gloo
    # (which moves device tensors through the host) -- the multi-rank logic of this file (shards, halo agents, the exchange, the
    # checks of the gathered tables) on a one-GPU box; RCCL refuses two ranks on one device.  Not a measurement.
    one_device = os.environ.get("DLSCQP_BENCH_GLOO_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # streams are mapped to a few hardware queues (4 by default) and two streams on one queue run one after the other: the swarm
    # groups' streams and the exchange's communication stream get a queue each (set before HIP starts; an explicit setting wins)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trajectory optimizer has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    # stdout carries ONE line, the JSON: libraries that write to file descriptor 1 themselves (RCCL prints its version
    # there) are sent to stderr for the life of the process
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL's kernels on a high-priority stream: with the pipelined exchange they share the GPU with the plan kernel of
        # the next step and should get the first free workgroup slots
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=opts)
        world = dist.get_world_size()  # n_gpus in the line = the ranks RCCL really has
        if world != args.gpus and not args.force_dist:
            raise SystemExit("bench.py --gpus %d, but the process group has %d ranks: the line would report the wrong N" % (args.gpus, world))
        if torch.cuda.device_count() < (1 if args.force_dist or one_device else args.gpus):
            raise SystemExit("bench.py --gpus %d, but only %d devices are visible" % (args.gpus, torch.cuda.device_count()))

    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    import __graft_entry__ as graft_entry
    build_state = graft_entry.build_state() if not os.environ.get("DLSCQP_LIB") else "DLSCQP_LIB=" + os.environ["DLSCQP_LIB"]

    cfg = workloads.s64_config(device=local_rank)
    opt = BatchTrajOptimizer(cfg)

    def solver(agents, ro, rows, sfc):
        return opt.plan(agents, ro, rows, sfc, 0, True, True)

    B = max(1, args.batches)
    t_gen = time.perf_counter()
    wl = workloads.generate(cfg, solver, n_swarms=args.swarms, swarm_size=64, seed=args.seed + 1000 * rank, keep=B)
    t_gen = time.perf_counter() - t_gen
    hist = wl["history"]
    N, K = len(wl["agents"]), wl["K"]

    class SwarmLoop:
        """The timed loop.  The swarms of this rank in `G` groups; every group is its own sequence of replanning steps on its own
        stream -- launch, launch order of the group's next step from its own iteration counts, exchange of its plans -- and
        no group waits for another: swarms are independent (an agent's neighbours are the other 63 agents of its swarm), so
        nothing requires them to replan in lockstep, and the tail of one group's launch (its few 30-iteration agents) runs
        beside the other groups' launches instead of holding an empty GPU.  G = 1 is one launch of all agents per step."""

        def __init__(self, G, n_use=None):
            n_use = N if n_use is None else n_use  # the first n_use agents of the workload (whole swarms)
            assert n_use % 64 == 0 and (n_use // 64) % G == 0, "whole swarms per group"
            self.G, self.per, self.n_use = G, n_use // G, n_use
            per = self.per
            self.opts = [opt] + [BatchTrajOptimizer(cfg) for _ in range(G - 1)]  # a context per group: launch-order scratch, event pairs
            self.dbs = [[self.opts[g].make_device_batch(h["agents"][g * per:(g + 1) * per], np.arange(per + 1) * K,
                                                         h["rows"][g * per * K:(g + 1) * per * K], None, 0, want_iters=True) for h in hist]
                        for g in range(G)]
            for g in range(G):
                for db in self.dbs[g]:
                    db.use_row_pitch()  # every agent has K rows: the kernel computes where they start
            self.streams = self.pick_streams(G)
            self.counter = 0
            self.device_order = False
            self.xchg = None
            self.cargs = None
            self.no_exchange = False  # time the loop with the exchange restricted to what a consumer needs (S64: nothing)
            self.older = [[[self.dbs[g][(j - k) % B].out_iters for k in (1, 2) if (j - k) % B != (j + 1) % B] for j in range(B)] for g in range(G)]
            self.planned = [[torch.cuda.Event() for _ in range(B)] for _ in range(G)]
            for i in range(max(args.warmup, B)):
                for g in range(G):
                    self.dbs[g][i % B].launch(True, True)
            torch.cuda.synchronize(dev)
            # launch order of batch j: most expensive first by the iteration counts of batch j-1, i.e. of the same agents one
            # replanning step earlier (for the first batch: of the last one) -- never by the batch's own counts
            self.iters = [np.concatenate([self.dbs[g][j].out_iters.cpu().numpy()[:, 0] for g in range(G)]) for j in range(B)]
            self.hint = [[np.argsort(-self.iters[(j - 1) % B][g * per:(g + 1) * per], kind="stable").astype(np.int32) for j in range(B)]
                         for g in range(G)]

        def pick_streams(self, G):
            """G streams whose work really runs side by side.  Streams are mapped to a few hardware queues, and two streams on one
            queue run one after the other (measured: the first two streams a process creates shared a queue -- two groups on them
            took 0.49 ms per step, on any other pair 0.28): of eight new streams those are kept that overlap with every stream kept
            before, judged by launching two of the loop's own batches on a pair and on one stream."""
            cands = [torch.cuda.Stream(dev) for _ in range(max(8, 2 * G) + 4)]
            self.spare, self.span = cands[1:], None
            if G == 1:
                return cands[:1]
            a, b = self.dbs[0][0], self.dbs[1][0]

            spin = hasattr(torch.cuda, "_sleep")  # torch's spin kernel: one workgroup that waits -- two of them on one hardware queue take
            # twice as long as on two, whatever else the chip does (launches of the loop's own batches, round 2-4's probe, fill the
            # chip each: a pair side by side is hardly shorter than the two in a row, and the test hung on their tails)

            def span(sa, sb):
                best = float("inf")
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    if spin:
                        with torch.cuda.stream(sa):
                            torch.cuda._sleep(1000000)
                        with torch.cuda.stream(sb):
                            torch.cuda._sleep(1000000)
                    else:
                        for _ in range(4):
                            a.launch(True, True, stream=sa)
                            b.launch(True, True, stream=sb)
                    torch.cuda.synchronize(dev)
                    best = min(best, time.perf_counter() - t0)
                return best
            serial = span(cands[0], cands[0])
            self.span = lambda sa, sb: span(sa, sb) < (0.75 if spin else 0.8) * serial  # do sa and sb run side by side?
            kept = [cands[0]]
            rest = []
            for s_ in cands[1:]:
                if len(kept) < G and all(self.span(k_, s_) for k_ in kept):
                    kept.append(s_)
                else:
                    rest.append(s_)
            self.spare = rest
            self.stream_note = "%d streams that overlap pairwise, picked from %d new ones" % (len(kept), len(cands))
            return kept + rest[:G - len(kept)] if len(kept) < G else kept

        def spare_stream(self):
            """a further stream that overlaps with every group's stream (the exchange's streams), if there is one"""
            for i, s_ in enumerate(self.spare):
                if self.span is None or all(self.span(k_, s_) for k_ in self.streams):
                    return self.spare.pop(i)
            return torch.cuda.Stream(dev)

        def make_exchange(self, force_torch=False):
            # per agent: control points, sub-goal, status (376 B at S64) -- the broadcast payload of the reference.  ONE exchange
            # per step for all groups: the groups' batches of a step write slices of one output table, a stream of its own
            # (`join`) waits for the groups' launches of the step and posts the table; no group's stream waits for another
            # group (four exchanges per step, one per group, were measured first: the step became bound by the host's calls)
            from dlsc_gc_planner_amd import sharding
            shape1 = tuple(self.dbs[0][0].out_ctrl.shape[1:])
            # the library's own exchange (ncclAllGather called from C on a stream it owns) wherever RCCL can run; torch.distributed
            # only for the two-ranks-on-one-device test hook (gloo)
            if force_torch:  # the fallback: torch.distributed's own all-gather (verify_exchange found the library's exchange wrong)
                if self.xchg is not None:
                    try:
                        self.xchg.close()
                    except Exception:  # noqa: BLE001
                        pass
                E0 = int(np.prod(shape1))
                self.xchg = sharding.PlanExchange(dist, world, N, (E0, 3, 1), dev)
                self.xchg.issued_by = "torch.distributed"
                self.cargs = None
            else:
                self.xchg = sharding.make_exchange(opt, dist, rank, world, N, dev)
            self.native = isinstance(self.xchg, sharding.NativeExchange)
            if not self.native:
                self.xchg.comm = self.spare_stream()  # (not one that shares a hardware queue with a group's launches)
            self.join = self.spare_stream()
            self.table_ready = None  # serial mode: the gathered table of the previous step (an event on `join`)
            per = self.per
            # ... as RECORDS (control points | sub-goal | verdict, 94 words at S64): the plan kernel writes the layout the exchange
            # sends (dlscqp_device_batch.out_record_stride), no packing pass between a launch and the collective
            E_ = int(np.prod(shape1))
            self.rec = [torch.zeros((N, E_ + 4), dtype=torch.float32, device=dev) for _ in range(B)]
            for r_ in self.rec:
                r_[:, E_ + 3:].view(torch.int32).fill_(-1)
            self.out = [(r_[:, :E_], r_[:, E_:E_ + 3], r_[:, E_ + 3:].view(torch.int32)[:, 0]) for r_ in self.rec]
            for g in range(self.G):
                for j in range(B):
                    self.dbs[g][j].use_records(self.rec[j][g * per:(g + 1) * per])

        def verify_exchange(self):
            """The gathered table of the LAST posted step, WHOLE, against torch.distributed's own all-gather of the same records (every
            rank's block, not only this rank's own: the library's exchange -- ncclAllGather bound by hand -- has never run with more
            than one rank on the boxes this repository is built on).  All ranks agree on the answer.  Returns (ok, reason)."""
            j = (self.counter - 1) % B
            ok, why = 1, ""
            try:
                table = self.xchg.table()
                torch.cuda.synchronize(dev)
                got = table.reshape(world * N, -1).clone()
                if os.environ.get("DLSCQP_BENCH_CORRUPT_EXCHANGE") == "1" and getattr(self, "native", False):
                    got[(world * N) // 2, 3] += 1.0  # TEST HOOK (tests/test_gpu_bench.py): a table that differs in one word
                ref = torch.empty_like(got)
                dist.all_gather_into_tensor(ref, self.rec[j].reshape(N, -1).contiguous())
                torch.cuda.synchronize(dev)
                if not torch.equal(ref.view(torch.int32), got.view(torch.int32)):
                    bad = int((ref.view(torch.int32) != got.view(torch.int32)).any(dim=1).sum())
                    ok, why = 0, "%d of %d records of the gathered table differ from torch.distributed's all-gather of the same records" % (bad, world * N)
            except Exception as e:  # noqa: BLE001
                ok, why = 0, "exchange check failed: %s" % e
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and ok:
                why = "another rank's gathered table is wrong"
            return bool(int(flag.item())), why

        def step_args(self):
            """One dlscqp_step_args per batch of the rotation: the whole step -- every group's launch on its stream and, with an
            exchange, the events that tie the step's output table to the post -- is then ONE call into the library."""
            import ctypes as C
            self.cargs = []
            for j in range(B):
                groups = (abi.StepGroup * self.G)()
                for g in range(self.G):
                    db = self.dbs[g][j]
                    db.c.do_goal, db.c.do_traj, db.c.launch_offset, db.c.launch_count = 1, 1, 0, 0
                    groups[g].ctx = self.opts[g].handle
                    groups[g].batch = C.pointer(db.c)
                    groups[g].stream = self.streams[g].cuda_stream
                a = abi.StepArgs()
                a.n_groups, a.groups, a.exchange_mode = self.G, groups, abi.EXCHANGE_NONE
                if self.xchg is not None and getattr(self, "native", False):
                    a.exchange = self.xchg.handle
                    a.records = self.rec[j].data_ptr()
                    a.rotation = B if 3 <= B <= 8 and not args.lockstep_exchange else 0
                self.cargs.append((a, groups))

        def step(self, pipelined):
            j = self.counter % B
            self.counter += 1
            if self.cargs is not None and not self.device_order and (self.xchg is None or self.native):
                a = self.cargs[j][0]
                a.exchange_mode = abi.EXCHANGE_NONE
                if self.xchg is not None and not self.no_exchange:
                    a.exchange_mode = abi.EXCHANGE_PIPELINED if pipelined else abi.EXCHANGE_SERIAL
                    self.xchg.last = self.xchg.k & 1  # (the post happens inside the call)
                    self.xchg.k += 1
                rc = opt.lib.dlscqp_step_device(a)
                if rc != abi.SUCCESS:
                    raise RuntimeError("dlscqp_step_device failed (%d): %s" % (rc, opt.lib.dlscqp_last_error().decode()))
                return
            for g in range(self.G):
                st = self.streams[g]  # (explicit streams and events made once: a stream context or an event object per call costs
                db, nxt = self.dbs[g][j], self.dbs[g][(j + 1) % B]  # the host ~10 us, and this loop makes dozens of calls per step)
                if use_dist and pipelined:
                    # the exchange of step k (packing included) runs on the communication stream while step k+1 writes another
                    # batch's outputs (B >= 3 batches rotate: the table a launch writes was posted B steps ago)
                    self.xchg.before_overwrite(st, tables=B if 3 <= B <= 8 and not args.lockstep_exchange else None)
                elif use_dist and self.table_ready is not None:
                    st.wait_event(self.table_ready)  # serial mode: the next launch waits for the exchange
                db.launch(True, True, stream=st)
                if use_dist:
                    self.planned[g][j].record(st)
                if self.device_order:
                    # the next batch's order from this batch's counts and the two batches' before it -- never from its own
                    nxt.update_order(stream=st, iters=db.out_iters, history=self.older[g][j])
            if use_dist:
                for g in range(self.G):
                    self.join.wait_event(self.planned[g][j])
                if self.native:
                    self.xchg.post_records(self.rec[j], stream=self.join)
                else:
                    self.xchg.post(*self.out[j], pack_on_comm=pipelined, stream=self.join)
                if not pipelined:
                    self.xchg.table(self.join)
                    if self.table_ready is None:
                        self.table_ready = torch.cuda.Event()
                    self.table_ready.record(self.join)

        def sync(self):
            torch.cuda.synchronize(dev)

        def timed(self, steps, pipelined, orders, record_kernel_times=False, repeats=1):
            """repeats > 1: the `steps`-long loop `repeats` times BACK TO BACK inside one bracket of synchronisations; every group's
            stream records an event at each boundary, and self.repeat_ms = the repeats' durations (a boundary = the moment the last
            group's stream passed it).  A bracket of 20 steps is 4.5 ms, of which a fixed ~0.2 ms are the bracket itself (first
            launches, the slowest group's tail, the last synchronisation): the median of back-to-back repeats does not carry it.
            orders: "kernel" = the plan kernel finds its launch order itself from the iteration counts of the three previous
            replanning steps (dlscqp_device_batch.hint_iters: the other batches of the rotation, never the batch's own -- no stream
            operation between two launches; round 5: from the class masks the previous step's launch left behind,
            dlscqp_device_batch.hint_mask_*, "iters" = round 4's form, a load of the 64 keys of the block in front of every agent);
            "device" = the same key sorted by dlscqp_launch_order_history_device after every
            step (round 3's method: two more kernels per group and step); "host" = permutations prepared before the loop;
            None = id order."""
            self.device_order = isinstance(orders, str) and orders == "device"
            kernel_order = isinstance(orders, str) and orders == "kernel"   # class masks (round 5)
            iters_order = isinstance(orders, str) and orders == "iters"     # the keys themselves (round 4's hint_iters)
            if kernel_order and not hasattr(self, "masks"):
                self.masks = [[db.new_hint_masks() for db in self.dbs[g]] for g in range(self.G)]
            for g in range(self.G):
                for j, db in enumerate(self.dbs[g]):
                    db.set_order(None if orders is None or self.device_order or kernel_order or iters_order else self.hint[g][j])
                    db.set_hint([self.dbs[g][(j - k) % B].out_iters for k in (1, 2, 3) if (j - k) % B != j] if iters_order else None)
                    if kernel_order and B >= 3:
                        # batch j is ordered by the masks batch j - 1 (the same agents one step earlier) left behind, writes its own,
                        # and clears those of batch j + 1 (last read a rotation ago)
                        self.masks[g][j].zero_()
                        db.set_hint_masks(self.masks[g][(j - 1) % B], self.masks[g][j], self.masks[g][(j + 1) % B])
                    else:
                        db.set_hint_masks(None, None, None)
            self.step_args()
            self.counter = 0
            for _ in range(B):
                self.step(pipelined)
            self.sync()
            self.counter = 0
            for o in self.opts:
                o.set_timing(steps if record_kernel_times else 0)
            if use_dist:
                dist.barrier()
            self.sync()
            marks = None
            if repeats > 1:
                marks = [[torch.cuda.Event(enable_timing=True) for _ in self.streams] for _ in range(repeats + 1)]
            t0 = time.perf_counter()
            for rep_ in range(repeats):
                if marks is not None:
                    for s_, e_ in zip(self.streams, marks[rep_]):
                        e_.record(s_)
                for _ in range(steps):
                    self.step(pipelined)
            if marks is not None:
                for s_, e_ in zip(self.streams, marks[repeats]):
                    e_.record(s_)
            self.host_ms_per_step = (time.perf_counter() - t0) / (steps * repeats) * 1e3  # what the host needs to queue a step
            self.sync()
            self.repeat_ms = None
            if marks is not None:
                # boundary r = when the LAST group's stream passed it, on the clock of the first boundary's first event
                at = [max(marks[0][0].elapsed_time(e_) for e_ in row) for row in marks]
                self.repeat_ms = [at[r_ + 1] - at[r_] for r_ in range(repeats)]
            if use_dist:
                dist.barrier()
            self.sync()
            el = time.perf_counter() - t0
            if use_dist:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            return el

        def kernel_times_ms(self, steps):
            return np.concatenate([np.asarray(o.kernel_times_ms(steps), dtype=np.float64) for o in self.opts])

        def results(self):
            """per batch of the rotation: the groups' outputs, concatenated in agent order"""
            per_group = [[db.results() for db in self.dbs[g]] for g in range(self.G)]
            return [{k: np.concatenate([per_group[g][j][k] for g in range(self.G)]) for k in per_group[0][j]} for j in range(B)]

        def close(self):
            self.dbs = None
            for o in self.opts[1:]:
                o.close()

    G = args.groups if args.groups > 0 else (2 if use_dist else 4)
    while (N // 64) % G:
        G -= 1
    loop = SwarmLoop(G)
    iters = loop.iters
    with np.errstate(invalid="ignore", divide="ignore"):
        hint_corr = float(np.nan_to_num(np.mean([np.corrcoef(iters[(j - 1) % B], iters[j])[0, 1] for j in range(B)]))) if B > 1 else 1.0

    pipelined_ok = False
    if use_dist:
        from dlsc_gc_planner_amd import sharding
        # Which agents of step k+1 need the gathered table of step k?  Those with a neighbour on another rank.  Every
        # rank owns whole swarms here (an agent's neighbours are the other 63 agents of its swarm), so there are none:
        # the exchange of step k may run while step k+1 is planned.
        lo = rank * N
        members = lo + (np.arange(N) // 64 * 64)[:, None] + np.arange(64)[None, :]  # [N, 64] ids of the agent's swarm
        nbr_ids = members[members != (lo + np.arange(N))[:, None]]                 # ... without the agent itself
        halo = sharding.halo_agents(np.arange(N + 1) * 63, nbr_ids, lo, lo + N)
        pipelined_ok = len(halo) == 0
        loop.make_exchange()

    head_pipelined = use_dist and pipelined_ok and args.exchange == "pipelined"
    if use_dist:
        assert B >= 3, "the pipelined exchange reads a step's output table while the next steps write other tables"
    # N > 1: the exchange is CHECKED around the timed loop -- the whole gathered table against torch.distributed's own all-gather of the
    # same records, once before and once after.  If the library's exchange fails either check the run falls back to torch.distributed
    # (config.exchange_issued_by says so) and the headline is timed with that: a multi-GPU line must not rest on a path that only
    # ever ran with one rank.
    exchange_check = None
    if use_dist:
        exchange_check = {"before": None, "after": None, "fallback": None}
        loop.timed(B, head_pipelined, "kernel")
        okx, whyx = loop.verify_exchange()
        exchange_check["before"] = "ok" if okx else whyx
        if not okx:
            if not loop.native:
                raise SystemExit("bench.py: torch.distributed's exchange does not gather the records it was given: %s" % whyx)
            print("bench.py: the library's own exchange failed its check before the timed loop (%s): torch.distributed instead" % whyx, file=sys.stderr)
            exchange_check["fallback"] = whyx
            loop.make_exchange(force_torch=True)
    elapsed = loop.timed(args.steps, head_pipelined, "kernel", record_kernel_times=True)
    kms = loop.kernel_times_ms(args.steps)
    if use_dist:
        okx, whyx = loop.verify_exchange()
        exchange_check["after"] = "ok" if okx else whyx
        if not okx and loop.native:
            print("bench.py: the library's own exchange failed its check after the timed loop (%s): torch.distributed instead, timed again" % whyx, file=sys.stderr)
            exchange_check["fallback"] = whyx
            loop.make_exchange(force_torch=True)
            elapsed = loop.timed(args.steps, head_pipelined, "kernel", record_kernel_times=True)
            kms = loop.kernel_times_ms(args.steps)
            okx, whyx = loop.verify_exchange()
            exchange_check["after"] = "ok (torch.distributed)" if okx else whyx
        if not okx:
            raise SystemExit("bench.py: the gathered table is wrong after the timed loop: %s" % whyx)
    # A bracket shorter than 50 ms (the driver's --steps 20: 4.5 ms) carries a fixed ~0.2 ms of its own -- first launches, the slowest
    # group's tail, the last synchronisation -- i.e. 4-5 % of 20 steps and 0.2 % of 400.  Then the same `steps`-long loop runs R times
    # back to back between two synchronisations, with an event per group stream at every boundary; ms_per_step = the MEDIAN repeat
    # (min / max and the single bracket beside it): the line no longer depends on --steps.
    single_bracket_s = elapsed
    timing = {"repeats": 1, "ms_per_step_single_bracket": elapsed / args.steps * 1e3, "ms_per_step_min": None, "ms_per_step_max": None,
              "ms_per_step_whole_region": None}
    if elapsed < 0.05 and not os.environ.get("DLSCQP_BENCH_SINGLE_BRACKET"):
        R = int(min(64, max(3, -(-0.05 // elapsed))))
        whole = loop.timed(args.steps, head_pipelined, "kernel", record_kernel_times=True, repeats=R)
        kms = loop.kernel_times_ms(args.steps)
        rep = np.asarray(loop.repeat_ms, dtype=np.float64)  # ms per repeat of `steps` steps
        if use_dist:  # a repeat is as long as its slowest rank's
            tt = torch.tensor(rep, dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            rep = tt.cpu().numpy()
        elapsed = float(np.median(rep)) * 1e-3
        timing = {"repeats": R, "ms_per_step_single_bracket": single_bracket_s / args.steps * 1e3,
                  "ms_per_step_min": float(rep.min()) / args.steps, "ms_per_step_max": float(rep.max()) / args.steps,
                  "ms_per_step_whole_region": whole / (R * args.steps) * 1e3,
                  "note": "steps x ms_per_step < 50 ms: the `steps`-long loop ran `repeats` times back to back between two barriers + "
                          "synchronisations (events on every group's stream at the boundaries); ms_per_step and value are the median "
                          "repeat; ms_per_step_single_bracket is ONE bracket of `steps` steps between two synchronisations (what "
                          "rounds 1-5 reported), ms_per_step_whole_region the whole region's wall time / (repeats x steps)"}
    if os.environ.get("DLSCQP_BENCH_DUMP_KMS"):  # developer aid: the launches' durations in the order each group recorded them
        print("kernel ms per launch, by group:", [[round(float(v), 4) for v in o.kernel_times_ms(args.steps)] for o in loop.opts], file=sys.stderr)
    host_queue_ms = loop.host_ms_per_step
    rank_breakdown = None
    if use_dist:
        # the stages of the exchange from a second, shorter pass of the same loop with events around them (the events cost
        # the host ~0.1 ms per step: not in the headline's pass)
        loop.xchg.enable_timing()
        loop.timed(max(B, min(args.steps, 100)), head_pipelined, "kernel")
        br = exchange_breakdown(loop.xchg, skip=B)  # the first B exchanges are the untimed warm-up steps of timed()
        loop.xchg.enable_timing(False)
        br["plan_kernel_ms"] = float(np.mean(kms)) if len(kms) else float("nan")
        rank_breakdown = over_ranks(dist, dev, br)
    if use_dist:
        # the gathered table must hold this rank's own block of the last step
        last = loop.out[(args.steps - 1) % B]
        ctrl_all, goal_all, status_all = loop.xchg.unpack(loop.xchg.table(), (torch.float32, torch.float32, torch.int32))
        torch.cuda.synchronize(dev)
        own = slice(rank * N, (rank + 1) * N)
        assert torch.equal(ctrl_all[own], last[0].reshape(N, -1)) and torch.equal(goal_all[own], last[1])
        assert torch.equal(status_all[own].flatten(), last[2].flatten())
        assert int((last[2] >= 0).sum()) == N  # every group wrote its slice
    res = loop.results()
    # the same loop without the launch-order hint, and (N > 1) with the other exchange mode
    side_steps = max(B, min(args.steps, 200))
    el_nohint = float("nan") if args.headline_only else loop.timed(side_steps, head_pipelined, None)
    el_itershint = float("nan") if args.headline_only else loop.timed(side_steps, head_pipelined, "iters")
    el_hostorder = float("nan") if args.headline_only else loop.timed(side_steps, head_pipelined, "host")
    el_sorted = float("nan") if args.headline_only else loop.timed(side_steps, head_pipelined, "device")
    el_halo_only = None
    if use_dist and pipelined_ok and not args.headline_only and getattr(loop, "native", False):
        # the step with the exchange restricted to what a consumer needs: the S64 swarms have no halo agents, so nothing has to travel
        loop.no_exchange = True
        el_halo_only = loop.timed(side_steps, False, "kernel")
        loop.no_exchange = False
    exch = None
    if use_dist and not args.headline_only:
        exch = {"pipelined": None, "serial": None, "steps_of_the_second_mode": side_steps}
        exch["pipelined" if head_pipelined else "serial"] = elapsed / args.steps * 1e3
        if pipelined_ok:  # the other mode, same batches, same launch order
            el_other = loop.timed(side_steps, not head_pipelined, "kernel")
            exch["serial" if head_pipelined else "pipelined"] = el_other / side_steps * 1e3
    G_used, per_launch = loop.G, loop.per
    native_xchg = bool(getattr(loop, "native", False))
    stream_note = getattr(loop, "stream_note", None)
    loop.close()
    del loop
    # ... and as ONE launch of all agents per step (rounds 1-2's loop): what the groups buy
    one_launch = None
    if G_used > 1 and not args.headline_only and not use_dist:
        l1 = SwarmLoop(1)
        one_launch = {"kernel_order": l1.timed(side_steps, False, "kernel") / side_steps * 1e3,
                      "sorted_on_device": l1.timed(side_steps, False, "device") / side_steps * 1e3,
                      "id_order": l1.timed(side_steps, False, None) / side_steps * 1e3}
        l1.close()
        del l1

    # rounds 1-5's batch: the same loop over the first 512 swarms of the workload (a step of 32 768 agents is as long as its slowest agent)
    at_512 = None
    if N > 512 * 64 and not args.headline_only and not use_dist:
        G5 = G_used
        while 512 % G5:
            G5 -= 1
        l5 = SwarmLoop(G5, n_use=512 * 64)
        ms5 = l5.timed(side_steps, False, "kernel") / side_steps * 1e3
        l5.close()
        del l5
        l51 = SwarmLoop(1, n_use=512 * 64)
        ms51 = l51.timed(side_steps, False, "kernel") / side_steps * 1e3
        l51.close()
        del l51
        at_512 = {"agents_per_gpu": 512 * 64, "swarm_groups": G5, "ms_per_step": ms5, "qp_per_s": 512 * 64 / ms5 * 1e3,
                  "one_launch_per_step": {"ms_per_step": ms51, "qp_per_s": 512 * 64 / ms51 * 1e3},
                  "note": "rounds 1-5 timed 512 swarms per GPU.  A step cannot be shorter than its slowest agent's solve (30-46 iterations, "
                          "~0.2 ms at three wavefronts per SIMD), and 32 768 agents no longer keep the chip busy for that long: the larger "
                          "batch is the same workload with enough of it (DESIGN.md section 8)"}

    # BASELINE.json configs[4] as a strong-scaling step on the same ranks (every N, so that the scaling runs carry it)
    stress = None
    if not args.no_extras and not args.headline_only:
        ss_all = StressStep(local_rank, rank, world, dist if use_dist else None, prune=False)
        ms_all_rows = ss_all.time(30, False)
        ss_all.close()
        ss_id = StressStep(local_rank, rank, world, dist if use_dist else None, order_hint=False)
        ms_id_order = ss_id.time(30, False)
        ss_id.close()
        ss = StressStep(local_rank, rank, world, dist if use_dist else None)
        stress_br = None
        if use_dist:
            ss.xchg.enable_timing()
        ms_serial = ss.time(30, False)
        if use_dist:
            stress_br = {"serial": over_ranks(dist, dev, exchange_breakdown(ss.xchg, skip=3))}
            ss.xchg.enable_timing()
        ms_pipe = ss.time(30, True) if use_dist else None
        if use_dist:
            stress_br["pipelined"] = over_ranks(dist, dev, exchange_breakdown(ss.xchg, skip=3))
            ss.xchg.enable_timing(False)
        stress = {"workload": "X2048 (BASELINE.json configs[4]): 2 048 agents in one 2-D maze-density crowd, M = 10, corridor boxes, "
                              "communication range 3, %.0f DLSC-GC rows per agent generated on the device; STRONG scaling, agents "
                              "sharded by contiguous id range" % ss.rows_per_agent,
                  "agents_total": ss.N, "agents_this_rank": ss.n_loc, "interior_agents_rank0": ss.n_interior, "halo_agents_rank0": ss.n_halo,
                  "ms_per_step_serial_exchange": ms_serial, "ms_per_step_pipelined_exchange": ms_pipe,
                  "exchange_per_rank_ms": stress_br,
                  "ms_per_step_without_presolve": ms_all_rows, "ms_per_step_id_launch_order": ms_id_order,
                  "launch_order": "every step computes the next step's launch order on the device from its own iteration counts "
                                  "(dlscqp_launch_order_device, inside the timed step); the batch is stationary, so the hint is "
                                  "better than a moving crowd's",
                  "rows_kept_per_agent_rank0": ss.rows_kept_per_agent(),
                  "presolve": "the half-space generator drops rows implied by the agent's corridor boxes (dlscqp_lsc_gen.prune): same "
                              "verdicts and solutions, fewer rows for the plan kernel to scan",
                  "agent_qp_per_s": ss.N / (min(ms_serial, ms_pipe) if ms_pipe else ms_serial) * 1e3,
                  "includes": "half-space generation + goal LP + QP" + (" + all-gather of the plans + copy into the plan table" if use_dist else ""),
                  "solved_fraction_rank0": ss.solved_fraction()}
        ss.close()
    # BASELINE.json configs[2]: empty60 sharded over the ranks, the exchange on the critical path
    sharded = None
    if use_dist and not args.no_extras and not args.headline_only:
        sharded = sharded_mission(dist, rank, world, local_rank)

    if rank == 0:
        ok = np.concatenate([r["status"] for r in res]) == abi.OK
        it = np.concatenate([r["iters"] for r in res])
        dim, M, n = cfg.world_dimension, cfg.M, cfg.n
        bytes_qp = algorithmic_bytes_per_qp(dim, M, n, K, False, 0)
        avg_kernel_s = float(np.mean(kms)) * 1e-3 if len(kms) else float("nan")
        # the roofline figure is what the STEP achieves: the algorithmic bytes of all agents of a step over the step's wall time
        # (the same clock as `value`).  Per launch (HIP events on the launch streams) and "if the groups' launches overlapped for
        # their whole duration" are kept beside it as named extras: launches of different groups overlap only partly.
        step_s = elapsed / args.steps
        achieved = N * bytes_qp / step_s / 1e9
        achieved_if_overlapped = G_used * per_launch * bytes_qp / avg_kernel_s / 1e9
        traffic, traffic_src, valu_frac, prof = None, None, None, {}
        try:
            with open(args.traffic_json) as fh:
                prof = json.load(fh)
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            import summarize
            if prof.get("agents_per_launch") == per_launch and prof.get("csrc_sha1") == summarize.csrc_sha1():
                traffic = prof.get("hbm_bytes_per_launch")
                if prof.get("valu_insts_per_agent"):
                    # share of the chip's vector-pipe cycles over the STEP: a wave64 FP64 instruction holds its SIMD's pipe for 4
                    # cycles, any other VALU instruction for 2; 1 024 SIMDs; 2.4 GHz nominal (the counters are per launch and do
                    # not depend on the overlap).  Without FP64 counters in the profile: every instruction at 4.
                    f64 = prof.get("valu_f64_insts_per_agent")
                    cyc = 4.0 * prof["valu_insts_per_agent"] if f64 is None else 4.0 * f64 + 2.0 * (prof["valu_insts_per_agent"] - f64)
                    valu_frac = cyc * N / (1024 * step_s * 2.4e9)
                traffic_src = "%s: rocprofv3 --pmc passes of this command (%s), taken on these kernel sources" % (
                    os.path.relpath(args.traffic_json, ROOT), prof.get("source", ""))
            elif prof.get("agents_per_launch") == per_launch:
                traffic_src = "%s was measured on other kernel sources (csrc hash differs): not reported" % os.path.relpath(args.traffic_json, ROOT)
        except Exception:
            pass
        R_rows, P_pts = K + 2 * 207, M * (n + 1)
        # per active-set iteration: one pricing pass over the R rows (6 flops a row), two triangular products with the q x q factor
        # (2 q^2), the step direction G (a - A'r) (one table row of P entries per active row and axis: 2 dim P q); the working
        # set grows from 0 to `active` over the adds, q ~ active / 2 on average
        q_avg = it[:, 3] / 2.0
        flops_qp = float((it[:, 0] * (6.0 * R_rows + 2.0 * q_avg * q_avg + 2.0 * dim * P_pts * (q_avg + 1.0))).mean()
                         + 6.0 * R_rows)  # + the scan that finds no violated row
        fp64 = {"flops_per_qp_model": flops_qp, "achieved_tflops": N * flops_qp / avg_kernel_s / 1e12,
                "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac": N * flops_qp / avg_kernel_s / 1e12 / FP64_VALU_PEAK_TFLOPS}
        lat, closed, hard = None, None, None
        if not args.no_extras and not args.headline_only and world == 1:
            # single-swarm replanning latency through the host-buffer entry (H2D + solve + D2H)
            a1, r1 = wl["agents"][:64], wl["rows"][:wl["row_offsets"][64]]
            ro1 = wl["row_offsets"][:65]
            lat = []
            for _ in range(1020):  # 1 000 samples: a p99 from fifty of them is the slowest call, not a percentile (profiles/r03_latency_tail.md)
                t1 = time.perf_counter()
                opt.plan(a1, ro1, r1, None, 0, True, True)
                lat.append((time.perf_counter() - t1) * 1e3)
            lat = np.array(lat[20:])
            # "hard" batch: only agents that end with at least five active rows, taken from the timed batches and tiled to the
            # same launch size.  Launch order: like the headline's, from what a closed loop knows -- the iteration count the SAME
            # agent had one replanning step earlier (the previous batch of the rotation) -- sorted on the device
            sel = [(j, a) for j in range(B) for a in np.nonzero((res[j]["iters"][:, 3] >= 5) & (res[j]["status"] == abi.OK))[0]]
            if len(sel) >= 64:
                pick = [sel[i % len(sel)] for i in range(N)]
                ha = np.concatenate([hist[j]["agents"][a:a + 1] for j, a in pick])
                hr = np.concatenate([hist[j]["rows"][a * K:(a + 1) * K] for j, a in pick])
                prev = np.zeros((N, 4), np.int32)
                prev[:, 0] = [res[(j - 1) % B]["iters"][a, 0] for j, a in pick]
                prev_t = torch.from_numpy(prev).to(dev)
                hms = {}
                hit = None
                # as one launch of all agents per step, and in the headline's groups (each group on its own stream)
                for gname, Gh in (("one_launch", 1), ("groups", G_used)):
                    ph = N // Gh
                    hopts = [opt] + [BatchTrajOptimizer(cfg) for _ in range(Gh - 1)]
                    hstreams = [torch.cuda.Stream(dev) for _ in range(Gh)]
                    hdbs = [hopts[g].make_device_batch(ha[g * ph:(g + 1) * ph], np.arange(ph + 1) * K, hr[g * ph * K:(g + 1) * ph * K], None, 0,
                                                       want_iters=True) for g in range(Gh)]
                    for db_ in hdbs:
                        db_.use_row_pitch()
                    for tag in ("id_order", "hinted"):
                        if tag == "hinted":
                            for g in range(Gh):
                                # the class masks a launch of the same agents one step earlier would have left (hint_mask_in)
                                pc = prev_t[g * ph:(g + 1) * ph, 0].cpu().numpy().astype(np.int64)
                                cl = np.where(pc <= 0, 0, np.minimum(7, np.floor(np.log2(np.maximum(pc, 1))).astype(np.int64) + 1))
                                mk = np.zeros(((ph + 63) // 64, 8), np.uint64)
                                np.bitwise_or.at(mk, (np.arange(ph) // 64, cl), np.uint64(1) << (np.arange(ph) % 64).astype(np.uint64))
                                hdbs[g].set_hint_masks(torch.from_numpy(mk.view(np.int64)).to(dev), None, None)
                        torch.cuda.synchronize(dev)
                        for rep_ in range(105):
                            if rep_ == 5:
                                torch.cuda.synchronize(dev)
                                t1 = time.perf_counter()
                            for g in range(Gh):
                                with torch.cuda.stream(hstreams[g]):
                                    hdbs[g].launch(True, True)
                        torch.cuda.synchronize(dev)
                        hms[gname, tag] = (time.perf_counter() - t1) / 100 * 1e3
                    if hit is None:
                        hit = hdbs[0].results()["iters"]
                    del hdbs
                    for o in hopts[1:]:
                        o.close()
                hard = {"agents_per_step": N, "groups": G_used, "distinct_agents": len(sel), "ms_per_step": hms["groups", "hinted"],
                        "qp_per_s": N / hms["groups", "hinted"] * 1e3, "ms_per_step_id_order": hms["groups", "id_order"],
                        "ms_per_step_one_launch": hms["one_launch", "hinted"], "ms_per_step_one_launch_id_order": hms["one_launch", "id_order"],
                        "qp_per_s_one_launch": N / hms["one_launch", "hinted"] * 1e3,
                        "mean_active_set_iterations": float(hit[:, 0].mean()), "mean_active_rows": float(hit[:, 3].mean()),
                        "note": "every agent has >= 5 active rows at its solution; launched in the order of the iteration counts the same "
                                "agents had one replanning step earlier (dlscqp_device_batch.hint_iters), in the headline's groups; the "
                                "one_launch figures are rounds 1-2's shape (one launch of all agents per step)"}
            closed = closed_loop_latency(local_rank)
            # two loops of 256 swarms on two streams (round 5, one MI355X: one loop 0.515 ms per step, two 0.436, four 0.564; as HIP
            # graphs the same within 1 % -- the step is not bound by the host)
            closed["s64_closed_loop"] = s64_closed_loop(local_rank, n_swarms=min(args.swarms, 512), groups=2)
            one = s64_closed_loop(local_rank, n_swarms=min(args.swarms, 512), groups=1, flights=2)
            closed["s64_closed_loop"]["ms_per_step_one_loop"] = one["ms_per_step"]
        line = {
            "metric": "agent-QP solves/sec (goal LP + trajectory QP per agent per replanning step)",
            "value": world * N * args.steps / elapsed,
            "unit": "QP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "timing": timing,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not one_device else "synthetic; TEST HOOK: all ranks on one device over gloo -- not a measurement",
            "config": {
                "workload": "S64: %d independent 64-agent swarms per GPU per step (BASELINE.json configs[1]: 3-D, 5 segments x degree 5, "
                            "%d DLSC rows per agent); the timed loop rotates through %d consecutive replanning steps of the closed-loop "
                            "generator" % (args.swarms, K, B),
                "agents_per_gpu": N, "swarm_groups": G_used, "agents_per_launch": per_launch,
                # "prebuilt": libdlscqp.so came with the tree (newer than all of its sources: nothing was compiled on this box);
                # "compiled": this process ran hipcc; "stale": the library is older than a source (the numbers are not the tree's)
                "build": build_state,
                "swarm_groups_note": "the %d swarms of a GPU are planned in %d independent groups, each on its own stream with its own sequence "
                                     "of replanning steps (launch, next launch order from its own iteration counts%s); swarms have no neighbours "
                                     "outside themselves, so no group waits for another, and a step is every group advanced by one replanning "
                                     "step%s" % (args.swarms, G_used, "", "; one all-gather of all groups' plans per step, posted from a stream that waits for "
                                                                        "the groups' launches of that step" if use_dist else ""),
                "swarm_group_streams": stream_note, "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "ms_per_step_one_launch_per_step": one_launch,
                "host_ms_to_queue_a_step": host_queue_ms,
                "segments": M, "degree": n, "lsc_rows_per_agent": K,
                "vars_per_agent": dim * M * (n + 1), "ineq_rows_per_agent": K + 2 * 207, "distinct_batches": B,
                "exchange": ("none (1 GPU)" if not use_dist else
                             "one RCCL all-gather of control points, sub-goal and status per step, posted on a communication stream: runs "
                             "while the next step plans the agents without remote neighbours (all of them here)"
                             if head_pipelined else "one RCCL all-gather of control points, sub-goal and status per step, after the launch"),
                # the whole gathered table against torch.distributed's own all-gather of the same records, before and after the timed loop
                # ("ok", or what differed); fallback: why the library's exchange was replaced by torch.distributed for this run
                "exchange_check": exchange_check,
                "exchange_issued_by": (None if not use_dist else
                                       "the library (dlscqp_exchange_post: ncclAllGather from librccl on a communication stream the exchange owns); "
                                       "the whole step -- every group's launch and the post -- is one call, dlscqp_step_device" if native_xchg
                                       else ("torch.distributed.all_gather_into_tensor (FALLBACK: the library's own exchange failed its check: %s)" % exchange_check["fallback"]
                                             if exchange_check and exchange_check.get("fallback") else "torch.distributed.all_gather_into_tensor (gloo test hook, or DLSCQP_NATIVE_EXCHANGE=0)")),
                "exchange_modes_ms_per_step": exch,
                "ms_per_step_exchange_restricted_to_halo_consumers": None if el_halo_only is None else el_halo_only / side_steps * 1e3,
                "exchange_restricted_note": "the reference broadcasts every plan to every agent (multi_sync_simulator.cpp:468-514) and the "
                                            "headline mirrors that: every rank receives the whole table.  The S64 swarms have no neighbours "
                                            "outside themselves, so NO agent reads a remote plan: restricted to what a consumer needs the "
                                            "exchange moves nothing, and this is the step then",
                "launch_order": "found by the plan kernel itself from the class masks the previous replanning step's launch of the same agents "
                                "left behind (dlscqp_device_batch.hint_mask_*: per block of 64 agents eight 64-bit masks, one per class of "
                                "iteration counts -- this step's count or one class below the class the agent was scheduled with -- into "
                                "which every agent ORs its bit when it finishes): slot s plans the (s / n_blocks)-th most expensive agent of "
                                "the block s %% n_blocks -- 64 bytes of scalar loads and ~40 scalar instructions per agent, no sort kernel, no "
                                "stream operation between two launches; correlation of consecutive steps' counts %.2f.  Round 4's form (every "
                                "wavefront loads its block's 64 keys, hint_iters) is timed as ms_per_step_order_from_hint_iters" % hint_corr,
                "ms_per_step_sorted_on_device": None if args.headline_only else el_sorted / side_steps * 1e3,
                "sorted_on_device": "round 3's method: the same key sorted by dlscqp_launch_order_history_device after every step (two "
                                    "more kernels per group and step, inside the timed loop)",
                "ms_per_step_host_order": None if args.headline_only else el_hostorder / side_steps * 1e3,
                "host_order": "the same hint as permutations prepared on the host before the timed loop (round 1's method)",
                "ms_per_step_no_order_hint": None if args.headline_only else el_nohint / side_steps * 1e3,
                "ms_per_step_order_from_hint_iters": None if args.headline_only else el_itershint / side_steps * 1e3,
                "qp_per_s_no_order_hint": None if args.headline_only else world * N * side_steps / el_nohint,
                "solved_fraction": float(ok.mean()), "mean_active_set_iterations": float(it[:, 0].mean()),
                "max_active_set_iterations": int(it[:, 0].max()), "zero_iteration_fraction": float((it[:, 0] == 0).mean()),
                "seed": args.seed, "workload_generation_s": t_gen,
            },
            "n_gpus_requested": args.gpus, "n_gpus_visible": int(torch.cuda.device_count()),
            # the figure that compares like for like with rounds 1-2 (ONE launch of all the GPU's agents per step, in lockstep); `value`
            # is the swarm groups advancing independently on their own streams (config.swarm_groups_note)
            "at_512_swarms": at_512,
            "scaling_model": scaling_model(elapsed / args.steps * 1e3 - (0.02 if use_dist else 0.0), N),
            "one_launch_per_step": None if not isinstance(one_launch, dict) or not one_launch.get("kernel_order") else
                                   {"ms_per_step": one_launch["kernel_order"], "qp_per_s": world * N / one_launch["kernel_order"] * 1e3},
            "per_rank_ms": rank_breakdown,
            "empty60_sharded": sharded,
            "hard_batch": hard,
            "x2048_strong": stress,
            "p50_replan_ms_one_swarm_host_to_host": None if lat is None else float(np.percentile(lat, 50)),
            "p99_replan_ms_one_swarm_host_to_host": None if lat is None else float(np.percentile(lat, 99)),
            "max_replan_ms_one_swarm_host_to_host": None if lat is None else float(lat.max()),
            "replan_ms_one_swarm_samples": None if lat is None else int(len(lat)),
            "closed_loop_replan_ms": closed,
            # flop model of the range-space iteration (see where flops_qp is computed): SURVEY.md 8(d)'s dense model (2 r^2 sweeps
            # of an orthogonal factor) no longer describes what the kernel does -- its arithmetic per iteration is O(active rows)
            "fp64_valu": fp64,
            "roofline": {
                # achieved / peak / frac are the HBM figures the contract asks for.  What binds the kernel is neither classical roof
                # (SURVEY.md 8d): ~5 KB and ~0.1 MFLOP per QP.  `limiter` says what the counters show (profiles/): the SIMDs' vector
                # pipes are busy for valu_issue_frac of their cycles (FP64 instructions at 4 cycles, the others at 2), the wavefronts
                # wait for wait_any_frac of theirs -- two wavefronts per SIMD (244-250 VGPRs) of dependent short phases: wave
                # issue / latency, not a throughput roof.
                # (`bound` names the roof that achieved / peak / frac refer to -- the contract's "hbm" | "mfma" -- not what limits the kernel:
                # that is `limiter`)
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None if traffic is None else traffic * G_used, "traffic_source": traffic_src,
                "limiter": "wave issue and dependent latencies at two wavefronts per SIMD (neither the HBM nor the FP64 roof)",
                "valu_issue_frac": valu_frac, "wait_any_frac": prof.get("wait_any_frac") if traffic is not None else None,
                "valu_insts_per_qp": prof.get("valu_insts_per_agent") if traffic is not None else None,
                "valu_f64_insts_per_qp": prof.get("valu_f64_insts_per_agent") if traffic is not None else None,
                "salu_insts_per_qp": prof.get("salu_insts_per_agent") if traffic is not None else None,
                # matrix-core instructions of the plan kernel (SQ_INSTS_MFMA per launch): zero BY DESIGN -- the active-set iteration is
                # matrix-vector work on <= 16 column records, and FP64 MFMA has the FP64 VALU's peak on this chip
                "mfma_insts": prof.get("mfma_insts_per_launch") if traffic is not None else None,
                "mfma_busy_cycles": prof.get("mfma_busy_cycles_per_launch") if traffic is not None else None,
                "kernel": prof.get("kernel", "dlscqp_plan_kernel"), "avg_kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_qp": bytes_qp, "qp_per_launch": per_launch, "launches_per_step": G_used,
                "achieved_one_launch": per_launch * bytes_qp / avg_kernel_s / 1e9,
                "achieved_if_launches_overlapped_fully": achieved_if_overlapped,
                "effective_concurrency": G_used * avg_kernel_s / step_s,
                "note": "achieved = algorithmic bytes (SURVEY.md 8d) of all agents of a step / ms_per_step -- the step's own clock, "
                        "as `value`; achieved_one_launch = one launch's bytes / its average duration (HIP events on the launch "
                        "streams over the timed region); achieved_if_launches_overlapped_fully = that x launches_per_step (round 3 "
                        "reported this as `achieved`; the launches of different groups overlap only partly: effective_concurrency); "
                        "traffic = PMC bytes per launch x launches_per_step; valu_issue_frac = (4 x FP64 + 2 x other VALU instructions "
                        "of a step) / (1 024 SIMDs x ms_per_step x 2.4 GHz); ~5 KB per QP can never load the HBM roof (SURVEY.md 8d)",
            },
        }
        if not args.no_cpu_baseline and not args.headline_only and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, hist[-1])
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    opt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
