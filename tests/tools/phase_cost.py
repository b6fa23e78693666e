#!/usr/bin/env python3
"""Developer tool: what the phases of dlscqp_plan_kernel cost in THROUGHPUT (launch time per 32 768 agents), not in one
wavefront's latency: a -DDQ_PHASE_PROFILE build (csrc/libdlscqp_phase.so) ends every agent's solve after phase k
(DLSCQP_DEBUG_EXIT_PHASE=k: 0 = agent record + row cache + goal LP, 1 = + x0 / bound table / scalings, 2 = + first pricing
pass); the differences of the launch times are the phases' costs.  A batch made only of agents that never iterate
gives the cost of the read-out as well.

    python tests/tools/phase_cost.py [--swarms 512]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=512)
    args = ap.parse_args()
    lib = os.path.join(CSRC, "libdlscqp_phase.so")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-DDQ_PHASE_PROFILE", "-o", lib] +
                              [os.path.join(CSRC, f) for f in ("dlscqp.hip", "qp_tables.cpp", "sfc_world.cpp", "grid_mapf.cpp")])
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    dev = torch.device("cuda", 0)
    cfg = workloads.s64_config(device=0)
    opt0 = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt0.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, swarm_size=64, seed=2000, keep=4)
    hist = wl["history"]
    N, K = len(wl["agents"]), wl["K"]
    res = [opt0.plan(h["agents"], h["row_offsets"], h["rows"], None, 0, True, True) for h in hist]
    its = [opt0.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
    for db in its:
        db.launch(True, True)
    iters = [db.results()["iters"] for db in its]
    del its, res
    sel = [(j, a) for j in range(4) for a in np.nonzero(iters[j][:, 0] == 0)[0]]
    pick = [sel[i % len(sel)] for i in range(N)]
    za = np.concatenate([hist[j]["agents"][a:a + 1] for j, a in pick])
    zr = np.concatenate([hist[j]["rows"][a * K:(a + 1) * K] for j, a in pick])
    opt0.close()
    abi._LIB = None
    os.environ["DLSCQP_LIB"] = lib
    os.environ["DLSCQP_NO_SELFTEST"] = "1"  # a kernel that stops after a phase does not pass the self-test of dlscqp_create, by design
    print("launch time of %d agents [ms]; 'rotation' = bench.py's four batches in id order, 'zero' = only agents that never iterate" % N)
    prev = {}
    for ph, name in ((0, "agent record, row/slot cache, goal LP"), (1, "+ x0, bound table, scalings"), (2, "+ first pricing pass"),
                     (-1, "whole solve")):
        os.environ["DLSCQP_DEBUG_EXIT_PHASE"] = str(ph)
        opt = BatchTrajOptimizer(cfg)
        dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
        zdb = opt.make_device_batch(za, np.arange(N + 1) * K, zr, None, 0, want_iters=True)
        if not os.environ.get("DLSCQP_NO_ROW_PITCH"):
            for db in dbs + [zdb]:
                db.use_row_pitch()  # as bench.py launches the workload
        out = {}
        for tag, group in (("rotation", dbs), ("zero", [zdb])):
            for i in range(8):
                group[i % len(group)].launch(True, True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(200):
                group[i % len(group)].launch(True, True)
            torch.cuda.synchronize(dev)
            out[tag] = (time.perf_counter() - t0) / 200 * 1e3
        print("  exit after phase %2d (%-40s): rotation %.4f (+%.4f)   zero %.4f (+%.4f)" %
              (ph, name, out["rotation"], out["rotation"] - prev.get("rotation", 0.0), out["zero"], out["zero"] - prev.get("zero", 0.0)), flush=True)
        prev = out
        del dbs, zdb
        opt.close()


if __name__ == "__main__":
    main()
