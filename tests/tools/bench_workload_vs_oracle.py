#!/usr/bin/env python3
"""Manual tool: the batches bench.py times (S64, four consecutive replanning steps of the closed-loop generator) planned by the HIP
kernel and by the CPU oracle, agent by agent: verdicts, sub-goals, control points.
    python tests/tools/bench_workload_vs_oracle.py [--swarms 512]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=512)
    a = ap.parse_args()
    import numpy as np
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    import oracle_lib as O
    cfg = workloads.s64_config(device=0)
    opt = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda ag, ro, r, s: opt.plan(ag, ro, r, s, 0, True, True), n_swarms=a.swarms, swarm_size=64, seed=2000, keep=4)
    tot = bad = 0
    for j, h in enumerate(wl["history"]):
        db = opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_f64=True, want_iters=True)
        db.launch()
        got = db.results()
        t0 = time.time()
        ref = O.plan_batch(cfg, h["agents"], h["row_offsets"], h["rows"], None, 0, True)
        sm = (got["status"] != ref["status"]) | (got["goal_status"] != ref["goal_status"])
        ok = (ref["status"] == abi.OK) & (got["status"] == abi.OK)
        gm = (got["goal"] != ref["goal"]).any(1) & (ref["goal_status"] != abi.INFEASIBLE)
        errx = np.abs(got["x"] - ref["x"]).max(1) * ok
        errc = np.abs(got["ctrl"].astype(np.float64) - ref["ctrl"]).max(axis=(1, 2, 3)) * ok
        n = len(sm)
        tot += n
        bad += int(sm.sum() + gm.sum() + (errx > 1e-6).sum())
        print("batch %d: %d agents, solved %.3f, infeasible %d, verdict mismatches %d, sub-goal mismatches %d, max |x - oracle| %.2e, max |ctrl - oracle| %.2e, "
              "most iterations %d (oracle %.0f s)" % (j, n, ok.mean(), int((ref["status"] == abi.INFEASIBLE).sum()), int(sm.sum()), int(gm.sum()), errx.max(),
                                                      errc.max(), got["iters"][:, 0].max(), time.time() - t0), flush=True)
    print("TOTAL", tot, "problems", bad)
    opt.close()


if __name__ == "__main__":
    main()
