#!/usr/bin/env python3
"""Where the tail of the single-swarm host-to-host latency comes from (bench.py: p50/p99_replan_ms_one_swarm_host_to_host).

1 000 calls of the host-buffer entry point (dlscqp_plan_batch: marshal -> H2D -> launch -> D2H -> wait) on the SAME 64-agent
swarm, so the kernel's work is identical every call; per call the wall time around the C call and the kernel's own time (HIP
events on the launch stream).  What is left is the host: marshalling, the runtime's submission and how the wait for the
stream ends.

    python profiles/latency_tail.py [--calls 1000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=1000)
    args = ap.parse_args()
    import numpy as np
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=1, swarm_size=64, seed=2000)
    a1, ro1, r1 = wl["agents"], wl["row_offsets"], wl["rows"]
    for _ in range(20):
        opt.plan(a1, ro1, r1, None, 0, True, True)
    for mode in (os.environ.get("DLSCQP_HOST_WAIT", "default"),):
        opt.set_timing(args.calls)
        wall = np.zeros(args.calls)
        for i in range(args.calls):
            t0 = time.perf_counter()
            opt.plan(a1, ro1, r1, None, 0, True, True)
            wall[i] = (time.perf_counter() - t0) * 1e3
        k = opt.kernel_times_ms(args.calls)
        host = wall - k
        q = lambda v, p: float(np.percentile(v, p))  # noqa: E731
        print("wait mode %s, %d calls of one 64-agent swarm (identical inputs):" % (mode, args.calls))
        print("  wall   p50 %.4f  p90 %.4f  p99 %.4f  max %.4f ms   (p99 / p50 = %.2f)" % (q(wall, 50), q(wall, 90), q(wall, 99), wall.max(), q(wall, 99) / q(wall, 50)))
        print("  kernel p50 %.4f  p90 %.4f  p99 %.4f  max %.4f ms   (HIP events; the same work every call)" % (q(k, 50), q(k, 90), q(k, 99), k.max()))
        print("  host   p50 %.4f  p90 %.4f  p99 %.4f  max %.4f ms   (wall - kernel: marshalling, submission, copies, the wait)" % (q(host, 50), q(host, 90), q(host, 99), host.max()))
        worst = np.argsort(-wall)[:5]
        print("  five slowest calls: " + ", ".join("#%d %.3f ms (kernel %.3f)" % (i, wall[i], k[i]) for i in worst))
    opt.close()


if __name__ == "__main__":
    main()
