#!/usr/bin/env python3
"""Developer tool: the S64 loop with its swarms in G independent groups, each group on its own stream with its own sequence
of replanning steps (launch, launch order for the group's next step from its own iteration counts): the tail of one group's
launch runs beside the other groups' launches.  Swarms are independent (no neighbour in another swarm), so no group ever
waits for another.  Prints ms per step of ALL swarms for G = 1, 2, 4.

    python tests/tools/two_groups.py [--swarms 512] [--steps 200]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=512)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    dev = torch.device("cuda", 0)
    cfg = workloads.s64_config(device=0)
    opt0 = BatchTrajOptimizer(cfg)
    B = 4
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt0.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, swarm_size=64, seed=2000, keep=B)
    hist, K, N = wl["history"], wl["K"], len(wl["agents"])
    opt0.close()
    for G in [int(x) for x in os.environ.get("NGROUPS", "1,2,4").split(",")]:
        per = N // G
        opts = [BatchTrajOptimizer(cfg) for _ in range(G)]
        streams = [torch.cuda.Stream(dev) for _ in range(G)]
        dbs = [[opts[g].make_device_batch(h["agents"][g * per:(g + 1) * per], np.arange(per + 1) * K, h["rows"][g * per * K:(g + 1) * per * K], None, 0,
                                          want_iters=True) for h in hist] for g in range(G)]

        def step(c):
            j = c % B
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    db, nxt = dbs[g][j], dbs[g][(j + 1) % B]
                    db.launch(True, True)
                    older = [dbs[g][(j - k) % B].out_iters for k in (1, 2) if (j - k) % B != (j + 1) % B]
                    nxt.update_order(iters=db.out_iters, history=older)

        if os.environ.get("STALE_ORDER"):
            # the order of batch j+1 from the counts of batches j-1 and j-2, sorted on a side stream while batch j is planned
            side = [torch.cuda.Stream(dev) for _ in range(G)]
            done = [[None] * B for _ in range(G)]
            sorted_ev = [[None] * B for _ in range(G)]

            def step(c):  # noqa: F811
                j = c % B
                for g in range(G):
                    db = dbs[g][j]
                    with torch.cuda.stream(streams[g]):
                        if sorted_ev[g][j] is not None:
                            streams[g].wait_event(sorted_ev[g][j])
                        db.launch(True, True)
                        done[g][j] = torch.cuda.Event()
                        done[g][j].record(streams[g])
                    with torch.cuda.stream(side[g]):
                        if done[g][(j - 1) % B] is not None:
                            side[g].wait_event(done[g][(j - 1) % B])
                            nxt = dbs[g][(j + 1) % B]
                            nxt.update_order(iters=dbs[g][(j - 1) % B].out_iters, history=[dbs[g][(j - 2) % B].out_iters])
                            sorted_ev[g][(j + 1) % B] = torch.cuda.Event()
                            sorted_ev[g][(j + 1) % B].record(side[g])
        if os.environ.get("PRIO_SORT"):
            # the three stream operations of the launch order on a high-priority stream per group: same dependencies, but their
            # workgroups do not queue behind the other groups' plan kernels
            side = [torch.cuda.Stream(dev, priority=-1) for _ in range(G)]
            e_plan = [torch.cuda.Event() for _ in range(G)]
            e_sort = [torch.cuda.Event() for _ in range(G)]

            def step(c):  # noqa: F811
                j = c % B
                for g in range(G):
                    db, nxt = dbs[g][j], dbs[g][(j + 1) % B]
                    if c:
                        streams[g].wait_event(e_sort[g])
                    db.launch(True, True, stream=streams[g])
                    e_plan[g].record(streams[g])
                    side[g].wait_event(e_plan[g])
                    older = [dbs[g][(j - k) % B].out_iters for k in (1, 2) if (j - k) % B != (j + 1) % B]
                    nxt.update_order(stream=side[g], iters=db.out_iters, history=older)
                    e_sort[g].record(side[g])
        for c in range(2 * B):
            step(c)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c in range(args.steps):
            step(c)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print("%d group(s) of %d agents: %.4f ms per step of all %d agents (%.1f M agent-QP/s)" % (G, per, ms, N, N / ms / 1e3), flush=True)
        del dbs
        for o in opts:
            o.close()


if __name__ == "__main__":
    main()
