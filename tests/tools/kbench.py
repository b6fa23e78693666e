#!/usr/bin/env python3
"""Developer tool: A/B of builds of libdlscqp.so on the SAME batches, in one process.

    python tests/tools/kbench.py --libs a.so,b.so [--swarms 512] [--steps 200]

The S64 workload (bench.py's: four consecutive replanning steps of the closed-loop generator) is generated once with the
first library; every library then plans the same four batches in bench.py's rotation with the launch order computed on
the device from the previous steps' iteration counts, and a "hard" batch (agents with >= 5 active rows, each ordered by
the iteration count the SAME agent had one replanning step earlier).  Prints one line per library."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--swarms", type=int, default=512)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=2000)
    args = ap.parse_args()
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    libs = [l for l in args.libs.split(",") if l] or [abi.library_path()]
    dev = torch.device("cuda", 0)
    cfg = workloads.s64_config(device=0)
    wl = None
    B = 4
    for li, lib in enumerate(libs):
        abi._LIB = None
        os.environ["DLSCQP_LIB"] = os.path.abspath(lib)
        opt = BatchTrajOptimizer(cfg)
        if wl is None:
            wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, swarm_size=64,
                                    seed=args.seed, keep=B)
        hist = wl["history"]
        N, K = len(wl["agents"]), wl["K"]
        dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
        for i in range(2 * B):
            dbs[i % B].launch(True, True)
        torch.cuda.synchronize(dev)
        res = [db.results() for db in dbs]

        def step(c):
            db, nxt = dbs[c % B], dbs[(c + 1) % B]
            db.launch(True, True)
            j = c % B
            older = [dbs[(j - k) % B].out_iters for k in (1, 2) if (j - k) % B != (j + 1) % B]
            nxt.update_order(iters=db.out_iters, history=older)

        for c in range(2 * B):
            step(c)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c in range(args.steps):
            step(c)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        for db in dbs:
            db.set_order(None)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c in range(args.steps):
            dbs[c % B].launch(True, True)
        torch.cuda.synchronize(dev)
        ms_id = (time.perf_counter() - t0) / args.steps * 1e3
        it = np.stack([r["iters"][:, 0] for r in res])
        # hard batch: agent a of step j (>= 5 active rows), hinted by its own count at step j-1
        sel = [(j, a) for j in range(1, B) for a in np.nonzero((res[j]["iters"][:, 3] >= 5) & (res[j]["status"] == abi.OK))[0]]
        pick = [sel[i % len(sel)] for i in range(N)]
        ha = np.concatenate([hist[j]["agents"][a:a + 1] for j, a in pick])
        hr = np.concatenate([hist[j]["rows"][a * K:(a + 1) * K] for j, a in pick])
        hint = np.array([res[j - 1]["iters"][a, 0] for j, a in pick])
        hdb = opt.make_device_batch(ha, np.arange(N + 1) * K, hr, None, 0, want_iters=True)
        out = {}
        for tag, order in (("id", None), ("hint", np.argsort(-hint, kind="stable").astype(np.int32))):
            hdb.set_order(order)
            for _ in range(5):
                hdb.launch(True, True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(100):
                hdb.launch(True, True)
            torch.cuda.synchronize(dev)
            out[tag] = (time.perf_counter() - t0) / 100 * 1e3
        hit = hdb.results()["iters"]
        print("%-40s headline %.4f ms (%.1f M QP/s; id order %.4f ms)  iters mean %.2f max %d zero %.2f | hard: id %.4f ms, hinted %.4f ms "
              "(%.1f M QP/s), iters mean %.2f active %.2f" %
              (os.path.basename(lib), ms, N / ms / 1e3, ms_id, it.mean(), it.max(), (it == 0).mean(), out["id"], out["hint"],
               N / out["hint"] / 1e3, hit[:, 0].mean(), hit[:, 3].mean()), flush=True)
        del hdb, dbs
        opt.close()


if __name__ == "__main__":
    main()
