#!/usr/bin/env python3
"""Developer tool: A/B of builds of libdlscqp.so on the SAME batches in one process, the way bench.py runs them.

    python tests/tools/ab.py --libs a.so,b.so [--swarms 512] [--steps 300] [--reps 2]

bench.py's S64 workload (four consecutive replanning steps of the closed-loop generator) is generated once with the first library;
every library then plans the same four batches (a) in four swarm groups on four streams, (b) as one launch of all agents per step,
both with the kernel's own launch order (hint_iters: the other batches' iteration counts), and (c) a "hard" batch -- only agents
with >= 5 active rows -- ordered by the count the same agent had one step earlier.  One line per library and repetition; the
libraries take turns (a, b, a, b ...) so that a drift of the box shows up as a difference between repetitions, not between builds."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--swarms", type=int, default=512)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--seed", type=int, default=2000)
    ap.add_argument("--no-hard", action="store_true")
    ap.add_argument("--order", default="masks", help="masks (hint_mask_*: round 5), iters (hint_iters: round 4), none; comma-separated: one line each")
    args = ap.parse_args()
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    libs = [l for l in args.libs.split(",") if l] or [abi.library_path()]
    dev = torch.device("cuda", 0)
    cfg = workloads.s64_config(device=0)
    B = 4
    wl, ref = None, None
    streams = None
    for rep in range(args.reps):
      for order in args.order.split(","):
        for lib in libs:
            abi._LIB = None
            os.environ["DLSCQP_LIB"] = os.path.abspath(lib)
            opt = BatchTrajOptimizer(cfg)
            if wl is None:
                wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, swarm_size=64,
                                        seed=args.seed, keep=B)
            hist = wl["history"]
            N, K = len(wl["agents"]), wl["K"]
            out = {}
            res_all = None
            for G in (4, 1):
                per = N // G
                opts = [opt] + [BatchTrajOptimizer(cfg) for _ in range(G - 1)]
                dbs = [[opts[g].make_device_batch(h["agents"][g * per:(g + 1) * per], np.arange(per + 1) * K, h["rows"][g * per * K:(g + 1) * per * K],
                                                   None, 0, want_iters=True) for h in hist] for g in range(G)]
                for g in range(G):
                    for db in dbs[g]:
                        db.use_row_pitch()
                if streams is None:
                    # four streams whose work really runs side by side (bench.py: SwarmLoop.pick_streams -- streams share a few
                    # hardware queues, and two streams on one queue run one after the other)
                    cands = [torch.cuda.Stream(dev) for _ in range(12)]

                    def span(sa, sb):
                        best = float("inf")
                        for _ in range(3):
                            torch.cuda.synchronize(dev)
                            t0 = time.perf_counter()
                            for _ in range(4):
                                dbs[0][0].launch(True, True, stream=sa)
                                dbs[1][0].launch(True, True, stream=sb)
                            torch.cuda.synchronize(dev)
                            best = min(best, time.perf_counter() - t0)
                        return best
                    serial = span(cands[0], cands[0])
                    streams = [cands[0]]
                    for s_ in cands[1:]:
                        if len(streams) < 4 and all(span(k_, s_) < 0.8 * serial for k_ in streams):
                            streams.append(s_)
                    streams += [c_ for c_ in cands if c_ not in streams][:4 - len(streams)]
                for i in range(2 * B):
                    for g in range(G):
                        dbs[g][i % B].launch(True, True, stream=streams[g])
                torch.cuda.synchronize(dev)
                masks = [[db.new_hint_masks() for db in dbs[g]] for g in range(G)] if order == "masks" and hasattr(opt.lib, "dlscqp_mask_order_preview_device") else None
                for g in range(G):
                    for j, db in enumerate(dbs[g]):
                        if order == "iters" or (order == "masks" and masks is None):
                            db.set_hint([dbs[g][(j - k) % B].out_iters for k in (1, 2, 3) if (j - k) % B != j])
                        elif order == "masks":
                            db.set_hint_masks(masks[g][(j - 1) % B], masks[g][j], masks[g][(j + 1) % B])

                def step(c):
                    for g in range(G):
                        dbs[g][c % B].launch(True, True, stream=streams[g])
                for c in range(2 * B):
                    step(c)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for c in range(args.steps):
                    step(c)
                torch.cuda.synchronize(dev)
                out[G] = (time.perf_counter() - t0) / args.steps * 1e3
                if G == 1:
                    res_all = [db.results() for db in dbs[0]]
                del dbs
                for o in opts[1:]:
                    o.close()
            it = np.stack([r["iters"][:, 0] for r in res_all])
            x = np.stack([r["ctrl"] for r in res_all])
            st = np.stack([r["status"] for r in res_all])
            if ref is None:
                ref = (x, st)
            same = "same verdicts as the first library, max |ctrl diff| %.2e" % float(np.abs(x - ref[0])[(st == 0) & (ref[1] == 0)].max()) \
                if np.array_equal(st, ref[1]) else "VERDICTS DIFFER from the first library (%d)" % int((st != ref[1]).sum())
            hard = ""
            if not args.no_hard:
                res = res_all
                sel = [(j, a) for j in range(1, B) for a in np.nonzero((res[j]["iters"][:, 3] >= 5) & (res[j]["status"] == abi.OK))[0]]
                pick = [sel[i % len(sel)] for i in range(N)]
                ha = np.concatenate([hist[j]["agents"][a:a + 1] for j, a in pick])
                hr = np.concatenate([hist[j]["rows"][a * K:(a + 1) * K] for j, a in pick])
                hint = np.array([res[j - 1]["iters"][a, 0] for j, a in pick])
                hdb = opt.make_device_batch(ha, np.arange(N + 1) * K, hr, None, 0, want_iters=True)
                hdb.use_row_pitch()
                hdb.set_order(np.argsort(-hint, kind="stable").astype(np.int32))
                for _ in range(5):
                    hdb.launch(True, True)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(60):
                    hdb.launch(True, True)
                torch.cuda.synchronize(dev)
                hms = (time.perf_counter() - t0) / 60 * 1e3
                hit = hdb.results()["iters"]
                hard = " | hard %.4f ms (%.1f M), iters %.2f active %.2f" % (hms, N / hms / 1e3, hit[:, 0].mean(), hit[:, 3].mean())
                del hdb
            print("%-5s %-28s groups %.4f ms (%.1f M QP/s) | one launch %.4f ms (%.1f M)%s | iters mean %.2f max %d zero %.2f | %s" %
                  (order, os.path.basename(lib), out[4], N / out[4] / 1e3, out[1], N / out[1] / 1e3, hard, it.mean(), it.max(), (it == 0).mean(), same), flush=True)
            opt.close()


if __name__ == "__main__":
    main()
