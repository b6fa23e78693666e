#!/usr/bin/env python3
"""Developer tool: builds of libdlscqp.so on a batch made ONLY of agents that never iterate (every wavefront the same work: no
tail, no launch order -- what is left is the cost of the prologue, the screen and the read-out), and on bench.py's rotation in id order.
    DLSCQP_NO_SELFTEST=1 python tests/tools/zero_ab.py --libs a.so,b.so"""
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
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    from dlsc_gc_planner_amd import abi, workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    dev = torch.device("cuda", 0)
    cfg = workloads.s64_config(device=0)
    libs = [l for l in args.libs.split(",") if l] or [abi.library_path()]
    wl = None
    for rep in range(args.reps):
        for lib in libs:
            abi._LIB = None
            lib, *envs = lib.split("@")  # path@VAR=VALUE@...: environment the library reads at its first launch (a COPY of the file per setting: statics)
            for k in [k for k in os.environ if k.startswith("DLSCQP_") and k not in ("DLSCQP_NO_SELFTEST",)]:
                del os.environ[k]
            for e in envs:
                os.environ[e.split("=")[0]] = e.split("=")[1]
            os.environ["DLSCQP_LIB"] = os.path.abspath(lib)
            opt = BatchTrajOptimizer(cfg)
            if wl is None:
                wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=args.swarms, swarm_size=64, seed=2000, keep=4)
                hist = wl["history"]
                N, K = len(wl["agents"]), wl["K"]
                its = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
                for db in its:
                    db.launch(True, True)
                iters = [db.results()["iters"] for db in its]
                del its
                sel = [(j, a) for j in range(4) for a in np.nonzero(iters[j][:, 0] == 0)[0]]
                pick = [sel[i % len(sel)] for i in range(N)]
                za = np.concatenate([hist[j]["agents"][a:a + 1] for j, a in pick])
                zr = np.concatenate([hist[j]["rows"][a * K:(a + 1) * K] for j, a in pick])
            dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
            zdb = opt.make_device_batch(za, np.arange(N + 1) * K, zr, None, 0, want_iters=True)
            for db in dbs + [zdb]:
                db.use_row_pitch()
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
            print("%-28s zero-only batch %.4f ms (%.0f M/s)   rotation, id order, one launch %.4f ms" % (os.path.basename(lib), out["zero"], N / out["zero"] / 1e3, out["rotation"]), flush=True)
            del dbs, zdb
            opt.close()


if __name__ == "__main__":
    main()
