#!/usr/bin/env python3
"""Developer tool: how much of a launch is lost to the launch ORDER, and which hint would lose less.

On the GPU (-DDLSCQP_TIMELINE build): the S64 workload's four consecutive batches are planned, every agent's solve time of
each batch is recorded with its iteration counts, and the arrays go to gpurun_out/order_sim.npz.  Off the GPU
(`--replay file`): a list-scheduling model of the 2 048 wave slots replays candidate launch orders on those times."""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc")
LIB = os.path.join(CSRC, "libdlscqp_timeline.so")


def record(swarms, out):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-DDLSCQP_TIMELINE", "-I" + os.path.join(ROOT, "include"), "-o", LIB]
                              + [os.path.join(CSRC, f) for f in ("dlscqp.hip", "qp_tables.cpp", "sfc_world.cpp", "grid_mapf.cpp")])
    os.environ["DLSCQP_LIB"] = LIB
    import torch
    from dlsc_gc_planner_amd import workloads
    from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
    cfg = workloads.s64_config()
    opt = BatchTrajOptimizer(cfg)
    wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=swarms, seed=2000, keep=4)
    durs, iters, spans = [], [], []
    for h in wl["history"]:
        db = opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True)
        N = len(h["agents"])
        stamps = torch.zeros((N, 12), dtype=torch.int64, device="cuda")
        db.launch()
        it = db.results()["iters"]
        order = torch.from_numpy(np.argsort(-it[:, 0], kind="stable").astype(np.int32)).cuda()
        db.c.agent_order = order.data_ptr()
        db.c.debug_stamps = stamps.data_ptr()
        for _ in range(3):
            db.launch()
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()
        durs.append((st[:, 1] - st[:, 0]) * 1e-2)
        spans.append((st[:, 1].max() - st[:, 0].min()) * 1e-2)
        iters.append(it.copy())
    np.savez(out, dur=np.stack(durs), iters=np.stack(iters), span=np.array(spans))
    print("recorded", out, "spans (own-count order)", spans)
    opt.close()


def makespan(dur, order, slots=2048, gap=0.85):
    """greedy list scheduling: the next agent of the order goes to the slot that frees first"""
    import heapq
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for a in order:
        t = heapq.heappop(free)
        t1 = t + gap + dur[a]
        end = max(end, t1)
        heapq.heappush(free, t1)
    return end


def replay(path):
    d = np.load(path)
    dur, iters = d["dur"], d["iters"]
    B, N = dur.shape
    print("recorded spans (own-count order):", d["span"])
    for j in range(1, B):
        it = [iters[(j - k) % B] for k in range(0, 4)]  # it[0] = this batch, it[1] = previous step, ...
        keys = {
            "id order": np.zeros(N),
            "own iterations (foresight)": it[0][:, 0],
            "own time (foresight)": dur[j],
            "previous step's iterations": it[1][:, 0],
            "max decay 3/4 over three steps": np.maximum(np.maximum(it[1][:, 0], 0.75 * it[2][:, 0]), 0.5625 * it[3][:, 0]) if j >= 3 else it[1][:, 0],
            "previous step's iterations x (8 + active rows)": it[1][:, 0] * (8.0 + it[1][:, 3]),
            "previous step's time": dur[j - 1],
            "max over three steps' time": np.maximum(np.maximum(dur[j - 1], dur[(j - 2) % B]), dur[(j - 3) % B]) if j >= 3 else dur[j - 1],
        }
        print("batch %d: ideal (sum / slots) %.1f us, longest agent %.1f us" % (j, (dur[j].sum() + 0.85 * N) / 2048, dur[j].max()))
        for name, key in keys.items():
            order = np.argsort(-key, kind="stable")
            print("   %-48s -> %.1f us" % (name, makespan(dur[j], order)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=512)
    ap.add_argument("--replay", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "order_sim.npz"))
    a = ap.parse_args()
    if a.replay:
        replay(a.replay)
    else:
        record(a.swarms, a.out)
