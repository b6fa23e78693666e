"""How good a launch-order hint is: the last batch of a history of consecutive replanning steps, launched in the order of
(a) its own iteration counts (perfect foresight), (b) the previous step's, (c) the maximum / sum of the previous two or three
steps', (d) id order.   python tests/tools/hint_quality.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
B = 8
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000, keep=B, steps=B + 2)
hist = wl["history"]
dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
its = []
for d in dbs:
    d.launch()
    its.append(d.results()["iters"].copy())


def timed(db, order, n=30):
    db.set_order(order)
    for _ in range(3):
        db.launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        db.launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def srt(key):
    return np.argsort(-key, kind="stable").astype(np.int32)


i = [x[:, 0].astype(np.int64) for x in its]
print("%-8s %-10s %-10s %-12s %-14s %-14s %-14s" % ("target", "id order", "foresight", "previous", "decay 3/4", "decay 1/2", "max of all"))
for t in range(3, B):
    h34 = np.zeros_like(i[0]); h12 = np.zeros_like(i[0]); hmax = np.zeros_like(i[0])
    for k in range(t):  # the running state a closed loop would hold when it plans step t
        h34 = np.maximum(i[k], (3 * h34) // 4)
        h12 = np.maximum(i[k], h12 // 2)
        hmax = np.maximum(i[k], hmax)
    d = dbs[t]
    print("step %-3d %-10.4f %-10.4f %-12.4f %-14.4f %-14.4f %-14.4f" % (t, timed(d, None), timed(d, srt(i[t])), timed(d, srt(i[t - 1])),
                                                              timed(d, srt(h34)), timed(d, srt(h12)), timed(d, srt(hmax))))
opt.close()
