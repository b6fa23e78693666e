"""Launch time of S64 batches made of one class of agents only: never iterating / exactly k iterations.  With rocprofv3 --pmc
SQ_INSTS_VALU the per-class instruction counts.   python tests/tools/class_batches.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000)
N = len(wl["agents"])
K = int(wl["row_offsets"][1] - wl["row_offsets"][0])
db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
db.launch()
res = db.results()
it = res["iters"][:, 0]


def timed(d, n=30):
    for _ in range(3):
        d.launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        d.launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("mixed batch: %.4f ms per %d agents (%.2f us of slot time per agent)" % (timed(db), N, timed(db) * 2048 / N * 1e3))
for lo, hi in ((0, 0), (1, 1), (2, 3), (4, 7), (8, 200)):
    sel = np.nonzero((it >= lo) & (it <= hi) & (res["status"] == abi.OK))[0]
    if len(sel) < 16:
        continue
    pick = sel[np.arange(N) % len(sel)]
    ha = wl["agents"][pick]
    hr = np.concatenate([wl["rows"][a * K:(a + 1) * K] for a in pick])
    d = opt.make_device_batch(ha, np.arange(N + 1) * K, hr, None, 0, want_iters=True)
    ms = timed(d)
    m = d.results()["iters"][:, 0].mean()
    print("agents with %d..%d iterations (%5.1f %% of the batch, mean %.2f): %.4f ms per %d agents = %.2f us of slot time per agent" %
          (lo, hi, 100 * len(sel) / N, m, ms, N, ms * 2048 / N * 1e3))
    del d
opt.close()
