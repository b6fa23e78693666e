"""How well what the FIRST pricing pass sees predicts an agent's cost (diagnostic library that writes the number of violated
rows and the best pivot score of the first pass into out_iters[2] / [1]):
    DLSCQP_LIB=build/lib_fs.so python tests/tools/first_scan_predictor.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
B = 5
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000, keep=B)
hist = wl["history"]
dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]
res = []
for d in dbs:
    d.launch()
    res.append(d.results())


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
    return np.argsort(-np.asarray(key, np.float64), kind="stable").astype(np.int32)


for t in range(1, B):
    r = res[t]["iters"]
    it, top, cnt, prev = r[:, 0], r[:, 1], r[:, 2], res[t - 1]["iters"][:, 0]
    cc = np.corrcoef(cnt, it)[0, 1]
    print("step %d: id %.4f  previous step %.4f  foresight %.4f | first pass: violated-row count %.4f (corr with iterations %.2f), "
          "count then previous %.4f, max(count, previous) %.4f, best score %.4f" % (
              t, timed(dbs[t], None), timed(dbs[t], srt(prev)), timed(dbs[t], srt(it)), timed(dbs[t], srt(cnt)), cc,
              timed(dbs[t], srt(cnt * 100.0 + prev)), timed(dbs[t], srt(np.maximum(cnt, prev))), timed(dbs[t], srt(top))))
opt.close()
