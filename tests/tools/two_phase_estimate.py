"""What a two-phase launch could gain: time the agents that iterate (or fail) alone, in the order of the previous step's hint,
and the agents that never iterate alone.   python tests/tools/two_phase_estimate.py"""
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
    return np.argsort(-key, kind="stable").astype(np.int32)


K = int(hist[0]["row_offsets"][1])
for t in range(1, B):
    it, prev = res[t]["iters"][:, 0], res[t - 1]["iters"][:, 0]
    hard = np.nonzero((it > 0) | (res[t]["status"] != abi.OK))[0]
    easy = np.nonzero((it == 0) & (res[t]["status"] == abi.OK))[0]
    h = hist[t]

    def sub(sel):
        rows = np.concatenate([h["rows"][a * K:(a + 1) * K] for a in sel])
        return opt.make_device_batch(h["agents"][sel], np.arange(len(sel) + 1) * K, rows, None, 0, want_iters=True)
    dh, de = sub(hard), sub(easy)
    print("step %d: all %d agents, hint order %.4f ms (foresight %.4f);  the %d that iterate or fail alone: hint order %.4f ms, foresight %.4f ms;"
          "  the %d that never iterate alone: %.4f ms" % (t, len(it), timed(dbs[t], srt(prev)), timed(dbs[t], srt(it)), len(hard),
                                                           timed(dh, srt(prev[hard])), timed(dh, srt(it[hard])), len(easy), timed(de, None)))
    del dh, de
opt.close()
