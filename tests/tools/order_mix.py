"""Launch-order experiment on the S64 batch: sorted by iteration count (the product) vs. orders that mix agents that iterate
with agents that do not, so that a SIMD's two wavefronts are more often of different kinds.  python tests/tools/order_mix.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000)
db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)
N = len(wl["agents"])


def timed(n=40):
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


db.launch()
it = db.results()["iters"][:, 0]
srt = np.argsort(-it, kind="stable").astype(np.int32)
print("id order            %.4f ms" % timed())
db.set_order(srt)
print("sorted descending   %.4f ms" % timed())
H, Z = srt[it[srt] > 0], srt[it[srt] == 0]
for head in (0, 2048, 4096, 8192):
    hh, rest = H[:head], H[head:]
    # after the head: the remaining iterating agents (still hardest first) spread evenly among the agents that never iterate
    out = list(hh)
    n_rest, n_z = len(rest), len(Z)
    pos = np.linspace(0, n_rest + n_z, n_rest, endpoint=False).astype(int) if n_rest else np.zeros(0, int)
    mix = np.full(n_rest + n_z, -1, np.int64)
    mix[pos] = rest
    mix[mix < 0] = Z
    order = np.concatenate([hh, mix]).astype(np.int32)
    assert np.array_equal(np.sort(order), np.arange(N))
    db.set_order(order)
    print("head %5d hardest, rest mixed evenly with the non-iterating   %.4f ms" % (head, timed()))
opt.close()
