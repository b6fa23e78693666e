#!/usr/bin/env python3
"""Developer tool: which pairs of freshly created streams really run side by side?  Two groups of the S64 loop on streams (i, j) of
eight created ones; a pair that shares a hardware queue shows about twice the step time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dlsc_gc_planner_amd import workloads  # noqa: E402
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer  # noqa: E402

dev = torch.device("cuda", 0)
cfg = workloads.s64_config(device=0)
opt0 = BatchTrajOptimizer(cfg)
B = 4
wl = workloads.generate(cfg, lambda a, ro, r, s: opt0.plan(a, ro, r, s, 0, True, True), n_swarms=256, swarm_size=64, seed=2000, keep=B)
hist, K, N = wl["history"], wl["K"], len(wl["agents"])
G = 2
per = N // G
opts = [opt0, BatchTrajOptimizer(cfg)]
dbs = [[opts[g].make_device_batch(h["agents"][g * per:(g + 1) * per], np.arange(per + 1) * K, h["rows"][g * per * K:(g + 1) * per * K], None, 0,
                                  want_iters=True) for h in hist] for g in range(G)]
streams = [torch.cuda.Stream(dev) for _ in range(8)]
for pair in ((0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (2, 3), (4, 5), (0, 0)):
    st = [streams[pair[0]], streams[pair[1]]]
    for rep in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for c in range(100):
            for g in range(G):
                dbs[g][c % B].launch(True, True, stream=st[g])
        torch.cuda.synchronize(dev)
    print(pair, "%.4f ms per step" % ((time.perf_counter() - t0) / 100 * 1e3), flush=True)
