import sys
sys.path.insert(0, ".")
import numpy as np
from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000, keep=4)
for t, h in enumerate(wl["history"]):
    db = opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True)
    db.launch(); r = db.results()
    it, st = r["iters"][:, 0], r["status"]
    inf = st == abi.INFEASIBLE
    print("step %d: infeasible %.1f %%; iterations of infeasible agents: mean %.1f p50 %d p90 %d max %d; of feasible: mean %.2f p99 %d max %d; share of all iterations spent on infeasible agents %.0f %%" % (
        t, 100 * inf.mean(), it[inf].mean(), np.percentile(it[inf], 50), np.percentile(it[inf], 90), it[inf].max(), it[~inf].mean(), np.percentile(it[~inf], 99), it[~inf].max(), 100 * it[inf].sum() / it.sum()))
    print("   histogram of infeasible agents' iterations:", np.bincount(np.minimum(it[inf], 40))[:41].tolist())
opt.close()
