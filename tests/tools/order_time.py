"""Time dlscqp_launch_order_device and a launch in its order on the S64 batch.  python tests/tools/order_time.py [swarms]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

n_swarms = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=n_swarms, seed=2000)
db = opt.make_device_batch(wl["agents"], wl["row_offsets"], wl["rows"], None, 0, want_iters=True)


def timed(fn, n=50):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_id = timed(db.launch)
it = db.results()["iters"][:, 0]
db.set_order(np.argsort(-it, kind="stable").astype(np.int32))
t_host = timed(db.launch)
db.update_order()
t_dev = timed(db.launch)
t_ord = timed(db.update_order)
print("agents %d: launch ms  id order %.4f  host argsort %.4f  device order %.4f;  dlscqp_launch_order_device %.4f ms" %
      (len(it), t_id, t_host, t_dev, t_ord))
opt.close()
