"""Two-phase launch (dlscqp_device_batch.two_phase) against the one-phase launch on consecutive steps of the S64 loop: same
results, time per launch.   python tests/tools/two_phase_time.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dlsc_gc_planner_amd import workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

cfg = workloads.s64_config()
opt = BatchTrajOptimizer(cfg)
B = 5
wl = workloads.generate(cfg, lambda a, ro, r, s: opt.plan(a, ro, r, s, 0, True, True), n_swarms=512, seed=2000, keep=B)
hist = wl["history"]
dbs = [opt.make_device_batch(h["agents"], h["row_offsets"], h["rows"], None, 0, want_iters=True) for h in hist]


def timed(db, n=30):
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


prev = None
for t, db in enumerate(dbs):
    db.c.two_phase = 0
    db.set_order(None)
    db.launch()
    one = db.results()
    t_id = timed(db)
    t_hint = float("nan")
    if prev is not None:
        db.set_order(np.argsort(-prev, kind="stable").astype(np.int32))
        t_hint = timed(db)
    db.set_order(np.argsort(-one["iters"][:, 0], kind="stable").astype(np.int32))
    t_fore = timed(db)
    db.set_order(None)
    db.c.two_phase = 1
    db.launch()
    two = db.results()
    t_two = timed(db)
    same = all(np.array_equal(one[k], two[k]) for k in ("status", "goal_status", "goal", "ctrl", "cost", "iters"))
    print("step %d: one phase id order %.4f  previous step's counts %.4f  own counts %.4f | two phases %.4f ms   identical results: %s" %
          (t, t_id, t_hint, t_fore, t_two, same))
    if not same:
        for k in ("status", "goal_status", "goal", "ctrl", "cost", "iters"):
            if not np.array_equal(one[k], two[k]):
                bad = np.argwhere(np.asarray(one[k]) != np.asarray(two[k]))
                print("   differs:", k, len(bad), bad[:3].tolist(), np.asarray(one[k])[tuple(bad[0])], np.asarray(two[k])[tuple(bad[0])])
    prev = one["iters"][:, 0]
opt.close()
