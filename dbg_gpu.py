import sys; sys.path.insert(0,'tests')
import numpy as np
from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import fixtures, oracle_lib as O
cfg,b,ans = fixtures.load("maze2d_M10_sfc_comm")
opt = BatchTrajOptimizer(cfg)
db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_f64=True, want_iters=True)
for rep in range(3):
    db.launch(); got = db.results()
    ref = O.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True)
    bad = np.where(got['status'] != ref['status'])[0]
    print("rep", rep, "mismatch agents", bad, got['status'][bad], ref['status'][bad], got['iters'][bad])
    ok = (got['status']==0)&(ref['status']==0)
    print("  max x err", np.abs(got['x'][ok]-ref['x'][ok]).max(), "iters", got['iters'][:,0])
