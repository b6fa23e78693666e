import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import fixtures
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
name = sys.argv[1] if len(sys.argv) > 1 else "forest3d_M10_dyn4"
cfg, b, ans = fixtures.load(name)
opt = BatchTrajOptimizer(cfg)
got = opt.plan(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True, True)
ok = fixtures.check_against_golden(cfg, b, ans, got)
db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_iters=True)
for _ in range(3): db.launch()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): db.launch()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print("%s golden ok=%d  %.3f ms per launch" % (name, ok, ms))
opt.close()
