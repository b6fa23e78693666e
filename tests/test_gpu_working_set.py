"""-m gpu: the kernel variant that keeps an agent's working set in a per-slot HBM buffer instead of LDS (WS_HBM: the fallback
for batches whose working set exceeds the 160 KiB of a CU -- since the solver keeps no orthogonal factor that takes a flag
array of > 60 000 rows per agent; the largest shape of the reference, the forest missions with four slack obstacles and up
to 124 active rows, needs 96 KB).  Forced with DLSCQP_DEBUG_WS=hbm on that shape: same bits as the LDS kernel of the same
(runtime-shape) instantiation, both equal to the golden batch."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fixtures

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r"""
import sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import numpy as np, torch
import fixtures
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
cfg, b, ans = fixtures.load("forest3d_M10_dyn4")
opt = BatchTrajOptimizer(cfg)
assert opt.lds_bytes(b["max_slack_obs"], int(np.diff(b["row_offsets"]).max())) <= 160 * 1024
got = opt.plan(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], True, True)
db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_iters=True)
for _ in range(3): db.launch()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): db.launch()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
it = db.results()["iters"][:, 0]
assert fixtures.check_against_golden(cfg, b, ans, got) >= 8
np.savez(sys.argv[1], ms=ms, iters=it, **{k: got[k] for k in ("status", "goal", "ctrl", "cost", "slack_cost", "goal_status")})
opt.close()
"""


def _run(tmp_path, tag, env_extra):
    out = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=os.path.dirname(HERE), here=HERE), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_working_set_in_lds_and_in_hbm_give_the_same_bits(gpu_required, tmp_path):
    a = _run(tmp_path, "lds", {"DLSCQP_FORCE_GENERIC": "1"})
    b = _run(tmp_path, "all_hbm", {"DLSCQP_FORCE_GENERIC": "1", "DLSCQP_DEBUG_WS": "hbm"})
    for k in ("status", "goal_status", "goal", "ctrl", "cost", "slack_cost", "iters"):
        assert np.array_equal(a[k], b[k]), k
    assert a["iters"].max() >= 5
    print("forest3d_M10_dyn4 (r = 124), %d agents per launch: working set in LDS %.3f ms, in HBM %.3f ms" %
          (len(a["status"]), float(a["ms"]), float(b["ms"])))
    assert float(a["ms"]) < float(b["ms"])
