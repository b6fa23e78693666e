"""-m gpu: the instantiations that exist twice -- two wavefronts per SIMD for large launches, one wavefront per SIMD (whole
register file) for launches of at most 1 024 agents (dlscqp.hip: WIDE) -- give the same bits.  A golden batch is tiled to
1 500 agents and planned once as a whole (large-launch build) and once in two halves (small-launch build)."""
import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
import fixtures

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["crowd3d_M5_400rows", "forest3d_M5_dyn", "lsc2d_M4_nostop"])
def test_small_and_large_launch_builds_agree(gpu_required, name):
    cfg, b, ans = fixtures.load(name)
    n0 = len(b["agents"])
    reps = -(-1500 // n0)
    ro = b["row_offsets"].astype(np.int64)
    agents = np.tile(b["agents"], reps)
    rows = np.concatenate([b["rows"]] * reps)
    offs = np.concatenate([[0], np.cumsum(np.tile(np.diff(ro), reps))]).astype(np.int64)
    sfc = None if b["sfc"] is None else np.concatenate([b["sfc"]] * reps)
    N = len(agents)
    assert N > 1024
    opt = BatchTrajOptimizer(cfg)
    try:
        db = opt.make_device_batch(agents, offs, rows, sfc, b["max_slack_obs"], want_f64=True, want_iters=True)
        db.launch()
        whole = db.results()
        for k in ("out_status", "out_ctrl", "out_cost", "out_iters"):  # poison what the halves must rewrite
            getattr(db, k).fill_(-7)
        db.out_x.zero_()  # (an agent writes only its own slack columns of x: the rest stays as allocated)
        half = N // 2
        db.launch(offset=0, count=half)
        db.launch(offset=half, count=N - half)
        halves = db.results()
    finally:
        opt.close()
    for k in ("status", "goal_status", "goal", "iters"):
        assert np.array_equal(whole[k], halves[k]), k
    ok = whole["status"] == abi.OK  # a failed agent's plan is not written at all
    assert ok.sum() > N // 4
    for k in ("ctrl", "x", "cost", "slack_cost"):
        assert np.array_equal(whole[k][ok], halves[k][ok]), k
    # and both are the golden answers
    first = {k: v[:n0] for k, v in whole.items()}
    assert fixtures.check_against_golden(cfg, b, ans, first) >= 5
    assert (whole["status"][:n0] == whole["status"][n0:2 * n0]).all()


SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import numpy as np
import fixtures
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer
cfg, b, ans = fixtures.load(%(name)r)
opt = BatchTrajOptimizer(cfg)
db = opt.make_device_batch(b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"], want_f64=True, want_iters=True)
db.launch()
got = db.results()
assert fixtures.check_against_golden(cfg, b, ans, got) >= 5
np.savez(sys.argv[1], **{k: got[k] for k in ("status", "goal_status", "goal", "ctrl", "x", "cost", "iters")})
opt.close()
"""


@pytest.mark.parametrize("name", ["crowd3d_M5_400rows", "maze2d_M10_sfc_comm", "forest3d_M10_dyn4", "long3d_M12_comm"])
def test_helper_wavefronts_change_nothing(gpu_required, tmp_path, name):
    """Small launches of agents with rows beyond the register cache run with four wavefronts per agent (dlscqp.hip: WPA): three
    helpers price their share of those rows in every pricing pass.  Same rows, same arithmetic, same maximum: forced on
    (DLSCQP_WPA=4) and off (=1), the answers are the same bits -- and the golden ones."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    # ... and without helpers the rows right behind the register cache are decoded once into LDS and priced from there (round 4,
    # gi_core.hpp: "rows in LDS"; DLSCQP_ROWS_IN_LDS=0: from HBM in every pass, as the helpers price theirs): the same bits again
    # (round 5: the helpers price their shares of the rows in LDS as well; "4-hbm": helpers with every row from HBM, as before)
    for tag, wpa, in_lds in (("1", "1", "1"), ("1-hbm", "1", "0"), ("4", "4", "1"), ("4-hbm", "4", "0")):
        out = str(tmp_path / ("wpa%s.npz" % tag))
        env = dict(os.environ, DLSCQP_WPA=wpa, DLSCQP_ROWS_IN_LDS=in_lds)
        r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=os.path.dirname(here), here=here, name=name), out], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(out)
    assert res["1"]["iters"][:, 0].max() >= 2
    for k in ("status", "goal_status", "goal", "ctrl", "x", "cost", "iters"):
        assert np.array_equal(res["1"][k], res["4"][k]), k
        assert np.array_equal(res["1"][k], res["1-hbm"][k]), k
        assert np.array_equal(res["1"][k], res["4-hbm"][k]), k
