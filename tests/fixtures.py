"""Loading of the committed golden fixtures (tests/golden/*.npz, kat_log.json)."""
import json
import os

import numpy as np

from dlsc_gc_planner_amd import abi, workloads

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

NAMES = ["s64", "empty3d_M5", "maze2d_M10_sfc_comm", "forest3d_M10_dyn", "forest3d_M10_dyn4", "forest3d_M5_dyn", "lsc2d_M4_nostop",
         "rsfc3d_M5_allslack", "crowd3d_M5_400rows", "long3d_M12_comm", "empty3d_M5_phin2", "maze2d_M10_phin3"]


def load(name):
    z = np.load(os.path.join(GOLDEN, "golden_%s.npz" % name), allow_pickle=False)
    ck = json.loads(str(z["cfg_kwargs"]))
    if ck.get("workload") == "s64":
        cfg = workloads.s64_config()
    else:
        for k in ("world_min", "world_max"):
            if k in ck:
                ck[k] = tuple(ck[k])
        cfg = abi.make_config(**ck)
    sfc = z["sfc"]
    batch = dict(agents=z["agents"].astype(abi.AGENT_DTYPE), row_offsets=z["row_offsets"], rows=z["rows"].astype(abi.ROW_DTYPE),
                 sfc=None if sfc.size == 0 else sfc, max_slack_obs=int(z["max_slack_obs"]))
    ans = dict(goal_t=z["goal_t"], goal_feasible=z["goal_feasible"], x=z["x"], qp_status=z["qp_status"], qp_obj=z["qp_obj"],
               x_source=z["x_source"])
    return cfg, batch, ans


def kat():
    return json.load(open(os.path.join(GOLDEN, "kat_log.json")))


def check_against_golden(cfg, batch, ans, got, x_key="x", x_tol=5e-6):
    """Shared assertions: verdicts identical to the independent solver, primal solution within x_tol
    (HiGHS' own accuracy at 1e-9 feasibility tolerances is ~1e-7 on the 90-variable shapes, 2e-6 on the 220-variable one)."""
    N = len(batch["agents"])
    reached = ans["qp_status"] >= 0
    # goal LP verdicts: infeasible LP <-> INFEASIBLE or the numerical-fallback branch
    g_inf = (got["goal_status"] == abi.INFEASIBLE) | (got["goal_status"] == abi.GOAL_FALLBACK)
    assert np.array_equal(g_inf, ~ans["goal_feasible"]), (g_inf, ans["goal_feasible"])
    # agents whose LP was infeasible and not rescued never reach the QP
    assert np.array_equal(got["goal_status"] == abi.INFEASIBLE, ~reached)
    st = got["status"][reached]
    assert np.array_equal(st != abi.OK, ans["qp_status"][reached] != abi.OK)
    assert set(np.unique(st)).issubset({abi.OK, abi.INFEASIBLE})
    ok = reached & (ans["qp_status"] == abi.OK)
    if x_key in got and ok.any():
        nx = got[x_key].shape[1]
        err = np.abs(got[x_key][ok] - ans["x"][ok, :nx]).max()
        assert err < x_tol, err
    if "cost" in got and ok.any():
        assert np.allclose(got["cost"][ok], ans["qp_obj"][ok], rtol=1e-6, atol=1e-7)
    # float32 control points: 1e-4 relative (north star), in practice rounding only
    if ok.any():
        dim, M, n = cfg.world_dimension, cfg.M, cfg.n
        xc = ans["x"][ok, :dim * M * (n + 1)].reshape(-1, dim, M, n + 1).transpose(0, 2, 3, 1)
        got_c = got["ctrl"][ok][..., :dim].astype(np.float64)
        scale = np.maximum(np.abs(xc).max(), 1.0)
        assert np.abs(got_c - xc).max() <= 1e-4 * scale
        if dim == 2:
            assert np.all(got["ctrl"][ok][..., 2] == np.float32(cfg.world_z_2d))
    return int(ok.sum())
