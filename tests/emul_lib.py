"""ctypes access to tests/emul/libdlscqp_emul.so -- TEST INFRASTRUCTURE ONLY.

The lane-serial host build of the kernel source (tests/emul/emul.cpp).  It lets the CPU-only
test tier check the device algorithm against the oracle.  Never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from dlsc_gc_planner_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")])
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_emul.so"))
    vp = C.c_void_p
    L.emul_plan_batch.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                  vp, vp, vp, vp, vp, vp, vp, vp]
    L.emul_plan_batch.restype = C.c_int
    L.emul_table_info.argtypes = [C.POINTER(abi.Config), vp, vp, vp, vp, C.c_int, C.c_int]
    L.emul_table_info.restype = C.c_int
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def plan_batch(cfg, agents, row_offsets, rows, sfc=None, max_slack_obs=0, do_goal=True, do_traj=True):
    L = lib()
    agents = np.ascontiguousarray(agents, dtype=abi.AGENT_DTYPE)
    rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
    row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int64)
    sfc = None if sfc is None else np.ascontiguousarray(sfc, dtype=np.float32)
    N = len(agents)
    M, n, dim = cfg.M, cfg.n, cfg.world_dimension
    nx = dim * M * (n + 1) + M * max_slack_obs
    out = dict(goal=np.zeros((N, 3), np.float32), goal_status=np.zeros(N, np.int32),
               ctrl=np.zeros((N, M, n + 1, 3), np.float32), x=np.zeros((N, nx)), cost=np.zeros(N),
               slack_cost=np.zeros((N, max(max_slack_obs, 1))), status=np.full(N, -1, np.int32),
               iters=np.zeros((N, 4), np.int32))
    rc = L.emul_plan_batch(C.byref(cfg), N, _ptr(agents), _ptr(row_offsets), _ptr(rows), _ptr(sfc),
                           max_slack_obs, int(do_goal), int(do_traj), _ptr(out["goal"]),
                           _ptr(out["goal_status"]), _ptr(out["ctrl"]), _ptr(out["x"]), _ptr(out["cost"]),
                           _ptr(out["slack_cost"]) if max_slack_obs else None, _ptr(out["status"]), _ptr(out["iters"]))
    if rc != 0:
        raise ValueError("emul_plan_batch: invalid configuration")
    return out


def table_info(cfg, max_slack_obs=0, kmax=0):
    L = lib()
    v = [C.c_int() for _ in range(4)]
    rc = L.emul_table_info(C.byref(cfg), *[C.byref(t) for t in v], max_slack_obs, kmax)
    if rc != 0:
        raise ValueError("invalid configuration")
    return dict(f=v[0].value, n_slots=v[1].value, nb=v[2].value, lds_bytes=v[3].value)
