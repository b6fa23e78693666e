#!/usr/bin/env python3
"""Study (CPU, trace build of the kernel's source): does the rounding noise of a dependent row's FIRST measure a.dx / |N'a|^2 -- what
DEP_SUSPECT* has to stay above -- follow the conditioning?  Per configuration of tests/test_dependency_margins.py: the largest first
measure of a truly dependent row, the smallest pivot ratio |z|^2 / |N'a|^2 a row had been added with at that moment (rho_min), the
condition number kappa of the Hessian (reduced Hessian and slack block, csrc/qp_tables.cpp: hr_cond), and the measure in units of
eps kappa / rho_min.     python tests/tools/dep_threshold_study.py 4242,62588,127 [config,config...]      (run from tests/)
Result: no.  kappa stays within 8e3 .. 5e7 over all configurations while the noise spans 1e-21 .. 8e-4, and the
worst case (seed 70162: 8.3e-4 at kappa 2.4e5, rho_min 1.6e-3) is 2.5e4 x the model -- it comes from the SIZE of the numbers in an
infeasible problem a few iterations before its verdict, which is what the whitened re-measurement's own noise scale measures."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C
from dlsc_gc_planner_amd import abi
import emul_lib, randqp
import test_dependency_margins as T
L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_emul_trace.so"))
vp = C.c_void_p
L.emul_plan_batch.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
L.emul_trace.argtypes = [C.c_int, vp, C.c_int]
EPS = 2.2e-16
seeds = [int(s) for s in sys.argv[1].split(",")]
names = T.NAMES if len(sys.argv) < 3 else sys.argv[2].split(",")
Lc = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_emul.so"))
def kappa(cfg, slack):
    M = cfg.M
    out = np.zeros(4 * M)
    Lc.emul_table_cond(C.byref(cfg), out.ctypes.data_as(C.c_void_p))
    lmax, lmin, sig = out[M:2*M], out[2*M:3*M], out[3*M:4*M]
    hs = 1.0 / np.maximum(sig, 1e-300) ** 2
    return max(lmax.max(), hs.max()) / min(lmin.min(), hs.min()) if slack else lmax.max() / lmin.min()
for name in names:
    worst = (0, None); rows = 0; dep = 0; lo_ind = 1e300
    allr = []
    for seed in seeds:
        ck, gk = randqp.CONFIGS[name]
        cfg = abi.make_config(**ck)
        n = 160 if "M10" not in name and "M12" not in name else 60
        gk = dict(gk, infeasible_frac=max(0.3, gk.get("infeasible_frac", 0.1)))
        b = randqp.random_batch(cfg, n, seed=seed, **gk)
        out, rec = T.traced_records(L, cfg, b)
        r6 = rec[rec[:, 0] == 6]; r7 = rec[rec[:, 0] == 7]
        adx, fl = r6[:, 1], r6[:, 4].astype(np.int64)
        truth = (fl & 4) != 0
        rho = r7[:, 3]
        rows += len(r6); dep += int(truth.sum())
        if truth.any():
            kap = kappa(cfg, b["max_slack_obs"] > 0)
            q = adx[truth] * np.minimum(rho[truth], 1.0) / (EPS * kap)   # first measure in units of eps kappa / rho_min
            allr.append(np.stack([adx[truth], rho[truth], q], 1))
        ind = ~truth
        if ind.any(): lo_ind = min(lo_ind, float(adx[ind].min()))
    if allr:
        a = np.concatenate(allr)
        i = int(np.argmax(a[:, 2])); j = int(np.argmax(a[:, 0]))
        print("%-34s rows %7d dependent %6d: largest first measure %.1e (rho_min %.1e); largest in units of eps kappa/rho_min %.1e (m1 %.1e rho %.1e); smallest rho_min %.1e; smallest independent first measure %.1e"
              % (name, rows, dep, a[j, 0], a[j, 1], a[i, 2], a[i, 0], a[i, 1], a[:, 1].min(), lo_ind), flush=True)
