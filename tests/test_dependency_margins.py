"""How far the kernel's dependency thresholds (csrc/qp_types.hpp: DEP_SUSPECT, DEP_SUSPECT_SLACK, DEP_NOISE, DEP_TOL) are from what
they separate -- measured, per random sweep, on the kernel's own source (lane-serial TRACE build, tests/emul).

Round 3's defect: the range-space kernel took rows that DEPEND on the active rows for independent ones (6 of 17 622 infeasible QPs
"solved").  The fix: a row whose step rate a.dx is below DEP_SUSPECT |N'a|^2 is measured again in the whitened space, |z|^2 against
DEP_TOL |N'a|^2 and against DEP_NOISE x the scale of its own rounding noise.  Those four constants were calibrated on batches at
the launch files' parameter values.  Here a trace build measures EVERY candidate row of every iteration in the whitened space as
well (the truth), over random batches that include the corners of the parameter box (control_input_weight 1 and 1e-4, dt 0.1 and
0.4, terminal_weight 20, slack_collision_weight 1), and the test asserts
  * no row that is decisively dependent by the whitened measure was taken for independent (the defect itself) -- on every seed,
  * every dependent row's first measure a.dx / |N'a|^2 is at least 100 x below the suspicion threshold of its shape,
  * in the whitened measure dependent rows are NOISE_BAR x below DEP_NOISE, or far below DEP_TOL, or measured precisely enough
    that their distance from DEP_TOL is 100 x their own uncertainty ... and independent ones as far above -- on the PINNED seeds;
    on the run's fresh seed a shortfall is a warning that names the seed (the quantities are continuous: see PINNED below).
The seed comes from the run (printed; FUZZ_SEED replays it)."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

from dlsc_gc_planner_amd import abi
import emul_lib
import randqp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEP_SUSPECT, DEP_SUSPECT_SLACK, DEP_NOISE, DEP_TOL, DEP_XSPACE = 1e-5, 1e-2, 1e-17, 1e-20, 1e-16  # csrc/qp_types.hpp
EPS = 2.2e-16
SEED = int(os.environ.get("FUZZ_SEED", str(int(time.time()) % 1000000)))
NAMES = ["empty3d_M5", "maze2d_M10_sfc_comm", "forest3d_M5_dyn", "crowd3d_M5_400rows", "long3d_M12_comm"] + randqp.OFF_DEFAULT
# The one configuration found so far in which the margins ARE eroded: control_input_weight 1e-4 together with dt 0.1 at M = 10 with
# slack variables (the reduced Hessian's condition number grows with 1 / dt^5 and 1 / weight).  Over seeds 100-140 a dependent row's
# first measure came within 2.6 x of the suspicion threshold and an independent row's |z|^2 / noise within 1.2 x of DEP_NOISE; every
# row was still classified correctly and the sweeps' verdicts equal the oracle's.  There the test asserts the classification only
# and REPORTS the margins (DESIGN.md section 9g says so): the thresholds are constants, not functions of the conditioning.
THIN = {"forest3d_M10_dyn_wc1e-4_dt01"}


@pytest.fixture(scope="module")
def trace_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "libdlscqp_emul_trace.so"])
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libdlscqp_emul_trace.so"))
    vp = C.c_void_p
    L.emul_plan_batch.argtypes = [C.POINTER(abi.Config), C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.emul_trace.argtypes = [C.c_int, vp, C.c_int]
    return L


def traced_records(L, cfg, b):
    """the (tag, a, b, c, d) records of a whole batch planned by the trace build"""
    old = emul_lib._LIB
    emul_lib._LIB = L
    try:
        L.emul_trace(1, None, 0)
        out = emul_lib.plan_batch(cfg, b["agents"], b["row_offsets"], b["rows"], b["sfc"], b["max_slack_obs"])
        cap = 40_000_000
        buf = np.zeros(cap)
        n = L.emul_trace(2, buf.ctypes.data_as(C.c_void_p), cap)
        L.emul_trace(0, None, 0)
    finally:
        emul_lib._LIB = old
    assert n < cap
    return out, buf[:n - n % 5].reshape(-1, 5)


# Seeds whose batches held the narrowest margins found so far (each was a fresh seed once): the margin assertions are binding on
# these.  On the run's own fresh seed only the classification is binding -- a margin below the bar there is printed and raised as
# a warning, with the seed, so that it can be looked at and pinned: the quantities are continuous, a bar that no seed ever crosses
# does not exist (220 consecutive seeds x 13 configurations in round 4: narrowest 86 x to DEP_NOISE, 50 x to DEP_XSPACE).
# (Round 5's second sweep, seeds 80000-80199: seed 80048 held the one MISCLASSIFICATION found so far -- after a pivot of 1.3e-12 |N'a|^2
# a dependent row's a.dx came out at 0.12 |N'a|^2, above any constant; DEP_PIVOT of csrc/qp_types.hpp is the answer, and
# test_a_dependent_row_behind_a_tiny_pivot_is_recognised below pins it.  The same seed has an independent row 2.5 x above DEP_XSPACE
# and a dependent one 2.9 x below: its margin bars cannot be made binding.)
PINNED = [4242, 62588, 127, 68727, 68740, 68821]
NOISE_BAR, XSPACE_BAR = 30.0, 10.0  # DEP_XSPACE sits in the middle of a gap that is 1 000 x wide (exact duplicates of an active
# row: noise^2 / |N'a|^2 <= 4e-18; duplicates up to float32 rounding of the normal: >= 5e-15), so 10 x is what there is


def _bar(ok, binding, what):
    if ok:
        return
    if binding:
        raise AssertionError(what)
    import warnings
    warnings.warn("dependency margin below the bar on a fresh seed (replay with FUZZ_SEED): %r" % (what,))


@pytest.mark.parametrize("seed", PINNED + [SEED], ids=lambda s: "seed%d" % s)
@pytest.mark.parametrize("name", NAMES)
def test_dependency_thresholds_keep_their_margins(trace_lib, name, seed, record_property):
    binding = seed in PINNED or "FUZZ_SEED" in os.environ
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    n = 160 if "M10" not in name and "M12" not in name else 60
    gk = dict(gk, infeasible_frac=max(0.3, gk.get("infeasible_frac", 0.1)))  # dependent rows live in infeasible problems
    b = randqp.random_batch(cfg, n, seed=seed, **gk)
    out, rec = traced_records(trace_lib, cfg, b)
    r6 = rec[rec[:, 0] == 6]
    r7 = rec[rec[:, 0] == 7]  # noise^2 / |N'a|^2 (what DEP_XSPACE is compared with), one record per record 6
    assert len(r6) > 0 and len(r7) == len(r6)
    ux = r7[:, 1]
    adx, zz, zn, fl = r6[:, 1], r6[:, 2], r6[:, 3], r6[:, 4].astype(np.int64)
    taken_dep, remeasured, truth, static = (fl & 1) != 0, (fl & 2) != 0, (fl & 4) != 0, (fl & 8) != 0
    suspect = np.where(static, DEP_SUSPECT, DEP_SUSPECT_SLACK)
    line = "seed %d  %-32s rows measured %6d, dependent %5d" % (seed, name, len(r6), int(truth.sum()))
    # 1. the defect itself: dependent by the whitened measure, taken for independent.  Only a row that the whitened measure calls
    # dependent WITH the bar to spare counts: the measure has a noise floor of its own (DEP_NOISE x its noise scale), and where an
    # ill-conditioned active set lifts that floor to 4e-3 |N'a|^2 (seed 68885, a row with slack variable: first measure 7e-3, whitened
    # 3.5e-4 +- 4e-3) "dependent" is not something it can say.  Such rows are counted and printed; the sweeps' verdicts on those
    # batches equal the oracle's.
    decisive = truth & ((zz <= DEP_TOL) | (ux <= DEP_XSPACE) | ((zn >= 0) & (zn <= DEP_NOISE / NOISE_BAR)))
    assert not (decisive & ~taken_dep).any(), (seed, name, r6[decisive & ~taken_dep][:5])
    undecided = int((truth & ~decisive & ~taken_dep).sum())
    if undecided:
        line += ", %d below the whitened measure's noise floor and taken for independent" % undecided
    # ... and nothing independent is thrown away: a row the solve calls dependent is dependent by the whitened measure
    assert not (taken_dep & ~truth).any(), (seed, name, r6[taken_dep & ~truth][:5])
    if truth.any():
        first = decisive if decisive.any() else truth
        worst_first = float((adx[first] / suspect[first]).max())  # how close a dependent row's first measure comes to the suspicion threshold
        dep_noise = float(zn[truth & (zn >= 0)].max()) if (truth & (zn >= 0)).any() else 0.0
        dep_tol = float(zz[truth].max())
        line += "; dependent rows: a.dx/|N'a|^2 at most %.1e of the suspicion threshold, |z|^2/noise <= %.1e, |z|^2/|N'a|^2 <= %.1e" % (worst_first, dep_noise, dep_tol)
        _bar(worst_first <= 1e-2 or name in THIN, binding, (seed, name, "first measure", worst_first))
        # dependent by at least one of the tests with 100 x to spare: rounding noise in the whitened space, |z|^2 far below DEP_TOL, or
        # nothing left of the normal even in absolute values (round 4: |z|^2 / |N'a|^2 of an exact duplicate comes out on both sides
        # of DEP_TOL -- 4.5e-21 at seed 62588, 5.8e-20 at seed 127 -- which is what DEP_XSPACE was added for)
        # (a row that DEP_XSPACE decides is reported with |z|^2 = 0: its margin is that of noise^2 / |N'a|^2, not DEP_TOL's)
        # (seeds 68727, 68803 ...: two rows that differ by float32 rounding of the same normal -- u = a - A'r cancels to 7e-8 of a, the
        # whitened length comes out at 4.3e-21 of |N'a|^2, 2.3 x below DEP_TOL, and is MEASURED to eleven digits: its noise scale
        # is 5e-15 |N'a|^2 x eps.  Nearly-dependent rows form a continuum, no constant keeps 100 x away from all of them; what has
        # to hold is that the measurement's own uncertainty is 100 x smaller than its distance from the threshold it is compared
        # with -- then this solver and the oracle, which measures the same quantity another way, read it the same.)
        precise = (ux[truth] > DEP_XSPACE) & (zz[truth] <= DEP_TOL) & (DEP_TOL - zz[truth] >= 100 * EPS * ux[truth])
        spare = (zn[truth] <= DEP_NOISE / NOISE_BAR) | (ux[truth] <= DEP_XSPACE / XSPACE_BAR) | ((zz[truth] <= DEP_TOL * 1e-2) & (ux[truth] > DEP_XSPACE)) | precise
        line += ", noise^2/|N'a|^2 <= %.1e where that decides" % float(ux[truth][ux[truth] <= DEP_XSPACE].max() if (ux[truth] <= DEP_XSPACE).any() else 0.0)
        _bar(spare.all() or name in THIN, binding, (seed, name, "dependent rows", r6[truth][~spare][:5], ux[truth][~spare][:5]))
    ind = ~truth & (zn >= 0)
    if ind.any():
        ind_noise, ind_tol = float(zn[ind].min()), float(zz[ind].min())
        line += "; independent rows: |z|^2/noise >= %.1e, |z|^2/|N'a|^2 >= %.1e" % (ind_noise, ind_tol)
        clear = (zz[ind] >= DEP_TOL * 1e2) | (zz[ind] - DEP_TOL >= 100 * EPS * ux[ind])  # as above, from the other side
        _bar((ind_noise >= DEP_NOISE * NOISE_BAR and clear.all()) or name in THIN, binding, (seed, name, "independent rows", ind_noise, ind_tol))
        ind_x = float(ux[~truth].min())
        line += ", noise^2/|N'a|^2 >= %.1e" % ind_x
        _bar(ind_x >= DEP_XSPACE * XSPACE_BAR, binding, (seed, name, "independent rows, DEP_XSPACE", ind_x))
    print(line)
    record_property("margins", line)
    log = os.environ.get("DEP_MARGIN_LOG")
    if log:
        with open(log, "a") as fh:
            fh.write(line + "\n")


def test_a_dependent_row_behind_a_tiny_pivot_is_recognised(trace_lib):
    """Seed 80048 of the thin corner (control_input_weight 1e-4, dt 0.1, M = 10, slack variables): one infeasible agent accepts a row with
    |z|^2 = 1.3e-12 |N'a|^2; from then on the first measure of rows that depend on the active ones is garbage (0.12 |N'a|^2).  With
    DEP_PIVOT every row behind such a pivot is measured in the whitened space: no dependent row is taken for an independent one, and the
    agent reaches its verdict in a fraction of the iterations (285 without the rule)."""
    name, seed = "forest3d_M10_dyn_wc1e-4_dt01", 80048
    ck, gk = randqp.CONFIGS[name]
    cfg = abi.make_config(**ck)
    gk = dict(gk, infeasible_frac=max(0.3, gk.get("infeasible_frac", 0.1)))
    b = randqp.random_batch(cfg, 60, seed=seed, **gk)
    out, rec = traced_records(trace_lib, cfg, b)
    r6, r7 = rec[rec[:, 0] == 6], rec[rec[:, 0] == 7]
    fl = r6[:, 4].astype(np.int64)
    taken_dep, truth = (fl & 1) != 0, (fl & 4) != 0
    zz, zn, ux = r6[:, 2], r6[:, 3], r7[:, 1]
    decisive = truth & ((zz <= DEP_TOL) | (ux <= DEP_XSPACE) | ((zn >= 0) & (zn <= DEP_NOISE / NOISE_BAR)))
    assert r7[:, 3].min() < 1e-9  # the batch does hold the tiny pivot
    assert not (decisive & ~taken_dep).any() and not (taken_dep & ~truth).any()
    assert out["iters"][:, 0].max() < 150
