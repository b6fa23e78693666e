"""Checks of an infeasibility certificate (dlscqp_device_batch.out_conflict) with plain numpy -- test infrastructure.

No solver and no oracle: the certificate's members, decoded into rows of the reference's model by the product's host code
(dlscqp_conflict_decode, csrc/model_rows.hpp), are looked up BY VALUE in the numpy twin of the model (tests/model_twin.py, written
independently row by row from traj_optimizer.cpp:225-527); then Farkas' lemma is checked on the twin's rows:
    lambda_i >= 0,   Z' sum_i lambda_i a_i = 0  (Z: null space of the twin's equality rows),   sum_i lambda_i b_i > (sum_i lambda_i a_i) . x_e
for any x_e that satisfies the equalities -- no point can then satisfy the equalities and the member rows together."""
import numpy as np

import model_twin

INF = 1e299


def twin_row_of(member, tw):
    """(a, b) of the twin's row `a . x >= b` that the member stands for; raises if the twin holds no such row."""
    nv = tw["nv"]
    a = np.zeros(nv)
    for v, c in zip(member["var"], member["coef"]):
        a[v] += c
    lo, hi = member["lo"], member["hi"]
    if member["kind"] in ("VAR_LB", "VAR_UB"):
        v = member["var"][0]
        if member["kind"] == "VAR_LB":
            assert np.isclose(tw["lb"][v], lo, rtol=1e-13, atol=0) and lo > -INF, ("variable lower bound", v, tw["lb"][v], lo)
            return a, lo
        assert np.isclose(tw["ub"][v], hi, rtol=1e-13, atol=0) and hi < INF, ("variable upper bound", v, tw["ub"][v], hi)
        return -a, -hi
    # an inequality row of the twin with these coefficients and this side
    A = tw["Ain"]
    hit = np.nonzero(np.all(np.isclose(A, a[None, :], rtol=1e-12, atol=1e-300), axis=1))[0]
    for r in hit:
        if lo > -INF and np.isclose(tw["lo"][r], lo, rtol=1e-12, atol=1e-15):
            return a, lo
        if hi < INF and np.isclose(tw["hi"][r], hi, rtol=1e-12, atol=1e-15):
            return -a, -hi
    raise AssertionError("the twin model has no row like %r" % (member,))


def check_certificate(cfg, agent, rows, sfc, members, gap, goal=None, row_scale=1.0, feas_tol=1e-6, resid_tol=1e-6):
    """Raises AssertionError unless the members prove the agent's QP infeasible.  Returns (margin, worst multiplier, residual)."""
    ag = np.array(agent, copy=True)
    if goal is not None:
        ag["current_goal_point"] = goal  # (rows do not depend on the goal; the twin wants a record)
    tw = model_twin.traj_model(cfg, ag, rows, sfc)
    assert len(members) >= 1 and gap > 0
    lam = np.array([m["lam"] for m in members])
    assert (lam >= -1e-9 * max(1.0, lam.max())).all(), ("negative multiplier", lam.min())
    lam = np.maximum(lam, 0.0)
    ab = [twin_row_of(m, tw) for m in members]
    u = sum(l * a for l, (a, _) in zip(lam, ab))
    rhs = sum(l * b for l, (_, b) in zip(lam, ab))
    Aeq, beq = tw["Aeq"], tw["beq"]
    # null space of the equalities and a point on them
    _, s, vt = np.linalg.svd(Aeq, full_matrices=True)
    rank = int((s > 1e-10 * s[0]).sum())
    Z = vt[rank:].T
    xe = np.linalg.lstsq(Aeq, beq, rcond=None)[0]
    assert np.abs(Aeq @ xe - beq).max() < 1e-8
    scale = max(np.abs(l * a).max() for l, (a, _) in zip(lam, ab))
    resid = np.abs(Z.T @ u).max() / scale
    # (measured over the nineteen configurations of tests/randqp.py: <= 1e-9 for 626 of 627 certificates, typically 1e-12; one agent with 82
    # members -- slack variables, M = 10, numbers blown up on the way to its verdict -- at 1.3e-7 of its largest term)
    assert resid <= resid_tol, ("sum lambda a does not vanish on the null space of the equalities", resid)
    margin = rhs - u @ xe
    # the margin IS the blocked row's violation (its multiplier is 1 in the kernel's scaling, the active rows hold with equality), and
    # in the reference's scaling of that row it is beyond CPLEX's feasibility tolerance -- else the kernel would have accepted the row
    assert margin > 0 and np.isclose(margin, gap, rtol=1e-4, atol=1e-9 * max(1.0, np.abs(rhs))), ("margin differs from the reported violation", margin, gap)
    assert gap * row_scale > feas_tol, ("violation within CPLEX's tolerance", gap, row_scale)
    return margin, float(lam.max()), float(resid)


def parse_lp(path):
    """The subset of the CPLEX-LP format dlscqp_write_lp writes -> (names, A, lo, hi, lb, ub): rows lo <= A x <= hi, bounds lb <= x <= ub.
    (The objective is skipped: a feasibility verdict does not depend on it.)"""
    txt = open(path).read().split("\n")
    sect = None
    cons, bounds = [], []
    for line in txt:
        t = line.strip()
        if not t or t.startswith("\\"):
            continue
        low = t.lower()
        if low in ("minimize", "subject to", "bounds", "end"):
            sect = low
            continue
        if sect == "subject to":
            cons.append(t)
        elif sect == "bounds":
            bounds.append(t)
    names = {}
    order = []

    def var(nm):
        if nm not in names:
            names[nm] = len(order)
            order.append(nm)
        return names[nm]

    for b in bounds:  # (every variable has a Bounds line: their order is the model's)
        tok = b.split()
        if tok[-1] == "free":
            var(tok[0])
        elif "<=" in tok and tok.count("<=") == 2:
            var(tok[2])
        else:
            var(tok[0])
    nv = len(order)
    lb, ub = np.full(nv, -np.inf), np.full(nv, np.inf)
    for b in bounds:
        tok = b.split()
        if tok[-1] == "free":
            continue
        if tok.count("<=") == 2:
            v = names[tok[2]]
            lb[v] = -np.inf if tok[0] == "-inf" else float(tok[0])
            ub[v] = float(tok[4])
        elif tok[1] == ">=":
            lb[names[tok[0]]] = float(tok[2])
    rows, lo, hi = [], [], []
    for c in cons:
        name, expr = c.split(":", 1)
        tok = expr.split()
        sense_at = next(i for i, x in enumerate(tok) if x in ("=", ">=", "<="))
        r = np.zeros(nv)
        for k in range(0, sense_at, 2):
            r[names[tok[k + 1]]] += float(tok[k])
        rhs = float(tok[sense_at + 1])
        rows.append(r)
        lo.append(rhs if tok[sense_at] in ("=", ">=") else -np.inf)
        hi.append(rhs if tok[sense_at] in ("=", "<=") else np.inf)
    return order, np.array(rows), np.array(lo), np.array(hi), lb, ub


def lp_feasible(path):
    """HiGHS (scipy.optimize.linprog) on the constraints of an .lp file: True / False."""
    from scipy.optimize import linprog
    _, A, lo, hi, lb, ub = parse_lp(path)
    eq = lo == hi
    A_ub = np.vstack([A[~eq & np.isfinite(hi)], -A[~eq & np.isfinite(lo)]])
    b_ub = np.concatenate([hi[~eq & np.isfinite(hi)], -lo[~eq & np.isfinite(lo)]])
    res = linprog(np.zeros(A.shape[1]), A_ub=A_ub if len(b_ub) else None, b_ub=b_ub if len(b_ub) else None, A_eq=A[eq], b_eq=lo[eq],
                  bounds=list(zip([None if not np.isfinite(v) else v for v in lb], [None if not np.isfinite(v) else v for v in ub])), method="highs")
    assert res.status in (0, 2), res.message
    return res.status == 0
