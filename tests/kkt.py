"""Solver-independent optimality certificate for the dense trajectory-QP model (test infrastructure).

For a convex QP  min x'Qx + c'x  s.t. Aeq x = beq, lo <= Ain x <= hi, lb <= x <= ub  a point is optimal iff it
is feasible and 2Qx + c = Aeq' nu + sum_i lambda_i a_i with lambda_i >= 0 supported on the active rows
(a_i oriented into the feasible side).  The multipliers are found by non-negative least squares, so the
check does not depend on how x was computed.  The reduced Hessian of this model is positive definite
(jerk cost + fixed initial state), hence the certified optimum is unique.
"""
import numpy as np
from scipy.optimize import lsq_linear


def certificate(md, x, act_tol=1e-6):
    """Returns dict(primal=max scaled infeasibility, stationarity=relative residual, n_active)."""
    nv = md["nv"]
    x = np.asarray(x, float)[:nv]
    g = (md["Q"] + md["Q"].T) @ x + md["c"]
    se = np.maximum(np.abs(md["Aeq"]).max(1), 1e-300) if md["neq"] else np.zeros(0)
    prim = np.abs(md["Aeq"] @ x - md["beq"]) / se if md["neq"] else np.zeros(0)
    pr = float(prim.max()) if prim.size else 0.0
    cols = []
    if md["nin"]:
        si = np.maximum(np.linalg.norm(md["Ain"], axis=1), 1e-300)
        ax = md["Ain"] @ x
        for i in range(md["nin"]):
            if md["lo"][i] > -1e299:
                v = (md["lo"][i] - ax[i]) / si[i]
                pr = max(pr, v)
                if v > -act_tol:
                    cols.append(md["Ain"][i] / si[i])
            if md["hi"][i] < 1e299:
                v = (ax[i] - md["hi"][i]) / si[i]
                pr = max(pr, v)
                if v > -act_tol:
                    cols.append(-md["Ain"][i] / si[i])
    for j in range(nv):
        if md["lb"][j] > -1e299:
            v = md["lb"][j] - x[j]
            pr = max(pr, v)
            if v > -act_tol:
                e = np.zeros(nv)
                e[j] = 1
                cols.append(e)
        if md["ub"][j] < 1e299:
            v = x[j] - md["ub"][j]
            pr = max(pr, v)
            if v > -act_tol:
                e = np.zeros(nv)
                e[j] = -1
                cols.append(e)
    na = len(cols)
    Ae = (md["Aeq"] / se[:, None]).T if md["neq"] else np.zeros((nv, 0))
    A = np.hstack([Ae, np.array(cols).T.reshape(nv, na)]) if na else Ae
    lb = np.concatenate([np.full(Ae.shape[1], -np.inf), np.zeros(na)])
    ub = np.full(A.shape[1], np.inf)
    if A.shape[1] == 0:
        res = np.linalg.norm(g)
    else:
        sol = lsq_linear(A, g, bounds=(lb, ub), method="bvls", tol=1e-14, max_iter=2000)
        res = np.linalg.norm(A @ sol.x - g)
    return dict(primal=pr, stationarity=res / (1.0 + np.linalg.norm(g)), n_active=na)
