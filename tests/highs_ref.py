"""HiGHS (scipy-bundled, private _highspy API) as an independent QP solver for the dense model.

Container-only: used by tests/golden/make_golden.py to produce committed golden vectors and by
the optional cross-solver test (skipped where the private module is unavailable)."""
import numpy as np


def available():
    try:
        from scipy.optimize._highspy import _core  # noqa: F401
        return True
    except Exception:
        return False


def solve_model(md, tol=1e-9):
    """Solve min x'Qx + c'x + c0 s.t. Aeq x = beq, lo <= Ain x <= hi, lb <= x <= ub.
    Returns (status_str, x, objective)."""
    from scipy.optimize._highspy import _core as h
    import scipy.sparse as sp
    nv = md["nv"]
    inf = h.kHighsInf
    A = np.vstack([md["Aeq"], md["Ain"]]) if md["nin"] else md["Aeq"]
    rl = np.concatenate([md["beq"], np.where(md["lo"] <= -1e299, -inf, md["lo"])])
    ru = np.concatenate([md["beq"], np.where(md["hi"] >= 1e299, inf, md["hi"])])
    lb = np.where(md["lb"] <= -1e299, -inf, md["lb"])
    ub = np.where(md["ub"] >= 1e299, inf, md["ub"])
    model = h.HighsModel()
    lp = model.lp_
    lp.num_col_ = nv
    lp.num_row_ = A.shape[0]
    lp.col_cost_ = md["c"].astype(float)
    lp.col_lower_ = lb
    lp.col_upper_ = ub
    lp.row_lower_ = rl
    lp.row_upper_ = ru
    lp.offset_ = float(md["c0"])
    Ac = sp.csc_matrix(A)
    lp.a_matrix_.format_ = h.MatrixFormat.kColwise
    lp.a_matrix_.num_col_ = nv
    lp.a_matrix_.num_row_ = A.shape[0]
    lp.a_matrix_.start_ = Ac.indptr.astype(np.int32)
    lp.a_matrix_.index_ = Ac.indices.astype(np.int32)
    lp.a_matrix_.value_ = Ac.data.astype(float)
    Hs = sp.csc_matrix(np.tril(md["Q"] + md["Q"].T))  # objective 1/2 x'Hx with H = 2Q
    hes = model.hessian_
    hes.dim_ = nv
    hes.format_ = h.HessianFormat.kTriangular
    hes.start_ = Hs.indptr.astype(np.int32)
    hes.index_ = Hs.indices.astype(np.int32)
    hes.value_ = Hs.data.astype(float)
    H = h._Highs()
    H.setOptionValue("output_flag", False)
    H.setOptionValue("primal_feasibility_tolerance", tol)
    H.setOptionValue("dual_feasibility_tolerance", tol)
    H.passModel(model)
    H.run()
    st = H.modelStatusToString(H.getModelStatus())
    sol = H.getSolution()
    x = np.array(sol.col_value)
    return st, x, H.getObjectiveValue()
