// qp_tables.cpp -- host-side, once-per-context precomputation for the batched trajectory QP.
//
// What TrajOptimizer's constructor caches in the reference (buildQBase / buildAeqBase,
// traj_optimizer.cpp:4-16, 172-223) plus what CPLEX's presolve/factorisation would redo for every
// agent and every replanning step.  Everything here depends only on (dim, M, n, phi, dt, weights,
// modes), i.e. it is identical for every agent, so it is done once and kept in HBM:
//   * per-axis equality block E c = S [pos vel acc]'  (traj_optimizer.cpp:333-381, 515-524)
//     -> orthonormal null-space basis Z and minimum-norm particular-solution map Xp (Householder QR)
//   * for every possible number of terminal segments T = 1..M (traj_optimizer.cpp:301-315, 543-551):
//     reduced Hessian Z'HZ = L L', whitening map N_T = Z L^{-T}, its Gram table G_T = N_T N_T'
//     (= Z (Z'HZ)^-1 Z', the inverse Hessian on the null space of the equalities: what the kernel reads),
//     and the unconstrained minimiser as a linear map of (pos, vel, acc, goal): x0 = X0s_T s + X0g_T goal
//   * the table of structured two-sided rows shared by all agents (bounds/SFC faces, dynamic
//     limits, communication range).
#include "qp_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace dlscqp {

namespace {

int n_choose_k(int n, int k) {
    if (k > n) return 0;
    if (k * 2 > n) k = n - k;
    if (k == 0) return 1;
    int result = n;
    for (int i = 2; i <= k; i++) {
        result *= (n - i + 1);
        result /= i;
    }
    return result;
}

int falling(int n, int k) {  // n (n-1) ... (n-k+1), 0 if n < k
    if (n < k) return 0;
    int c = 1;
    for (int i = 0; i < k; i++) c *= n - i;
    return c;
}

using Mat = std::vector<double>;  // row-major

}  // namespace

// Jerk Gram matrix of one segment in Bernstein coordinates, times dt^(1-2*phi)
// (traj_optimizer.cpp:172-187; B from polynomial.hpp:280-293).
static void build_q_base(int n, int phi, int phi_n, double dt, double *Qb) {
    const int N = n + 1;
    Mat B(N * N, 0.0), Z(N * N), T(N * N);
    for (int i = 0; i < N; i++)
        for (int j = i; j < N; j++)
            B[i * N + j] = n_choose_k(n, i) * n_choose_k(n - i, n - j) * (((j - i) & 1) ? -1.0 : 1.0);
    std::fill(Qb, Qb + N * N, 0.0);
    for (int k = phi; k > phi - phi_n; k--) {
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                int den = i + j - 2 * k + 1;
                Z[i * N + j] = den > 0 ? (double)falling(i, k) * falling(j, k) / den : 0.0;
            }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int l = 0; l < N; l++) s += B[i * N + l] * Z[l * N + j];
                T[i * N + j] = s;
            }
        const double sc = std::pow(dt, -2 * k + 1);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int l = 0; l < N; l++) s += T[i * N + l] * B[j * N + l];
                Qb[i * N + j] += s * sc;
            }
    }
}

// Per-axis equality rows: E (ne x P) and S (ne x 3) with E c = S [pos vel acc]'.
static int build_axis_equalities(int M, int n, int phi, double dt, bool stop, Mat &E, Mat &S) {
    const int N = n + 1, P = M * N;
    const int ne = 6 + (M - 2) * phi + (stop ? phi - 1 : 0);
    E.assign((size_t)ne * P, 0.0);
    S.assign((size_t)ne * 3, 0.0);
    int e = 0;
    auto at = [&](int row, int m, int i) -> double & { return E[(size_t)row * P + m * N + i]; };
    // traj_optimizer.cpp:336-366
    at(e, 0, 0) = 1; S[e * 3 + 0] = 1; e++;                               // front position
    at(e, 0, n) = 1; at(e, 1, 0) = -1; e++;                               // back position
    at(e, 0, 1) = n / dt; at(e, 0, 0) = -n / dt; S[e * 3 + 1] = 1; e++;   // front velocity
    {
        double c = n * (n - 1) / (dt * dt);
        at(e, 0, 2) = c; at(e, 0, 1) = -2 * c; at(e, 0, 0) = c; S[e * 3 + 2] = 1; e++;  // front acceleration
    }
    at(e, 1, 1) += 1; at(e, 1, 0) += -1; at(e, 0, n) += -1; at(e, 0, n - 1) += 1; e++;  // back velocity
    at(e, 1, 2) += 1; at(e, 1, 1) += -2; at(e, 1, 0) += 1;
    at(e, 0, n) += -1; at(e, 0, n - 1) += 2; at(e, 0, n - 2) += -1; e++;                // back acceleration
    // traj_optimizer.cpp:189-223, 370-381: d^j/dt^j continuity between segments m-1 and m, m >= 2.
    // Row j of A_T / A_0 is the j-th forward difference at the end / start of a segment.
    for (int m = 2; m < M; m++) {
        int nn = 1;
        for (int j = 0; j < phi; j++) {
            const double sc = std::pow(dt, -j) * nn;
            for (int l = 0; l <= j; l++) {
                const double binom = n_choose_k(j, l) * (((j - l) & 1) ? -1.0 : 1.0);
                at(e, m - 1, n - j + l) += sc * binom;  // end of segment m-1
                at(e, m, l) += -sc * binom;             // start of segment m
            }
            nn *= (n - j);
            e++;
        }
    }
    if (stop)  // traj_optimizer.cpp:515-524
        for (int i = 1; i < phi; i++) {
            at(e, M - 1, n) = 1;
            at(e, M - 1, n - i) = -1;
            e++;
        }
    return ne;
}

static bool cholesky_lower(Mat &A, int n) {
    for (int j = 0; j < n; j++) {
        double s = A[j * n + j];
        for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0)) return false;
        const double d = std::sqrt(s);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[i * n + j];
            for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / d;
        }
    }
    return true;
}

bool build_tables(const dlscqp_config &cfg, int dim, int M, int n, int phi, double dt, HostTables &out,
                  std::string &err) {
    if (dim < 2 || dim > 3) { err = "Invalid output dimension, output_dim > 3"; return false; }  // :248
    if (!(n == 5 && phi == 3)) { err = "Currently, only n=5, phi=3 is available"; return false; }  // :209
    if (cfg.phi_n < 1 || cfg.phi_n > phi) { err = "phi_n must be in [1, phi]"; return false; }
    if (M < 2 || M > 32) { err = "M must be in [2, 32]"; return false; }
    if (!(dt > 0)) { err = "dt must be positive"; return false; }
    const int N = n + 1, P = M * N;
    const bool stop = cfg.planner_mode == DLSCQP_PLANNER_DLSCGC || cfg.planner_mode == DLSCQP_PLANNER_LSC;

    KParams &kp = out.kp;
    std::memset(&kp, 0, sizeof(kp));
    kp.dim = dim; kp.M = M; kp.n1 = N; kp.P = P; kp.phi = phi;
    kp.dt = dt;
    kp.w_c = cfg.control_input_weight; kp.w_t = cfg.terminal_weight; kp.w_s = cfg.slack_collision_weight;
    kp.comm_range = cfg.communication_range;
    kp.has_comm = cfg.communication_range > 0 ? 1 : 0;
    kp.use_sfc = cfg.world_use_octomap ? 1 : 0;
    kp.stop_mode = stop ? 1 : 0;
    kp.rsfc_mode = cfg.planner_mode == DLSCQP_PLANNER_RECIPROCALRSFC ? 1 : 0;
    kp.z_2d = cfg.world_z_2d;
    for (int k = 0; k < 3; k++) { kp.world_min[k] = cfg.world_min[k]; kp.world_max[k] = cfg.world_max[k]; }
    // objective_upper_bound() of gi_core.hpp bounds c'Q_base c by the integral of jerk^2 only: with phi_n > 1 Q_base also
    // holds the lower-derivative Gram terms (traj_optimizer.cpp:172-187) and the bound would not hold
    kp.wd_cut = cfg.phi_n == 1 ? 1 : 0;
    // the parameter values of the reference's launch files (all 143 of them: SURVEY.md appendix C): what gi_core.hpp's narrow
    // dependency threshold was calibrated on
    kp.dep_calibrated = (std::fabs(cfg.control_input_weight - 0.01) < 1e-12 && std::fabs(cfg.terminal_weight - 1.0) < 1e-12 &&
                         std::fabs(dt - 0.2) < 1e-12 && cfg.phi_n == 1) ? 1 : 0;
    kp.dep_suspect = kp.dep_calibrated ? DEP_SUSPECT : DEP_SUSPECT_SLACK;
    if (std::getenv("DLSCQP_DEBUG_NO_WD")) kp.wd_cut = 0;  // debugging aid: verdicts with and without the test must agree (tests)
    kp.vel_scale = dt / n;
    kp.acc_scale = dt * dt / (n * (n - 1));

    double Qb[36];
    build_q_base(n, phi, cfg.phi_n, dt, Qb);
    for (int i = 0; i < 36; i++) kp.Qb[i] = cfg.control_input_weight != 0 ? cfg.control_input_weight * Qb[i] : 0.0;
    for (int m = 0; m < M; m++) {
        const double h = 2.0 * cfg.slack_collision_weight * (double)(M - m) / M;
        kp.sigma[m] = h > 0 ? 1.0 / std::sqrt(h) : 0.0;
    }

    // ---- null space of the per-axis equalities: Householder QR of E' (P x ne) -------------------
    Mat E, S;
    const int ne = build_axis_equalities(M, n, phi, dt, stop, E, S);
    const int f = P - ne;
    if (f <= 0) { err = "no degrees of freedom left"; return false; }
    kp.f = f;
    Mat At((size_t)ne * P), V((size_t)ne * P, 0.0);  // column k of E' stored at At[k*P ..]
    for (int e = 0; e < ne; e++)
        for (int j = 0; j < P; j++) At[(size_t)e * P + j] = E[(size_t)e * P + j];
    double rmax = 0;
    for (int k = 0; k < ne; k++) {
        double *col = &At[(size_t)k * P];
        double nrm = 0;
        for (int j = k; j < P; j++) nrm += col[j] * col[j];
        nrm = std::sqrt(nrm);
        rmax = std::max(rmax, nrm);
        if (!(nrm > 1e-12 * std::max(rmax, 1.0))) { err = "dependent equality constraints"; return false; }
        const double alpha = col[k] > 0 ? -nrm : nrm;
        double *v = &V[(size_t)k * P];
        for (int j = k; j < P; j++) v[j] = col[j];
        v[k] -= alpha;
        double vn = 0;
        for (int j = k; j < P; j++) vn += v[j] * v[j];
        vn = std::sqrt(vn);
        for (int j = k; j < P; j++) v[j] /= vn;
        for (int c = k; c < ne; c++) {
            double *cc = &At[(size_t)c * P];
            double dot = 0;
            for (int j = k; j < P; j++) dot += v[j] * cc[j];
            for (int j = k; j < P; j++) cc[j] -= 2 * dot * v[j];
        }
    }
    Mat Qf((size_t)P * P, 0.0);
    for (int i = 0; i < P; i++) Qf[(size_t)i * P + i] = 1;
    for (int k = ne - 1; k >= 0; k--) {
        const double *v = &V[(size_t)k * P];
        for (int c = 0; c < P; c++) {
            double dot = 0;
            for (int j = k; j < P; j++) dot += v[j] * Qf[(size_t)j * P + c];
            if (dot != 0)
                for (int j = k; j < P; j++) Qf[(size_t)j * P + c] -= 2 * dot * v[j];
        }
    }
    // Xp = Q1 R^{-T} S   (P x 3)
    Mat Y((size_t)ne * 3), Xp((size_t)P * 3), Z((size_t)P * f);
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < ne; i++) {
            double t = S[i * 3 + c];
            for (int k = 0; k < i; k++) t -= At[(size_t)i * P + k] * Y[k * 3 + c];  // R(k,i)
            Y[i * 3 + c] = t / At[(size_t)i * P + i];
        }
    for (int j = 0; j < P; j++) {
        for (int c = 0; c < 3; c++) {
            double t = 0;
            for (int k = 0; k < ne; k++) t += Qf[(size_t)j * P + k] * Y[k * 3 + c];
            Xp[j * 3 + c] = t;
        }
        for (int k = 0; k < f; k++) Z[(size_t)j * f + k] = Qf[(size_t)j * P + ne + k];
    }

    // ---- one whitening per number of terminal segments ----------------------------------------
    out.tabN.assign((size_t)M * P * f, 0.0);
    out.tabX0s.assign((size_t)M * P * 3, 0.0);
    out.tabX0g.assign((size_t)M * P, 0.0);
    out.tabPhi.assign((size_t)M * 10, 0.0);
    out.hr_cond.assign((size_t)M, 0.0);
    out.hr_lmax.assign((size_t)M, 0.0);
    out.hr_lmin.assign((size_t)M, 0.0);
    kp.phi_ok = 1;
    Mat H((size_t)P * P), HZ((size_t)P * f), Hr((size_t)f * f), Nt((size_t)P * f), HXp((size_t)P * 3), G((size_t)f * 3), tv(f);
    for (int T = 1; T <= M; T++) {
        std::fill(H.begin(), H.end(), 0.0);
        for (int m = 0; m < M; m++)
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) H[(size_t)(m * N + i) * P + m * N + j] = 2.0 * kp.Qb[i * N + j];
        for (int m = M - T; m < M; m++) H[(size_t)(m * N + n) * P + m * N + n] += 2.0 * cfg.terminal_weight;
        for (int i = 0; i < P; i++) {
            for (int k = 0; k < f; k++) {
                double t = 0;
                for (int j = 0; j < P; j++) t += H[(size_t)i * P + j] * Z[(size_t)j * f + k];
                HZ[(size_t)i * f + k] = t;
            }
            for (int c = 0; c < 3; c++) {
                double t = 0;
                for (int j = 0; j < P; j++) t += H[(size_t)i * P + j] * Xp[j * 3 + c];
                HXp[i * 3 + c] = t;
            }
        }
        for (int a = 0; a < f; a++)
            for (int b = 0; b < f; b++) {
                double t = 0;
                for (int i = 0; i < P; i++) t += Z[(size_t)i * f + a] * HZ[(size_t)i * f + b];
                Hr[a * f + b] = t;
            }
        for (int a = 0; a < f; a++)
            for (int b = 0; b < a; b++) Hr[a * f + b] = Hr[b * f + a] = 0.5 * (Hr[a * f + b] + Hr[b * f + a]);
        Mat L = Hr;
        if (!cholesky_lower(L, f)) { kp.t_valid[T - 1] = 0; continue; }
        kp.t_valid[T - 1] = 1;
        {
            // condition number of the reduced Hessian: largest eigenvalue by power iteration, smallest by inverse iteration with
            // the factor just made (what the rounding noise of everything computed with the whitening map scales with)
            std::vector<double> v(f, 1.0), y(f);
            double lmax = 0.0, lmin_inv = 0.0;
            for (int it = 0; it < 200; it++) {
                double nrm = 0;
                for (int a = 0; a < f; a++) { double t = 0; for (int b = 0; b < f; b++) t += Hr[a * f + b] * v[b]; y[a] = t; nrm += t * t; }
                nrm = std::sqrt(nrm);
                if (!(nrm > 0)) break;
                lmax = nrm;
                for (int a = 0; a < f; a++) v[a] = y[a] / nrm;
            }
            std::fill(v.begin(), v.end(), 1.0);
            for (int a = 0; a < f; a++) v[a] += 0.01 * a;  // (not an eigenvector of a symmetric pattern by accident)
            for (int it = 0; it < 200; it++) {
                for (int i = 0; i < f; i++) { double t = v[i]; for (int k = 0; k < i; k++) t -= L[i * f + k] * y[k]; y[i] = t / L[i * f + i]; }
                for (int i = f - 1; i >= 0; i--) { double t = y[i]; for (int k = i + 1; k < f; k++) t -= L[k * f + i] * v[k]; v[i] = t / L[i * f + i]; }
                double nrm = 0;
                for (int a = 0; a < f; a++) nrm += v[a] * v[a];
                nrm = std::sqrt(nrm);
                if (!(nrm > 0)) break;
                lmin_inv = nrm;
                for (int a = 0; a < f; a++) v[a] /= nrm;
            }
            out.hr_cond[T - 1] = lmax * lmin_inv;
            out.hr_lmax[T - 1] = lmax;
            out.hr_lmin[T - 1] = lmin_inv > 0 ? 1.0 / lmin_inv : 0.0;
        }
        // N = Z L^{-T}:  row p of N solves  L N[p,:]' = Z[p,:]'
        for (int p = 0; p < P; p++)
            for (int i = 0; i < f; i++) {
                double t = Z[(size_t)p * f + i];
                for (int k = 0; k < i; k++) t -= L[i * f + k] * Nt[(size_t)p * f + k];
                Nt[(size_t)p * f + i] = t / L[i * f + i];
            }
        // G = N' H Xp (f x 3), tv = 2 w_t sum_{m >= M-T} N[(m,n), :]
        for (int a = 0; a < f; a++) {
            for (int c = 0; c < 3; c++) {
                double t = 0;
                for (int i = 0; i < P; i++) t += Nt[(size_t)i * f + a] * HXp[i * 3 + c];
                G[a * 3 + c] = t;
            }
            double t = 0;
            for (int m = M - T; m < M; m++) t += Nt[(size_t)(m * N + n) * f + a];
            tv[a] = 2.0 * cfg.terminal_weight * t;
        }
        double *oN = &out.tabN[(size_t)(T - 1) * P * f];
        double *oXs = &out.tabX0s[(size_t)(T - 1) * P * 3];
        double *oXg = &out.tabX0g[(size_t)(T - 1) * P];
        for (int p = 0; p < P; p++) {
            for (int a = 0; a < f; a++) oN[(size_t)p * f + a] = Nt[(size_t)p * f + a];
            for (int c = 0; c < 3; c++) {
                double t = Xp[p * 3 + c];
                for (int a = 0; a < f; a++) t -= Nt[(size_t)p * f + a] * G[a * 3 + c];
                oXs[p * 3 + c] = t;
            }
            double t = 0;
            for (int a = 0; a < f; a++) t += Nt[(size_t)p * f + a] * tv[a];
            oXg[p] = t;
        }
        // The objective at the unconstrained minimiser, per axis, as a quadratic form s' Psi_T s of s = (pos, vel, acc, goal):
        // x0 = [X0s X0g] s, so c' Qb c and the terminal terms (c_{m,n} - goal)^2 are quadratic in s with the very tables the
        // kernel builds x0 from.  What an agent that never iterates reports as its cost (cplex.getObjValue(),
        // traj_optimizer.cpp:109) without a pass over its control points.  (A 3 x 3 form in (goal - pos, vel, acc) would do
        // in exact arithmetic -- the model is translation-invariant -- but the tables are so only to ~1e-10, which shows in
        // costs of 1e-2.)
        {
            double *oPhi = &out.tabPhi[(size_t)(T - 1) * 10];
            auto Bcol = [&](int p, int c) { return c == 3 ? oXg[p] : oXs[p * 3 + c]; };
            for (int a = 0, q = 0; a < 4; a++)
                for (int b = a; b < 4; b++, q++) {
                    double t = 0;
                    for (int m = 0; m < M; m++)
                        for (int i = 0; i < N; i++)
                            for (int j = 0; j < N; j++) t += Bcol(m * N + i, a) * kp.Qb[i * N + j] * Bcol(m * N + j, b);
                    for (int m = M - T; m < M; m++) {
                        const int pe = m * N + n;
                        t += cfg.terminal_weight * (Bcol(pe, a) - (a == 3 ? 1.0 : 0.0)) * (Bcol(pe, b) - (b == 3 ? 1.0 : 0.0));
                    }
                    oPhi[q] = t;
                }
        }
    }

    // ---- structured two-sided rows ----------------------------------------------------------------
    out.slots.clear();
    auto add = [&](int kind, int bsel, float scale, int i0, int c0, int i1 = -1, int c1 = 0, int i2 = -1, int c2 = 0) {
        Slot s;
        s.idx[0] = (int16_t)i0; s.idx[1] = (int16_t)i1; s.idx[2] = (int16_t)i2;
        s.coef[0] = (int8_t)c0; s.coef[1] = (int8_t)c1; s.coef[2] = (int8_t)c2;
        s.kind = (uint8_t)kind; s.bsel = (uint16_t)bsel; s.scale = scale;
        out.slots.push_back(s);
    };
    const int B_VEL = dim * M, B_ACC = dim * M + 1, B_COMM = dim * M + 2, B_WP = dim * M + 3;
    kp.nb = dim * M + 3 + dim;
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < N; i++) {
                if (m == 0 && i < 3) continue;  // traj_optimizer.cpp:260-262, 390-392
                add(SLOT_BOUND, k * M + m, 1.0f, k * P + m * N + i, 1);
            }
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++) {
            for (int i = 0; i < n; i++) {
                if (m == 0 && (i == 0 || i == 1)) continue;  // :457-459
                add(SLOT_VEL, B_VEL, (float)(n / dt), k * P + m * N + i + 1, 1, k * P + m * N + i, -1);
            }
            for (int i = 0; i < n - 1; i++) {
                if (m == 0 && i == 0) continue;  // :471-473
                add(SLOT_ACC, B_ACC, (float)(n * (n - 1) / (dt * dt)), k * P + m * N + i + 2, 1,
                    k * P + m * N + i + 1, -2, k * P + m * N + i, 1);
            }
        }
    kp.n_struct = (int)out.slots.size();
    if (kp.has_comm) {  // :490-513
        for (int k = 0; k < dim; k++)
            for (int mi = 0; mi < M; mi++)
                for (int m = mi; m < M; m++) add(SLOT_COMM, B_COMM, 1.0f, k * P + m * N + n, 1, k * P + mi * N + 0, -1);
        for (int k = 0; k < dim; k++)
            for (int m = 0; m < M; m++) add(SLOT_WAYPOINT, B_WP + k, 1.0f, k * P + m * N + n, 1);
    }
    kp.n_slots = (int)out.slots.size();
    // pivot scaling: a violated row is ranked by its distance in the whitened space, violation / |N_T' a|
    // (steepest dual ascent); this roughly halves the active-set iterations of the hard agents
    out.tabNN.assign((size_t)M * P, 0.0);
    out.tabSIW.assign((size_t)M * kp.n_slots, 0.0);
    // Gram tables G_T = N_T N_T' (P x P, row stride P | 1): all the kernel needs of the whitened space -- a row of the model
    // touches at most three control points, so a_i' G a_j is a handful of entries and G (a - A' r) a handful of rows
    const int ldg = P | 1;
    out.tabG.assign((size_t)M * P * ldg, 0.0);
    out.tabKIW.assign((size_t)M * 3 * P, 0.0);
    for (int T = 1; T <= M; T++) {
        if (!kp.t_valid[T - 1]) continue;
        const double *oN = &out.tabN[(size_t)(T - 1) * P * f];
        double *oG = &out.tabG[(size_t)(T - 1) * P * ldg];
        for (int p = 0; p < P; p++)
            for (int p2 = 0; p2 <= p; p2++) {
                double t = 0;
                for (int a = 0; a < f; a++) t += oN[(size_t)p * f + a] * oN[(size_t)p2 * f + a];
                oG[(size_t)p * ldg + p2] = oG[(size_t)p2 * ldg + p] = t;  // exactly symmetric
            }
        for (int p = 0; p < P; p++) {
            double t = 0;
            for (int a = 0; a < f; a++) t += oN[(size_t)p * f + a] * oN[(size_t)p * f + a];
            out.tabNN[(size_t)(T - 1) * P + p] = t;
        }
        // the same scalings by kind and first control point (they do not depend on the axis): what the static-shape kernels
        // read instead of the slot table
        double *oK = &out.tabKIW[(size_t)(T - 1) * 3 * P];
        for (int p = 0; p < P; p++) {
            const double *g = oG;
            auto at = [&](int a, int b) { return (a < P && b < P) ? g[(size_t)a * ldg + b] : 0.0; };
            const double wb = at(p, p);
            const double wv = at(p + 1, p + 1) - 2.0 * at(p + 1, p) + at(p, p);
            const double c[3] = {1.0, -2.0, 1.0};
            double wa = 0;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) wa += c[a] * c[b] * at(p + a, p + b);
            oK[p] = wb > 0 ? 1.0 / std::sqrt(wb) : 0.0;
            // rows that do not exist (they would reach into the next segment) get scaling 0: across a joint the difference
            // of two control points is pinned by the continuity rows, its whitened length is ~0 and 1 / it would overflow
            const int i = p % N;
            oK[P + p] = (i < n && wv > 0) ? 1.0 / std::sqrt(wv) : 0.0;
            oK[2 * P + p] = (i < n - 1 && wa > 0) ? 1.0 / std::sqrt(wa) : 0.0;
        }
        for (int si = 0; si < kp.n_slots; si++) {
            const Slot &sl = out.slots[si];
            double t = 0;
            for (int a = 0; a < f; a++) {
                double v = 0;
                for (int e = 0; e < 3; e++)
                    if (sl.idx[e] >= 0) v += (double)sl.coef[e] * oN[(size_t)(sl.idx[e] % P) * f + a];
                t += v * v;
            }
            out.tabSIW[(size_t)(T - 1) * kp.n_slots + si] = t > 0 ? 1.0 / std::sqrt(t) : 0.0;
        }
    }
    kp.max_iter = 0;  // 0: 20 (r + 8) + 200, chosen per agent in plan_agent
    kp.dbg_exit_phase = -1;
    if (const char *dbg = std::getenv("DLSCQP_DEBUG_EXIT_PHASE")) kp.dbg_exit_phase = std::atoi(dbg);  // read by -DDQ_PHASE_PROFILE builds only
    if (const char *dbg = std::getenv("DLSCQP_DEBUG_MAX_ITER")) kp.max_iter = std::atoi(dbg);  // debugging aid
    {
        // DEP_PIVOT x eps x the condition number of the Hessian the solve works with: the reduced Hessians of all T and the slack
        // variables' diagonal 2 w_s (M - m) / M = 1 / sigma_m^2 (present or not: the larger number is the careful one)
        double lmax = 0.0, lmin = 1e300;
        for (int T = 1; T <= M; T++)
            if (kp.t_valid[T - 1]) { lmax = std::max(lmax, out.hr_lmax[T - 1]); lmin = std::min(lmin, out.hr_lmin[T - 1]); }
        if (cfg.slack_collision_weight > 0)
            for (int m = 0; m < M; m++)
                if (kp.sigma[m] > 0) { const double h = 1.0 / (kp.sigma[m] * kp.sigma[m]); lmax = std::max(lmax, h); lmin = std::min(lmin, h); }
        const double kappa = (lmax > 0 && lmin > 0 && lmin < 1e300) ? lmax / lmin : 1.0;
        kp.dep_pivot = DEP_PIVOT * 2.220446049250313e-16 * kappa;
    }
    return true;
}

}  // namespace dlscqp
