// native_qp.cpp -- TEST / BENCH INFRASTRUCTURE: a CPU-shaped solver for the same per-agent problem as the gfx950 kernel.
//
// SURVEY.md 8d(ii) / BASELINE.md 3 ask for "the build's own C++ FP64 solver, OpenMP over agents" beside the GPU figure.
// tests/emul runs the wave64 program lane by lane (every loop 64 trips); THIS is a scalar program: plain loops over the agent's
// actual rows, variables and active rows, one agent per thread.  Same mathematics as csrc/gi_core.hpp -- equalities eliminated and
// Hessian whitened by the host tables of csrc/qp_tables.cpp, Goldfarb-Idnani dual active-set method in range-space form with the
// columns Z = G A' S of the orthonormalised active normals kept beside S = R^-1 -- written independently of the kernel's source,
// so that it also checks it.  Model: /root/reference/src/traj_optimizer.cpp:225-527, goal LP: goal_optimizer.cpp:138-198.
// Never loaded by the product library (dlsc_gc_planner_amd/); parity-checked against oracle/ by tests/test_native.py.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../dlsc_gc_planner_amd/csrc/qp_tables.hpp"

using namespace dlscqp;

namespace {

struct SparseRow {   // a . x >= rhs
    int idx[4];      // x-space indices (control-point coordinates, then at most one slack variable)
    double cf[4];
    int n;
    double rhs, iw, scale;  // pivot scaling 1 / |N'a|, factor back to the reference's row scaling
    int src;                // the agent's input row it came from, -1: a row of the model's own
};

#ifdef SKIP_STUDY
// tests/tools/crowd_skip_study.py: how many blocks of 64 input rows a pricing pass could leave out if it kept, per block, the smallest
// scaled slack it saw and the length of the path x has gone since (a scaled slack changes by at most |dx|_H = t sqrt(a.dx) per step)
struct SkipStudy { double passes, blocks, priced, blocks_late, priced_late, agents; };
SkipStudy g_study;
int g_study_first = 3, g_study_gran = 1;
#endif

struct Problem {
    const HostTables *tb;
    int dim, M, P, f, nxc, nslk, nx, r, T;
    const double *G, *N;
    int ldg;
    std::vector<double> sig2;  // per slack variable: sigma^2
};

double f32_norm3d(float a, float b, float c) {
    float sq = a * a + b * b + c * c;
    return sqrt((double)sq);
}

int terminal_segments(const KParams &kp, const float *gl, const float *pos, double nominal_velocity) {
    const int M = kp.M;
    const double ift = f32_norm3d(gl[0] - pos[0], gl[1] - pos[1], gl[2] - pos[2]) / nominal_velocity;
    int T = (int)((M * kp.dt - ift + SP_EPSILON) / kp.dt);
    T = T < 1 ? 1 : T;
    return T > M ? M : T;
}

// goal_optimizer.cpp:7-198 in closed form (bound propagation on t in [0, 1 + 1e-5])
int goal_lp(const KParams &kp, const dlscqp_agent &ag, const dlscqp_lsc_row *rows, int K, const float *sfc, float *goal) {
    const int M = kp.M, n = kp.n1 - 1, dim = kp.dim;
    const float *cg = ag.current_goal_point, *wp = ag.next_waypoint;
    for (int k = 0; k < 3; k++) goal[k] = cg[k];
    const float diff[3] = {cg[0] - wp[0], cg[1] - wp[1], cg[2] - wp[2]};
    if (f32_norm3d(diff[0], diff[1], diff[2]) < SP_EPSILON_FLOAT) {
        for (int k = 0; k < 3; k++) goal[k] = wp[k];
        return DLSCQP_OK;
    }
    double t_lo = 0.0, t_hi = 1.0 + SP_EPSILON_FLOAT;
    bool infeasible = false, viol_any = false;
    auto row = [&](double a, double b) {
        if (a > 0) { if (-b / a > t_lo) t_lo = -b / a; }
        else if (a < 0) { if (-b / a < t_hi) t_hi = -b / a; }
        else if (b < -CPLEX_FEAS_TOL) infeasible = true;
    };
    const float *lb = (kp.use_sfc && sfc) ? sfc + (M - 1) * 6 : nullptr;
    if (lb)
        for (int k = 0; k < dim; k++) {
            row((double)diff[k], (double)wp[k] - (double)lb[k]);
            row(-(double)diff[k], -(double)wp[k] + (double)lb[3 + k]);
        }
    for (int j = 0; j < K; j++) {
        const dlscqp_lsc_row &R = rows[j];
        if ((R.flags & DLSCQP_ROW_DYNAMIC) || R.m != M - 1 || R.i != n) continue;
        if (f32_norm3d(R.normal[0], R.normal[1], R.normal[2]) < SP_EPSILON_FLOAT) continue;
        const float o[3] = {wp[0] - R.point[0], wp[1] - R.point[1], wp[2] - R.point[2]};
        double a = 0, b = -R.d;
        for (int k = 0; k < dim; k++) { a += (double)R.normal[k] * (double)diff[k]; b += (double)R.normal[k] * (double)o[k]; }
        row(a, b);
        const float e[3] = {cg[0] - R.point[0], cg[1] - R.point[1], cg[2] - R.point[2]};
        const float dot = R.normal[0] * e[0] + R.normal[1] * e[1] + R.normal[2] * e[2];
        if ((double)dot - R.d < -SP_EPSILON_FLOAT) viol_any = true;
    }
    if (t_lo > t_hi + CPLEX_FEAS_TOL) infeasible = true;
    if (!infeasible) {
        const float tf = (float)t_lo;
        for (int k = 0; k < 3; k++) goal[k] = diff[k] * tf + wp[k];
        return DLSCQP_OK;
    }
    bool numerical = true;  // goal_optimizer.cpp:54-81
    if (lb)
        for (int k = 0; k < 3; k++)
            if (!((double)cg[k] > (double)lb[k] - SP_EPSILON_FLOAT && (double)cg[k] < (double)lb[3 + k] + SP_EPSILON_FLOAT)) numerical = false;
    if (viol_any) numerical = false;
    return numerical ? DLSCQP_GOAL_FALLBACK : DLSCQP_INFEASIBLE;
}

struct Solver {
    // scratch of one thread, reused from agent to agent
    std::vector<SparseRow> rows;
    std::vector<char> skip;
    std::vector<double> x, Ga, dx, u, z, Zc, S, d1, rv, dual, blo, bhi;
    std::vector<int> act;

    int solve(const HostTables &tb, const dlscqp_agent &ag, const dlscqp_lsc_row *lsc, int K, const float *sfc, int max_slack_obs, const float *goal,
              float *out_ctrl, double *out_x, double *out_cost, double *out_slack_cost, int *out_iters) {
        const KParams &kp = tb.kp;
        const int dim = kp.dim, M = kp.M, n1 = kp.n1, n = n1 - 1, P = kp.P, f = kp.f;
        const int nxc = dim * P;
        int nso = ag.n_slack_obs;
        nso = nso > max_slack_obs ? max_slack_obs : (nso < 0 ? 0 : nso);
        const int nslk = M * nso, nx = nxc + nslk, r = dim * f + nslk;
        const int T = terminal_segments(kp, goal, ag.position, ag.nominal_velocity);
        if (out_iters) out_iters[0] = out_iters[1] = out_iters[2] = out_iters[3] = 0;
        if (!kp.t_valid[T - 1] || (nslk > 0 && !(kp.w_s > 0))) return DLSCQP_NUMERICAL;
        const int ldg = P | 1;
        const double *G = &tb.tabG[(size_t)(T - 1) * P * ldg], *N = &tb.tabN[(size_t)(T - 1) * P * f];
        const double *Xs = &tb.tabX0s[(size_t)(T - 1) * P * 3], *Xg = &tb.tabX0g[(size_t)(T - 1) * P];
        const double *NN = &tb.tabNN[(size_t)(T - 1) * P], *SIW = &tb.tabSIW[(size_t)(T - 1) * kp.n_slots];
        auto sig2 = [&](int s) { const double sg = kp.sigma[s % M]; return sg * sg; };
        // unconstrained minimiser
        x.assign(nx, 0.0);
        for (int k = 0; k < dim; k++)
            for (int p = 0; p < P; p++)
                x[k * P + p] = Xs[p * 3 + 0] * (double)ag.position[k] + Xs[p * 3 + 1] * (double)ag.velocity[k] + Xs[p * 3 + 2] * (double)ag.acceleration[k] +
                               Xg[p] * (double)goal[k];
        // bound table (traj_optimizer.cpp:237-270, 385-410, 453-513)
        blo.assign(kp.nb, 0.0); bhi.assign(kp.nb, 0.0);
        for (int k = 0; k < dim; k++)
            for (int m = 0; m < M; m++) {
                double lo = kp.world_min[k], hi = kp.world_max[k];
                if (kp.rsfc_mode && k == 2 && m == 0) { lo = -100.0; hi = 100.0; }
                if (kp.use_sfc && sfc) {
                    lo = std::max(lo, (double)sfc[m * 6 + k]);
                    hi = std::min(hi, (double)sfc[m * 6 + 3 + k]);
                }
                blo[k * M + m] = lo; bhi[k * M + m] = hi;
            }
        const double hv = ag.max_vel * kp.vel_scale, ha = ag.max_acc * kp.acc_scale;
        blo[dim * M] = -hv; bhi[dim * M] = hv;
        blo[dim * M + 1] = -ha; bhi[dim * M + 1] = ha;
        const double hc = 0.5 * kp.comm_range - ag.radius, cw = 0.5 * kp.comm_range - SP_EPSILON_FLOAT;
        blo[dim * M + 2] = -hc; bhi[dim * M + 2] = hc;
        for (int k = 0; k < dim; k++) { blo[dim * M + 3 + k] = (double)ag.next_waypoint[k] - cw; bhi[dim * M + 3 + k] = (double)ag.next_waypoint[k] + cw; }
        // every inequality row of the model, a . x >= rhs
        rows.clear();
        for (int j = 0; j < K; j++) {
            const dlscqp_lsc_row &R = lsc[j];
            const int m = R.m, i = R.i;
            if (m < 0 || m >= M || i < 0 || i >= n1) continue;
            if ((m == 0 && i < kp.phi) || f32_norm3d(R.normal[0], R.normal[1], R.normal[2]) < SP_EPSILON_FLOAT) continue;  // :417-424
            SparseRow a;
            a.n = 0;
            double rhs = R.d, nsq = 0;
            for (int k = 0; k < dim; k++) {
                a.idx[a.n] = k * P + m * n1 + i; a.cf[a.n] = (double)R.normal[k]; a.n++;
                rhs += (double)R.normal[k] * (double)R.point[k];
                nsq += (double)R.normal[k] * (double)R.normal[k];
            }
            double wsq = nsq * NN[m * n1 + i];
            if (R.slack >= 0 && R.slack < nso) {
                a.idx[a.n] = nxc + R.slack * M + m; a.cf[a.n] = -1.0; a.n++;
                wsq += sig2(R.slack * M + m);
            }
            if (!(wsq > 0)) continue;
            a.rhs = rhs; a.iw = 1.0 / sqrt(wsq); a.scale = 1.0; a.src = j;
            rows.push_back(a);
        }
        for (int s = 0; s < kp.n_slots; s++) {
            const Slot &sl = tb.slots[s];
            if (!(SIW[s] > 0)) continue;
            for (int side = 0; side < 2; side++) {
                SparseRow a;
                a.n = 0;
                const double sg = side == 0 ? 1.0 : -1.0;
                for (int e = 0; e < 3; e++)
                    if (sl.idx[e] >= 0) { a.idx[a.n] = sl.idx[e]; a.cf[a.n] = sg * (double)sl.coef[e]; a.n++; }
                a.rhs = side == 0 ? blo[sl.bsel] : -bhi[sl.bsel];
                a.iw = SIW[s];
                a.scale = sl.bsel == dim * M ? 1.0 / kp.vel_scale : (sl.bsel == dim * M + 1 ? 1.0 / kp.acc_scale : 1.0);
                a.src = -1;
                rows.push_back(a);
            }
        }
        for (int s = 0; s < nslk; s++) {  // eps <= 0 (:276)
            if (!(kp.sigma[s % M] > 0)) continue;
            SparseRow a;
            a.n = 1; a.idx[0] = nxc + s; a.cf[0] = -1.0; a.rhs = 0.0; a.iw = 1.0 / kp.sigma[s % M]; a.scale = 1.0; a.src = -1;
            rows.push_back(a);
        }
        const int R_ = (int)rows.size();
        skip.assign(R_, 0);
        // active set
        Zc.assign((size_t)r * nx, 0.0); S.assign((size_t)r * r, 0.0);
        dual.assign(r, 0.0); act.assign(r, -1);
        Ga.assign(nx, 0.0); dx.assign(nx, 0.0); u.assign(nx, 0.0); d1.assign(r, 0.0); rv.assign(r, 0.0); z.assign(f, 0.0);
        int q = 0, iters = 0, n_add = 0, n_drop = 0, status = DLSCQP_OK;
        double suspect_now = DEP_SUSPECT_SLACK;  // raised by a tiny pivot (qp_types.hpp: DEP_PIVOT)
        const int max_iter = 20 * (r + 8) + 200;
        auto value = [&](const SparseRow &a) { double v = 0; for (int e = 0; e < a.n; e++) v += a.cf[e] * x[a.idx[e]]; return v; };
#ifdef SKIP_STUDY
        const int nblk = (K + 64 * g_study_gran - 1) / (64 * g_study_gran);
        std::vector<double> bmin(nblk, -1e300), bmark(nblk, 0.0);
        double path = 0.0;
        int pass_no = 0;
        g_study.agents += 1;
#endif
        for (;;) {
#ifdef SKIP_STUDY
            {
                g_study.passes += 1;
                for (int b = g_study_first / g_study_gran; b < nblk; b++) {
                    g_study.blocks += 1;
                    if (pass_no >= 2) g_study.blocks_late += 1;
                    if (bmin[b] - (path - bmark[b]) > 1e-7) continue;  // no row of the block can be violated
                    g_study.priced += 1;
                    if (pass_no >= 2) g_study.priced_late += 1;
                    double mn = 1e300;
                    for (int j = 0; j < R_; j++) {
                        if (rows[j].src < 0 || rows[j].src / (64 * g_study_gran) != b || skip[j]) continue;
                        double v = 0;
                        for (int e = 0; e < rows[j].n; e++) v += rows[j].cf[e] * x[rows[j].idx[e]];
                        mn = std::min(mn, (v - (rows[j].rhs - VIOL_TOL)) * rows[j].iw);
                    }
                    bmin[b] = mn; bmark[b] = path;
                }
                pass_no++;
            }
#endif
            int ip = -1;
            double best = 0.0;
            for (int j = 0; j < R_; j++) {
                if (skip[j]) continue;
                const double sc = ((rows[j].rhs - VIOL_TOL) - value(rows[j])) * rows[j].iw;
                if (sc > best) { best = sc; ip = j; }
            }
            if (ip < 0) break;
            const SparseRow &a = rows[ip];
            double viol = a.rhs - value(a);
            // G a
            std::fill(Ga.begin(), Ga.end(), 0.0);
            for (int e = 0; e < a.n; e++) {
                const int j = a.idx[e];
                if (j < nxc) {
                    const int k = j / P, p = j - k * P;
                    const double *g = G + (size_t)p * ldg;
                    double *o = &Ga[k * P];
                    for (int p2 = 0; p2 < P; p2++) o[p2] += a.cf[e] * g[p2];
                } else {
                    Ga[j] += a.cf[e] * sig2(j - nxc);
                }
            }
            double dall = 0;
            for (int e = 0; e < a.n; e++) dall += a.cf[e] * Ga[a.idx[e]];
            double uq = 0.0;
            bool resolved = false;
            while (!resolved) {
                if (++iters > max_iter) { status = DLSCQP_ITERATION_LIMIT; break; }
                for (int k = 0; k < q; k++) {
                    const double *zk = &Zc[(size_t)k * nx];
                    double s = 0;
                    for (int e = 0; e < a.n; e++) s += a.cf[e] * zk[a.idx[e]];
                    d1[k] = s;
                }
                for (int i = 0; i < q; i++) {
                    double s = 0;
                    for (int k = i; k < q; k++) s += S[(size_t)i * r + k] * d1[k];
                    rv[i] = s;
                }
                dx = Ga;
                for (int k = 0; k < q; k++) {
                    const double *zk = &Zc[(size_t)k * nx];
                    const double w = d1[k];
                    for (int j = 0; j < nx; j++) dx[j] -= w * zk[j];
                }
                double dd2 = 0;
                for (int e = 0; e < a.n; e++) dd2 += a.cf[e] * dx[a.idx[e]];
                double noise2 = 0.0;
                if (q > 0 && !(dd2 > suspect_now * dall)) {
                    // measured again in the whitened space: u = a - A' rv, z = N' u (gi_core.hpp: whitened_length)
                    std::fill(u.begin(), u.end(), 0.0);
                    for (int e = 0; e < a.n; e++) u[a.idx[e]] += a.cf[e];
                    for (int i = 0; i < q; i++) {
                        const SparseRow &b = rows[act[i]];
                        for (int e = 0; e < b.n; e++) u[b.idx[e]] -= rv[i] * b.cf[e];
                    }
                    double zz = 0;
                    for (int k = 0; k < dim; k++)
                        for (int c = 0; c < f; c++) {
                            double s = 0, sn = 0;
                            for (int p = 0; p < P; p++) { s += N[(size_t)p * f + c] * u[k * P + p]; sn += fabs(N[(size_t)p * f + c]) * fabs(u[k * P + p]); }
                            zz += s * s; noise2 += sn * sn;
                        }
                    for (int s = 0; s < nslk; s++) { const double v = kp.sigma[s % M] * u[nxc + s]; zz += v * v; noise2 += v * v; }
                    dd2 = noise2 > DEP_XSPACE * dall ? zz : 0.0;
                }
                double t1 = 1e300;
                int ldrop = -1;
                for (int k = 0; k < q; k++)
                    if (rv[k] > 0) { const double t = dual[k] / rv[k]; if (t < t1) { t1 = t; ldrop = k; } }
                const bool dependent = !(dd2 > DEP_TOL * dall) || !(dd2 > DEP_NOISE * noise2) || !(dd2 > 0) || q >= r;
                const double t2 = dependent ? 1e300 : viol / dd2;
                if (t1 >= 1e300 && t2 >= 1e300) {
                    if (viol * a.scale <= CPLEX_FEAS_TOL) { skip[ip] = 1; resolved = true; break; }
                    status = DLSCQP_INFEASIBLE;
                    break;
                }
                const double t = t1 < t2 ? t1 : t2;
                if (!dependent) {
                    for (int j = 0; j < nx; j++) x[j] += t * dx[j];
                    viol -= t * dd2;
#ifdef SKIP_STUDY
                    path += 1.02 * t * sqrt(dd2);
#endif
                }
                for (int k = 0; k < q; k++) dual[k] -= t * rv[k];
                uq += t;
                if (!dependent && t2 <= t1) {
                    const double ia = 1.0 / sqrt(dd2);
                    {
                        double by_pivot = kp.dep_pivot * (dall * ia * ia);
                        by_pivot = by_pivot >= 1.0 ? 1e300 : by_pivot;
                        if (by_pivot > suspect_now) suspect_now = by_pivot;
                    }
                    for (int i = 0; i < q; i++) S[(size_t)i * r + q] = -rv[i] * ia;
                    S[(size_t)q * r + q] = ia;
                    double *zq = &Zc[(size_t)q * nx];
                    for (int j = 0; j < nx; j++) zq[j] = dx[j] * ia;
                    dual[q] = uq; act[q] = ip; skip[ip] = 1;
                    q++; n_add++;
                    resolved = true;
                } else {
                    // drop active row ldrop: delete row l of S, rotate columns (t, t+1) so that row l's entries are swept away
                    const int l = ldrop;
                    skip[act[l]] = 0;
#ifdef SKIP_STUDY
                    if (rows[act[l]].src >= 0) bmin[rows[act[l]].src / (64 * g_study_gran)] = -1e300;  // back among the priced rows with slack 0
#endif
                    double carry = 0;  // running |s_l..s_t|, the first pivot keeps its sign
                    for (int t2i = l; t2i < q - 1; t2i++) {
                        const double a0 = t2i == l ? S[(size_t)l * r + l] : carry, b0 = S[(size_t)l * r + t2i + 1];
                        const double sig = sqrt(a0 * a0 + b0 * b0);
                        double c = 1.0, s = 0.0;
                        if (sig > 0) { c = b0 / sig; s = a0 / sig; }
                        // new column t = c * col_t - s * col_{t+1};  carried column = s * col_t + c * col_{t+1}
                        for (int i = 0; i <= t2i + 1; i++) {
                            const double y = i <= t2i ? S[(size_t)i * r + t2i] : 0.0, w = S[(size_t)i * r + t2i + 1];  // (S is upper triangular)
                            S[(size_t)i * r + t2i] = c * y - s * w;
                            S[(size_t)i * r + t2i + 1] = s * y + c * w;
                        }
                        double *zt = &Zc[(size_t)t2i * nx], *zn = &Zc[(size_t)(t2i + 1) * nx];
                        for (int j = 0; j < nx; j++) {
                            const double y = zt[j], w = zn[j];
                            zt[j] = c * y - s * w;
                            zn[j] = s * y + c * w;
                        }
                        carry = sig;
                    }
                    // row l of S is now (0 .. 0, +-|row|) in the last column: drop that column and the row
                    for (int k = 0; k < q - 1; k++)
                        for (int i = l; i < q - 1; i++) S[(size_t)i * r + k] = S[(size_t)(i + 1) * r + k];
                    for (int k = l; k < q - 1; k++) { dual[k] = dual[k + 1]; act[k] = act[k + 1]; }
                    q--; n_drop++;
                }
            }
            if (status != DLSCQP_OK) break;
        }
        if (out_iters) { out_iters[0] = iters; out_iters[1] = n_add; out_iters[2] = n_drop; out_iters[3] = q; }
        if (status != DLSCQP_OK) return status;
        if (out_ctrl)
            for (int p = 0; p < P; p++)
                for (int k = 0; k < 3; k++) out_ctrl[p * 3 + k] = k < dim ? (float)x[k * P + p] : (float)kp.z_2d;
        if (out_x) for (int j = 0; j < nxc + M * max_slack_obs; j++) out_x[j] = j < nx ? x[j] : 0.0;
        if (out_cost) {  // traj_optimizer.cpp:286-331 (no 1/2), cplex.getObjValue()
            double c = 0;
            for (int k = 0; k < dim; k++)
                for (int m = 0; m < M; m++) {
                    const double *seg = &x[k * P + m * n1];
                    for (int i = 0; i < n1; i++) {
                        double t = 0;
                        for (int j = 0; j < n1; j++) t += kp.Qb[i * n1 + j] * seg[j];
                        c += seg[i] * t;
                    }
                    if (m >= M - T) { const double e = seg[n] - (double)goal[k]; c += kp.w_t * e * e; }
                }
            for (int s = 0; s < nslk; s++) c += kp.w_s * ((double)(M - s % M) / M) * x[nxc + s] * x[nxc + s];
            *out_cost = c;
        }
        if (out_slack_cost)
            for (int o = 0; o < max_slack_obs; o++) {
                double sc = 0;
                if (o < nso) for (int m = 0; m < M; m++) sc += fabs(x[nxc + o * M + m]);
                out_slack_cost[o] = sc;
            }
        return DLSCQP_OK;
    }
};

}  // namespace

#ifdef SKIP_STUDY
extern "C" void native_study(int first_block, int gran, double *out) {
    if (out) memcpy(out, &g_study, sizeof(g_study));
    g_study = SkipStudy();
    g_study_first = first_block; g_study_gran = gran;
}
#endif

// one agent after another: the interface of tests/emul's emul_plan_batch
extern "C" int native_plan_batch(const dlscqp_config *cfg, int n_agents, const dlscqp_agent *agents, const int64_t *row_offsets,
                                 const dlscqp_lsc_row *rows, const float *sfc, int max_slack_obs, int do_goal, int do_traj, float *out_goal,
                                 int *out_goal_status, float *out_ctrl, double *out_x, double *out_cost, double *out_slack_cost, int *out_status,
                                 int *out_iters) {
    HostTables tb;
    std::string err;
    if (!build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err)) return -1;
    const int M = cfg->M, n1 = cfg->n + 1, dim = cfg->world_dimension;
    const size_t xs = (size_t)dim * M * n1 + (size_t)M * max_slack_obs;
    Solver sv;
    for (int a = 0; a < n_agents; a++) {
        const dlscqp_lsc_row *r0 = rows + row_offsets[a];
        const int K = (int)(row_offsets[a + 1] - row_offsets[a]);
        const float *sf = sfc ? sfc + (size_t)a * M * 6 : nullptr;
        float goal[3] = {agents[a].current_goal_point[0], agents[a].current_goal_point[1], agents[a].current_goal_point[2]};
        int gst = DLSCQP_OK;
        if (do_goal) {
            gst = goal_lp(tb.kp, agents[a], r0, K, sf, goal);
            if (out_goal_status) out_goal_status[a] = gst;
            if (out_goal && gst != DLSCQP_INFEASIBLE) for (int k = 0; k < 3; k++) out_goal[a * 3 + k] = goal[k];
        }
        if (!do_traj) continue;
        int st;
        if (do_goal && gst == DLSCQP_INFEASIBLE) {
            st = DLSCQP_INFEASIBLE;  // the reference throws before trajOptimization is reached (traj_planner.cpp:491-496)
            if (out_iters) for (int k = 0; k < 4; k++) out_iters[a * 4 + k] = 0;
        } else {
            st = sv.solve(tb, agents[a], r0, K, sf, max_slack_obs, goal, out_ctrl ? out_ctrl + (size_t)a * M * n1 * 3 : nullptr, out_x ? out_x + a * xs : nullptr,
                          out_cost ? out_cost + a : nullptr, out_slack_cost ? out_slack_cost + (size_t)a * max_slack_obs : nullptr,
                          out_iters ? out_iters + (size_t)a * 4 : nullptr);
        }
        if (out_status) out_status[a] = st;
    }
    return 0;
}

// throughput: n_threads threads, each walking its own range of the batch again and again for `seconds`
extern "C" int native_bench(const dlscqp_config *cfg, int n_agents, const dlscqp_agent *agents, const int64_t *row_offsets, const dlscqp_lsc_row *rows,
                            const float *sfc, int max_slack_obs, int n_threads, double seconds, int64_t *n_done, double *elapsed_s) {
    HostTables tb;
    std::string err;
    if (!build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err)) return -1;
    const int M = cfg->M, n1 = cfg->n + 1;
    if (n_threads < 1) n_threads = 1;
    std::vector<int64_t> done(n_threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int t) {
        Solver sv;
        std::vector<float> ctrl((size_t)M * n1 * 3);
        std::vector<double> slack(std::max(max_slack_obs, 1));
        const int lo = (int)((int64_t)n_agents * t / n_threads), hi = (int)((int64_t)n_agents * (t + 1) / n_threads);
        if (hi <= lo) return;
        int a = lo;
        for (;;) {
            const dlscqp_lsc_row *r0 = rows + row_offsets[a];
            const int K = (int)(row_offsets[a + 1] - row_offsets[a]);
            const float *sf = sfc ? sfc + (size_t)a * M * 6 : nullptr;
            float goal[3];
            double cost;
            int it[4];
            const int gst = goal_lp(tb.kp, agents[a], r0, K, sf, goal);
            if (gst != DLSCQP_INFEASIBLE) sv.solve(tb, agents[a], r0, K, sf, max_slack_obs, goal, ctrl.data(), nullptr, &cost, max_slack_obs ? slack.data() : nullptr, it);
            done[t]++;
            if (++a == hi) a = lo;
            if ((done[t] & 15) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= seconds) break;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t tot = 0;
    for (auto v : done) tot += v;
    *n_done = tot;
    return 0;
}
