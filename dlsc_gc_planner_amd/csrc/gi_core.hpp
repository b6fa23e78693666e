// gi_core.hpp -- per-agent goal LP + trajectory QP, one agent per wavefront.
//
// Replaces GoalOptimizer::solve (goal_optimizer.cpp:7-198) and TrajOptimizer::solve
// (traj_optimizer.cpp:18-165, model at :225-527) of the reference.  The equalities are eliminated
// and the Hessian whitened on the host (qp_tables.cpp), which leaves, per agent, a projection
// problem  min 1/2 |w|^2  s.t. sparse half-spaces in control-point space, solved here with a
// Goldfarb-Idnani dual active-set method:
//   * Q (r x r, orthogonal) and R (packed upper triangle) of the active normals live in LDS;
//   * a constraint is added with one Householder reflection (a rank-1 update of the trailing
//     block of Q, no serial chain of rotations) and dropped with Givens rotations;
//   * every O(r), O(r^2) or O(rows) loop is a lane loop; reductions are wave-level butterflies.
//
// Compiled with -ffp-contract=off: the float32 expressions mirror octomap::point3d arithmetic of
// the reference and must not be fused; the hot FP64 loops call fma() explicitly.
//
// The code is written against a tiny "wave" interface (lane_loop / sync / reductions) so that
// the very same source also compiles as a lane-serial host build that the CPU-only unit tests
// use to check the algorithm (tests/emul); that build is test infrastructure and is never
// loaded by the product library.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/dlscqp.h"
#include "qp_types.hpp"

#ifndef DQ_FN
#define DQ_FN inline
#endif

namespace dlscqp {

struct Cand {
    double v;
    int code;
};

struct MemLayout {
    int r_max, nx_max, ldq, nflag;
    int oQ, oRp, oRinv, oN, oX, oNa, oD, oZ, oRv, oU, oBlo, oBhi;  // in doubles
    int oA;     // byte offset of the int active list
    int oFlag;  // byte offset of the row flags
    int bytes;
};

DQ_FN MemLayout make_layout(const KParams &kp, int max_slack_obs, int kmax) {
    MemLayout L;
    const int nslk = kp.M * max_slack_obs;
    L.r_max = kp.dim * kp.f + nslk;
    L.nx_max = kp.dim * kp.P + nslk;
    L.ldq = L.r_max | 1;  // odd leading dimension: conflict-free column and row sweeps
    L.nflag = kmax + 2 * kp.n_slots + nslk;
    int o = 0;
    L.oQ = o; o += L.r_max * L.ldq;
    L.oRp = o; o += L.r_max * (L.r_max + 1) / 2;
    L.oRinv = o; o += L.r_max;
    L.oN = o; o += kp.P * kp.f;
    L.oX = o; o += L.nx_max;
    L.oNa = o; o += L.r_max;
    L.oD = o; o += L.r_max;
    L.oZ = o; o += L.r_max;
    L.oRv = o; o += L.r_max;
    L.oU = o; o += L.r_max + 1;
    L.oBlo = o; o += kp.nb;
    L.oBhi = o; o += kp.nb;
    L.oA = o * 8;
    L.oFlag = L.oA + 4 * (L.r_max + 1);
    L.bytes = (L.oFlag + L.nflag + 15) & ~15;
    return L;
}

// octomath::Vector3 semantics: float32 products and sum, sqrt in double
DQ_FN double f32_norm3(float a, float b, float c) {
    float sq = a * a + b * b + c * c;
    return sqrt((double)sq);
}

#define DQ_RP(i, j) Rp[(j) * ((j) + 1) / 2 + (i)]

template <class W>
DQ_FN void plan_agent(const W &w, const KParams &kp, const MemLayout &lay, unsigned char *mem,
                      const dlscqp_agent *agp, const RowsView &rows, int row_begin, int K,
                      const float *sfc, int max_slack_obs, int do_goal, int do_traj,
                      float *out_goal, int *out_goal_status, float *out_ctrl, double *out_x,
                      double *out_cost, double *out_slack_cost, int *out_status, int *out_iters) {
    const int dim = kp.dim, M = kp.M, n1 = kp.n1, P = kp.P, f = kp.f;
    const int n = n1 - 1;
    // ---- agent record (uniform) ---------------------------------------------------------------
    float pos[3], vel[3], acc[3], cg[3], wp[3];
    for (int k = 0; k < 3; k++) {
        pos[k] = agp->position[k];
        vel[k] = agp->velocity[k];
        acc[k] = agp->acceleration[k];
        cg[k] = agp->current_goal_point[k];
        wp[k] = agp->next_waypoint[k];
    }
    const double max_vel = agp->max_vel, max_acc = agp->max_acc, radius = agp->radius;
    const double nominal_velocity = agp->nominal_velocity;
    int nso = agp->n_slack_obs;
    if (nso > max_slack_obs) nso = max_slack_obs;
    if (nso < 0) nso = 0;
    const int nslk = M * nso;

    const float *Rnx = rows.nx + row_begin, *Rny = rows.ny + row_begin, *Rnz = rows.nz + row_begin;
    const float *Rpx = rows.px + row_begin, *Rpy = rows.py + row_begin, *Rpz = rows.pz + row_begin;
    const double *Rd = rows.d + row_begin;
    const uint32_t *Rkey = rows.key + row_begin;

    // ================================ goal LP ===================================================
    // min t in [0, 1+1e-5] s.t. a_r t + b_r >= 0 (goal_optimizer.cpp:138-198), closed form.
    float goal[3] = {cg[0], cg[1], cg[2]};
    int goal_status = DLSCQP_OK;
    if (do_goal) {
        float dx = cg[0] - wp[0], dy = cg[1] - wp[1], dz = cg[2] - wp[2];
        if (f32_norm3(dx, dy, dz) < SP_EPSILON_FLOAT) {  // goal_optimizer.cpp:12-14
            goal[0] = wp[0]; goal[1] = wp[1]; goal[2] = wp[2];
        } else {
            const float diff[3] = {dx, dy, dz};
            double t_lo = 0.0, t_hi = 1.0 + SP_EPSILON_FLOAT;
            int infeasible = 0;
            if (kp.use_sfc && sfc) {  // goal_optimizer.cpp:158-173
                const float *bmin = sfc + (M - 1) * 6, *bmax = bmin + 3;
                for (int k = 0; k < dim; k++) {
                    for (int side = 0; side < 2; side++) {
                        double nr = side == 0 ? 1.0 : -1.0;
                        double dd = side == 0 ? (double)bmin[k] : -(double)bmax[k];
                        double a = nr * (double)diff[k], b = nr * (double)wp[k] - dd;
                        if (a > 0) { if (-b / a > t_lo) t_lo = -b / a; }
                        else if (a < 0) { if (-b / a < t_hi) t_hi = -b / a; }
                        else if (b < -CPLEX_FEAS_TOL) infeasible = 1;
                    }
                }
            }
            // goal_optimizer.cpp:176-195: last control point of the last segment, static obstacles
            auto lp_row = [&](int j, double &a, double &b) -> bool {
                uint32_t key = Rkey[j];
                int m = key & 0xFF, i = (key >> 8) & 0xF, dyn = (key >> 12) & 1;
                if (dyn || m != M - 1 || i != n) return false;
                float nx = Rnx[j], ny = Rny[j], nz = Rnz[j];
                if (f32_norm3(nx, ny, nz) < SP_EPSILON_FLOAT) return false;
                float ox = wp[0] - Rpx[j], oy = wp[1] - Rpy[j];
                a = (double)nx * (double)diff[0] + (double)ny * (double)diff[1];
                b = -Rd[j] + (double)nx * (double)ox + (double)ny * (double)oy;
                if (dim == 3) {
                    float oz = wp[2] - Rpz[j];
                    a += (double)nz * (double)diff[2];
                    b += (double)nz * (double)oz;
                }
                return true;
            };
            double lo2 = w.max(K, [&](int j) -> double {
                double a, b;
                if (!lp_row(j, a, b)) return -1e300;
                return a > 0 ? -b / a : -1e300;
            });
            double hi2 = w.min(K, [&](int j) -> double {
                double a, b;
                if (!lp_row(j, a, b)) return 1e300;
                return a < 0 ? -b / a : 1e300;
            });
            double bad = w.max(K, [&](int j) -> double {
                double a, b;
                if (!lp_row(j, a, b)) return 0.0;
                return (a == 0 && b < -CPLEX_FEAS_TOL) ? 1.0 : 0.0;
            });
            if (lo2 > t_lo) t_lo = lo2;
            if (hi2 < t_hi) t_hi = hi2;
            if (bad > 0.5 || t_lo > t_hi + CPLEX_FEAS_TOL) infeasible = 1;
            if (!infeasible) {
                float tf = (float)t_lo;  // Vector3::operator*(float), goal_optimizer.cpp:51
                for (int k = 0; k < 3; k++) goal[k] = diff[k] * tf + wp[k];
            } else {
                // exception branch, goal_optimizer.cpp:54-81
                int numerical_error = 1;
                if (kp.use_sfc && sfc) {
                    const float *bmin = sfc + (M - 1) * 6, *bmax = bmin + 3;
                    for (int k = 0; k < 3; k++)
                        if (!((double)cg[k] > (double)bmin[k] - SP_EPSILON_FLOAT &&
                              (double)cg[k] < (double)bmax[k] + SP_EPSILON_FLOAT))
                            numerical_error = 0;
                }
                double viol = w.max(K, [&](int j) -> double {
                    uint32_t key = Rkey[j];
                    int m = key & 0xFF, i = (key >> 8) & 0xF, dyn = (key >> 12) & 1;
                    if (dyn || m != M - 1 || i != n) return 0.0;
                    float nx = Rnx[j], ny = Rny[j], nz = Rnz[j];
                    if (f32_norm3(nx, ny, nz) < SP_EPSILON_FLOAT) return 0.0;
                    float e0 = cg[0] - Rpx[j], e1 = cg[1] - Rpy[j], e2 = cg[2] - Rpz[j];
                    float dot = nx * e0 + ny * e1 + nz * e2;
                    double delta = (double)dot - Rd[j];
                    return delta < -SP_EPSILON_FLOAT ? 1.0 : 0.0;
                });
                if (viol > 0.5) numerical_error = 0;
                if (numerical_error) {
                    goal_status = DLSCQP_GOAL_FALLBACK;  // goal = current_goal_point
                } else {
                    goal_status = DLSCQP_INFEASIBLE;
                }
            }
        }
        w.lane_loop(3, [&](int k) { if (out_goal && goal_status != DLSCQP_INFEASIBLE) out_goal[k] = goal[k]; });
        w.lane_loop(1, [&](int) { if (out_goal_status) *out_goal_status = goal_status; });
        if (goal_status == DLSCQP_INFEASIBLE) {
            // the reference throws before trajOptimization is reached (traj_planner.cpp:491-496)
            w.lane_loop(1, [&](int) { if (do_traj && out_status) *out_status = DLSCQP_INFEASIBLE; });
            return;
        }
    }
    if (!do_traj) return;

    // ================================ trajectory QP =============================================
    double *base = reinterpret_cast<double *>(mem);
    double *Q = base + lay.oQ, *Rp = base + lay.oRp, *Rinv = base + lay.oRinv, *Nl = base + lay.oN;
    double *x = base + lay.oX, *na = base + lay.oNa, *d = base + lay.oD, *z = base + lay.oZ;
    double *rv = base + lay.oRv, *u = base + lay.oU, *blo = base + lay.oBlo, *bhi = base + lay.oBhi;
    int *A = reinterpret_cast<int *>(mem + lay.oA);
    unsigned char *flag = mem + lay.oFlag;
    const int ldq = lay.ldq;
    const int rc = dim * f;      // whitened control-point dimension
    const int r = rc + nslk;     // whitened dimension
    const int nxc = dim * P;     // control-point variables
    const int nflag = K + 2 * kp.n_slots + nslk;

    // terminal segments, traj_optimizer.cpp:543-551 (float32 point difference)
    int T;
    {
        double ift = f32_norm3(goal[0] - pos[0], goal[1] - pos[1], goal[2] - pos[2]) / nominal_velocity;
        T = (int)((M * kp.dt - ift + SP_EPSILON) / kp.dt);
        if (T < 1) T = 1;
        if (T > M) T = M;
    }
    int status = DLSCQP_OK;
    if (!kp.t_valid[T - 1]) status = DLSCQP_NUMERICAL;
    if (nslk > 0 && !(kp.w_s > 0)) status = DLSCQP_NUMERICAL;
    if (status != DLSCQP_OK) {
        w.lane_loop(1, [&](int) { if (out_status) *out_status = status; });
        return;
    }

    const double *tN = kp.tabN + (size_t)(T - 1) * P * f;
    const double *tXs = kp.tabX0s + (size_t)(T - 1) * P * 3;
    const double *tXg = kp.tabX0g + (size_t)(T - 1) * P;
    w.lane_loop(P * f, [&](int j) { Nl[j] = tN[j]; });
    // unconstrained minimiser (start of the dual method)
    w.lane_loop(nxc, [&](int j) {
        int k = j / P, p = j - k * P;
        x[j] = tXs[p * 3 + 0] * (double)pos[k] + tXs[p * 3 + 1] * (double)vel[k] +
               tXs[p * 3 + 2] * (double)acc[k] + tXg[p] * (double)goal[k];
    });
    w.lane_loop(nslk, [&](int s) { x[nxc + s] = 0.0; });
    w.lane_loop(r * ldq, [&](int j) {
        int i = j / ldq, c = j - i * ldq;
        Q[j] = (i == c) ? 1.0 : 0.0;
    });
    w.lane_loop(nflag, [&](int j) { flag[j] = 0; });
    // per-agent bound table: boxes (world ∩ SFC), velocity, acceleration, comm range, waypoint
    w.lane_loop(kp.nb, [&](int b) {
        double lo, hi;
        if (b < dim * M) {
            int k = b / M, m = b - k * M;
            lo = kp.world_min[k];
            hi = kp.world_max[k];
            if (kp.rsfc_mode && k == 2 && m == 0) { lo = -100; hi = 100; }  // traj_optimizer.cpp:254-258
            if (kp.use_sfc && sfc) {  // Box::convertToLSCs, collision_constraints.cpp:66-87
                double smin = (double)sfc[m * 6 + k], smax = (double)sfc[m * 6 + 3 + k];
                if (smin > lo) lo = smin;
                if (smax < hi) hi = smax;
            }
        } else if (b == dim * M) {
            hi = max_vel * kp.vel_scale; lo = -hi;
        } else if (b == dim * M + 1) {
            hi = max_acc * kp.acc_scale; lo = -hi;
        } else if (b == dim * M + 2) {
            hi = 0.5 * kp.comm_range - radius; lo = -hi;  // traj_optimizer.cpp:495-500
        } else {
            int k = b - (dim * M + 3);
            double cw = 0.5 * kp.comm_range - SP_EPSILON_FLOAT;  // traj_optimizer.cpp:507-510
            lo = (double)wp[k] - cw; hi = (double)wp[k] + cw;
        }
        blo[b] = lo;
        bhi[b] = hi;
    });
    w.sync();

    int q = 0, iters = 0, n_add = 0, n_drop = 0;
    const int max_iter = kp.max_iter > 0 ? kp.max_iter : 20 * (r + 8) + 200;
    const int n_items = K + kp.n_slots + nslk;
    const Slot *slots = kp.slots;

    for (;;) {
        // ---- most violated inactive row ---------------------------------------------------------
        Cand best = w.argmax(n_items, [&](int j) -> Cand {
            Cand c;
            c.v = -1e300;
            c.code = -1;
            if (j < K) {
                if (flag[j]) return c;
                uint32_t key = Rkey[j];
                int m = key & 0xFF, i = (key >> 8) & 0xF;
                if (m == 0 && i < kp.phi) return c;  // traj_optimizer.cpp:417-419
                float nx = Rnx[j], ny = Rny[j], nz = Rnz[j];
                if (f32_norm3(nx, ny, nz) < SP_EPSILON_FLOAT) return c;  // :422-424
                int p = m * n1 + i;
                double val = (double)nx * x[p] + (double)ny * x[P + p];
                double rhs = Rd[j] + (double)nx * (double)Rpx[j] + (double)ny * (double)Rpy[j];
                if (dim == 3) {
                    val += (double)nz * x[2 * P + p];
                    rhs += (double)nz * (double)Rpz[j];
                }
                uint32_t sl = key >> 16;
                if (sl != 0xFFFFu && (int)sl < nso) val -= x[nxc + (int)sl * M + m];
                c.v = rhs - val;
                c.code = j;
            } else if (j < K + kp.n_slots) {
                int s = j - K;
                Slot sl = slots[s];
                double val = 0;
                for (int t = 0; t < 3; t++)
                    if (sl.idx[t] >= 0) val += (double)sl.coef[t] * x[sl.idx[t]];
                double vlo = blo[sl.bsel] - val, vhi = val - bhi[sl.bsel];
                if (flag[K + 2 * s]) vlo = -1e300;
                if (flag[K + 2 * s + 1]) vhi = -1e300;
                if (vlo >= vhi) { c.v = vlo; c.code = K + 2 * s; }
                else { c.v = vhi; c.code = K + 2 * s + 1; }
            } else {
                int s = j - K - kp.n_slots;
                if (flag[K + 2 * kp.n_slots + s]) return c;
                c.v = x[nxc + s];  // eps <= 0
                c.code = K + 2 * kp.n_slots + s;
            }
            return c;
        });
        if (!(best.v > VIOL_TOL) || best.code < 0) break;  // optimal

        // ---- decode the row into sparse x-space entries (uniform) -------------------------------
        const int ip = best.code;
        int cnt = 0, ek[3] = {0, 0, 0}, ep[3] = {0, 0, 0}, sidx = -1;
        double cf[3] = {0, 0, 0}, row_scale = 1.0;
        int axes_mask = 0;
        if (ip < K) {
            uint32_t key = Rkey[ip];
            int m = key & 0xFF, i = (key >> 8) & 0xF;
            int p = m * n1 + i;
            float nn[3] = {Rnx[ip], Rny[ip], Rnz[ip]};
            for (int k = 0; k < dim; k++) {
                ek[cnt] = k; ep[cnt] = p; cf[cnt] = (double)nn[k]; cnt++;
                axes_mask |= 1 << k;
            }
            uint32_t sl = key >> 16;
            if (sl != 0xFFFFu && (int)sl < nso) sidx = (int)sl * M + m;
        } else if (ip < K + 2 * kp.n_slots) {
            int s = (ip - K) >> 1, side = (ip - K) & 1;
            Slot sl = slots[s];
            double sg = side == 0 ? 1.0 : -1.0;  // val >= lo   |   -val >= -hi
            for (int t = 0; t < 3; t++)
                if (sl.idx[t] >= 0) {
                    int k = sl.idx[t] / P;
                    ek[cnt] = k; ep[cnt] = sl.idx[t] - k * P; cf[cnt] = sg * (double)sl.coef[t]; cnt++;
                    axes_mask |= 1 << k;
                }
            row_scale = (double)sl.scale;
        } else {
            sidx = ip - K - 2 * kp.n_slots;
        }
        // whitened normal na = N' a  (slack part: -sigma)
        w.lane_loop(r, [&](int l) {
            double v = 0;
            if (l < rc) {
                int k = l / f, j = l - k * f;
                for (int e = 0; e < 3; e++)  // unused entries carry cf = 0
                    if (ek[e] == k) v = fma(cf[e], Nl[ep[e] * f + j], v);
            } else if (l - rc == sidx) {
                v = -kp.sigma[sidx % M];
            }
            na[l] = v;
        });
        w.sync();

        double viol = best.v;  // = -(a.x - b) > 0
        double uq = 0.0;
        bool resolved = false;
        while (!resolved) {
            if (++iters > max_iter) { status = DLSCQP_ITERATION_LIMIT; break; }
            // d = Q' na   (na is non-zero only in the touched axis blocks and one slack entry)
            w.lane_loop(r, [&](int j) {
                double s = 0;
                for (int k = 0; k < dim; k++)
                    if (axes_mask & (1 << k)) {
                        const int i0 = k * f;
                        for (int i = i0; i < i0 + f; i++) s = fma(Q[i * ldq + j], na[i], s);
                    }
                if (sidx >= 0) s = fma(Q[(rc + sidx) * ldq + j], na[rc + sidx], s);
                d[j] = s;
            });
            w.sync();
            // z = Q2 d2, |d2|^2, |d|^2
            w.lane_loop(r, [&](int i) {
                double s = 0;
                for (int j = q; j < r; j++) s = fma(Q[i * ldq + j], d[j], s);
                z[i] = s;
            });
            const double dd2 = w.sum(r, [&](int j) -> double { return j >= q ? d[j] * d[j] : 0.0; });
            const double dall = w.sum(r, [&](int j) -> double { return d[j] * d[j]; });
            // rv = R^{-1} d1  (column-oriented back substitution)
            w.lane_loop(q, [&](int i) { rv[i] = d[i]; });
            w.sync();
            for (int k = q - 1; k >= 0; k--) {
                const double rk = rv[k] * Rinv[k];
                w.sync();
                w.lane_loop(k + 1, [&](int i) {
                    if (i == k) rv[k] = rk;
                    else rv[i] = fma(-DQ_RP(i, k), rk, rv[i]);
                });
                w.sync();
            }
            // step lengths
            Cand r1 = w.argmin(q, [&](int k) -> Cand {
                Cand c;
                c.code = k;
                c.v = rv[k] > 0 ? u[k] / rv[k] : 1e300;
                return c;
            });
            const double t1 = (q > 0 && r1.code >= 0) ? r1.v : 1e300;
            const int l = r1.code;
            const bool dependent = !(dd2 > DEP_TOL * dall) || !(dd2 > 0);
            const double t2 = dependent ? 1e300 : viol / dd2;
            if (t1 >= 1e300 && t2 >= 1e300) {
                // no step possible: infeasible, unless within CPLEX's feasibility tolerance
                if (viol * row_scale <= CPLEX_FEAS_TOL) {
                    w.lane_loop(1, [&](int) { flag[ip] = 2; });
                    w.sync();
                    resolved = true;
                    break;
                }
                status = DLSCQP_INFEASIBLE;
                break;
            }
            const double t = t1 < t2 ? t1 : t2;
            if (!dependent) {
                // primal step  x += t N z
                w.lane_loop(nxc, [&](int j) {
                    int k = j / P, p = j - k * P;
                    double s = 0;
                    for (int c = 0; c < f; c++) s = fma(Nl[p * f + c], z[k * f + c], s);
                    x[j] = fma(t, s, x[j]);
                });
                w.lane_loop(nslk, [&](int s) { x[nxc + s] += t * kp.sigma[s % M] * z[rc + s]; });
                viol -= t * dd2;
            }
            w.lane_loop(q, [&](int k) { u[k] = fma(-t, rv[k], u[k]); });
            uq += t;
            w.sync();
            if (!dependent && t2 <= t1) {
                // ---- add row ip: Householder reflection of d2 onto e_1 ---------------------------
                const double dq = d[q];
                const double nrm = sqrt(dd2);
                const double alpha = dq > 0 ? -nrm : nrm;
                const double v0 = dq - alpha;
                const double beta = 1.0 / (nrm * (nrm + fabs(dq)));  // 2 / v'v
                w.lane_loop(r, [&](int i) {
                    double *qi = Q + i * ldq;
                    const double g = beta * (z[i] - alpha * qi[q]);
                    qi[q] = fma(-g, v0, qi[q]);
                    for (int j = q + 1; j < r; j++) qi[j] = fma(-g, d[j], qi[j]);
                });
                w.lane_loop(q + 1, [&](int i) {
                    if (i < q) DQ_RP(i, q) = d[i];
                    else {
                        DQ_RP(q, q) = alpha;
                        Rinv[q] = 1.0 / alpha;
                        u[q] = uq;
                        A[q] = ip;
                        flag[ip] = 1;
                    }
                });
                w.sync();
                q++;
                n_add++;
                resolved = true;
            } else {
                // ---- drop active row l: delete column l of R, restore the triangle with Givens ----
                w.lane_loop(1, [&](int) { flag[A[l]] = 0; });
                w.sync();
                for (int tcol = l; tcol < q - 1; tcol++) {
                    // rotation from old column tcol+1, rows (tcol, tcol+1)
                    const double a = DQ_RP(tcol, tcol + 1), b = DQ_RP(tcol + 1, tcol + 1);
                    const double rho = sqrt(a * a + b * b);
                    const double cs = rho > 0 ? a / rho : 1.0, sn = rho > 0 ? b / rho : 0.0;
                    w.sync();
                    w.lane_loop(q - 1 - tcol, [&](int cc) {  // old columns tcol+1 .. q-1
                        int c = tcol + 1 + cc;
                        double r0 = DQ_RP(tcol, c), r1v = DQ_RP(tcol + 1, c);
                        DQ_RP(tcol, c) = cs * r0 + sn * r1v;
                        DQ_RP(tcol + 1, c) = -sn * r0 + cs * r1v;
                    });
                    w.lane_loop(r, [&](int i) {
                        double *qi = Q + i * ldq;
                        double q0 = qi[tcol], q1 = qi[tcol + 1];
                        qi[tcol] = cs * q0 + sn * q1;
                        qi[tcol + 1] = -sn * q0 + cs * q1;
                    });
                    w.sync();
                }
                // shift columns l+1..q-1 left by one (packed storage: serial over columns)
                for (int c = l + 1; c < q; c++) {
                    w.lane_loop(c, [&](int i) { Rp[(c - 1) * c / 2 + i] = DQ_RP(i, c); });
                    w.sync();
                }
                w.lane_loop(1, [&](int) {
                    for (int c = l; c < q - 1; c++) {
                        u[c] = u[c + 1];
                        A[c] = A[c + 1];
                    }
                });
                w.sync();
                q--;
                w.lane_loop(q, [&](int k) { Rinv[k] = 1.0 / DQ_RP(k, k); });
                w.sync();
                n_drop++;
                // loop again with the same row ip (step 2(a) of Goldfarb-Idnani)
            }
        }
        if (status != DLSCQP_OK) break;
    }

    // ================================ read-out ==================================================
    w.lane_loop(1, [&](int) {
        if (out_status) *out_status = status;
        if (out_iters) {
            out_iters[0] = iters;
            out_iters[1] = n_add;
            out_iters[2] = n_drop;
            out_iters[3] = q;
        }
    });
    if (status != DLSCQP_OK) return;
    // traj_optimizer.cpp:69-83: point3d(float) control points, z = world_z_2d in 2-D
    if (out_ctrl)
        w.lane_loop(M * n1 * 3, [&](int j) {
            int p = j / 3, k = j - p * 3;
            out_ctrl[j] = k < dim ? (float)x[k * P + p] : (float)kp.z_2d;
        });
    if (out_x) w.lane_loop(nxc + nslk, [&](int j) { out_x[j] = x[j]; });
    // traj_optimizer.cpp:85-106
    if (out_slack_cost)
        w.lane_loop(max_slack_obs, [&](int s) {
            double sc = 0;
            if (s < nso)
                for (int m = 0; m < M; m++) sc += fabs(x[nxc + s * M + m]);
            out_slack_cost[s] = sc;
        });
    // cplex.getObjValue(), traj_optimizer.cpp:109, objective at :286-331
    if (out_cost) {
        double cj = w.sum(nxc, [&](int j) -> double {
            int k = j / P, p = j - k * P, m = p / n1, i = p - m * n1;
            const double *seg = x + k * P + m * n1;
            double s = 0;
            for (int c = 0; c < n1; c++) s += kp.Qb[i * n1 + c] * seg[c];
            double term = x[j] * s;
            if (i == n && m >= M - T) {
                double e = x[j] - (double)goal[k];
                term += kp.w_t * e * e;
            }
            return term;
        });
        double cs = w.sum(nslk, [&](int s) -> double {
            int m = s % M;
            double e = x[nxc + s];
            return kp.w_s * ((double)(M - m) / M) * e * e;
        });
        w.lane_loop(1, [&](int) { *out_cost = cj + cs; });
    }
}

#undef DQ_RP

}  // namespace dlscqp
