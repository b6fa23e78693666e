// model_rows.hpp -- the reference's QP of ONE agent written out row by row on the host (no device code).
//
// TrajOptimizer::populatebyrow (/root/reference/src/traj_optimizer.cpp:225-527) in its own variable and row order.  Two users, both
// the debugging aids of the reference's failure path:
//   * the model as a CPLEX-LP file: cplex.exportModel (traj_optimizer.cpp:45-52, :113) -> dlscqp_write_lp;
//   * the kernel's infeasibility certificate (dlscqp_device_batch.out_conflict) decoded into rows of that model, numbered like the
//     conflict refiner numbers them (:118-146: `infeas` = the rows of `con`, then for every variable its lower and its upper
//     bound) -> dlscqp_conflict_decode.
// Nothing here is on the planning path: the kernel never sees this dense form (gi_core.hpp assembles its rows implicitly).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dlscqp.h"
#include "qp_types.hpp"

namespace dlscqp {

struct ModelRow {
    int kind;  // DLSCQP_MODEL_*
    int n;
    int var[6];
    double coef[6];
    double lo, hi;  // lo <= sum coef x <= hi; equality rows: lo == hi; +-1e300: none
};

struct HostModel {
    int dim = 0, M = 0, n = 0, phi = 0, nv = 0, n_ctrl = 0, n_slack = 0, T = 0;
    double goal[3] = {0, 0, 0};
    std::vector<std::string> var_name;
    std::vector<double> lb, ub;
    std::vector<ModelRow> con;
    // where the rows the kernel can name sit in `con` (-1: not in the model)
    std::vector<int> sfc_index;   // [m][2 k + side][j]
    std::vector<int> lsc_index;   // [row of the agent's list]
    std::vector<int> vel_index;   // [k][m][i][0: + row, 1: - row]
    std::vector<int> acc_index;   // [k][m][i][0: +, 1: -]
    std::vector<int> comm_index;  // [k][mi][m][0: +, 1: -]
    std::vector<int> wp_index;    // [k][m][0: +, 1: -]
    std::vector<double> Qb;       // control_input_weight * Q_base, (n+1)^2
    double w_t = 0, w_s = 0;
};

constexpr double MODEL_INF = 1e300;

inline void model_add(HostModel &md, int kind, std::initializer_list<std::pair<int, double>> ent, double lo, double hi) {
    ModelRow r;
    r.kind = kind;
    r.n = 0;
    for (const auto &e : ent) {
        if (e.second == 0.0) continue;
        r.var[r.n] = e.first;
        r.coef[r.n] = e.second;
        r.n++;
    }
    r.lo = lo;
    r.hi = hi;
    md.con.push_back(r);
}

// terminal segments, traj_optimizer.cpp:543-551 (float32 point difference, as the kernel and the oracle have it)
inline int model_terminal_segments(const dlscqp_config &cfg, const dlscqp_agent &ag, const float *goal) {
    const float dx = goal[0] - ag.position[0], dy = goal[1] - ag.position[1], dz = goal[2] - ag.position[2];
    const float sq = dx * dx + dy * dy + dz * dz;
    const double ift = sqrt((double)sq) / ag.nominal_velocity;
    int T = (int)((cfg.M * cfg.dt - ift + SP_EPSILON) / cfg.dt);
    T = T < 1 ? 1 : T;
    return T > cfg.M ? cfg.M : T;
}

// goal: the point the terminal cost pulls to (agent.current_goal_point when the trajectory QP is solved alone; the goal LP's answer in the
// fused launch, traj_planner.cpp:494), or null = the agent's current_goal_point
inline bool build_host_model(const dlscqp_config &cfg, const KParams &kp, const dlscqp_agent &ag, const dlscqp_lsc_row *rows, int n_rows,
                             const float *sfc, int max_slack_obs, const float *goal, HostModel &md, std::string &err) {
    const int dim = cfg.world_dimension, M = cfg.M, n = cfg.n, phi = cfg.phi, N = n + 1;
    const double dt = cfg.dt;
    if (n != 5 || phi != 3) { err = "only n = 5, phi = 3 (traj_optimizer.cpp:209)"; return false; }
    md = HostModel();
    md.dim = dim; md.M = M; md.n = n; md.phi = phi;
    const int os_ = N, od = M * N, osl = dim * od;
    int nso = ag.n_slack_obs < 0 ? 0 : ag.n_slack_obs;
    nso = nso > max_slack_obs ? max_slack_obs : nso;
    md.n_ctrl = osl; md.n_slack = M * nso; md.nv = osl + M * nso;
    md.lb.assign(md.nv, -MODEL_INF);
    md.ub.assign(md.nv, MODEL_INF);
    md.var_name.resize(md.nv);
    const char axis[3] = {'x', 'y', 'z'};
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < N; i++) {  // :237-270
                const int v = k * od + m * os_ + i;
                char b[48];
                snprintf(b, sizeof b, "%c_%d_%d", axis[k], m, i);
                md.var_name[v] = b;
                double lo = cfg.world_min[k], up = cfg.world_max[k];
                if (k == 2 && m == 0 && cfg.planner_mode == DLSCQP_PLANNER_RECIPROCALRSFC) { lo = -100; up = 100; }
                if (!(m == 0 && i < 3)) { md.lb[v] = lo; md.ub[v] = up; }
            }
    std::vector<int> slack_obs(nso, -1);
    for (int r = 0; r < n_rows; r++)
        if (rows[r].slack >= 0 && rows[r].slack < nso && slack_obs[rows[r].slack] < 0) slack_obs[rows[r].slack] = rows[r].obs;
    for (int s = 0; s < nso; s++)
        for (int m = 0; m < M; m++) {  // :272-283
            char b[64];
            snprintf(b, sizeof b, "epsilon_slack_col_%d_%d", slack_obs[s] >= 0 ? slack_obs[s] : s, m);
            md.var_name[osl + M * s + m] = b;
            md.ub[osl + M * s + m] = 0.0;
        }
    const float g3[3] = {goal ? goal[0] : ag.current_goal_point[0], goal ? goal[1] : ag.current_goal_point[1], goal ? goal[2] : ag.current_goal_point[2]};
    for (int k = 0; k < 3; k++) md.goal[k] = g3[k];
    md.T = model_terminal_segments(cfg, ag, g3);
    md.Qb.assign(kp.Qb, kp.Qb + N * N);
    md.w_t = cfg.terminal_weight;
    md.w_s = cfg.slack_collision_weight;
    // ---- equality rows, :333-381 -------------------------------------------------------------------------
    const double c1 = pow(dt, -1) * n, c2 = pow(dt, -2) * n * (n - 1);
    for (int k = 0; k < dim; k++) {
        const int b0 = k * od, b1 = k * od + os_;
        model_add(md, DLSCQP_MODEL_EQ, {{b0, 1.0}}, ag.position[k], ag.position[k]);
        if (M > 1) model_add(md, DLSCQP_MODEL_EQ, {{b0 + n, 1.0}, {b1, -1.0}}, 0.0, 0.0);
        model_add(md, DLSCQP_MODEL_EQ, {{b0 + 1, c1}, {b0, -c1}}, ag.velocity[k], ag.velocity[k]);
        model_add(md, DLSCQP_MODEL_EQ, {{b0 + 2, c2}, {b0 + 1, -2 * c2}, {b0, c2}}, ag.acceleration[k], ag.acceleration[k]);
        model_add(md, DLSCQP_MODEL_EQ, {{b1 + 1, 1.0}, {b1, -1.0}, {b0 + n, -1.0}, {b0 + n - 1, 1.0}}, 0.0, 0.0);
        model_add(md, DLSCQP_MODEL_EQ, {{b1 + 2, 1.0}, {b1 + 1, -2.0}, {b1, 1.0}, {b0 + n, -1.0}, {b0 + n - 1, 2.0}, {b0 + n - 2, -1.0}}, 0.0, 0.0);
    }
    static const double A0[3][6] = {{1, 0, 0, 0, 0, 0}, {-1, 1, 0, 0, 0, 0}, {1, -2, 1, 0, 0, 0}};
    static const double AT[3][6] = {{0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, -1, 1}, {0, 0, 0, 1, -2, 1}};
    for (int k = 0; k < dim; k++)
        for (int m = 2; m < M; m++) {  // Aeq_base, :188-223, rows phi (m - 2) + j
            int nn = 1;
            for (int j = 0; j < phi; j++) {
                ModelRow r;
                r.kind = DLSCQP_MODEL_EQ; r.n = 0; r.lo = r.hi = 0.0;
                const double sc = pow(dt, -j) * nn;
                for (int l = 0; l < N; l++)
                    if (AT[j][l] != 0) { r.var[r.n] = k * od + (m - 1) * os_ + l; r.coef[r.n] = sc * AT[j][l]; r.n++; }
                for (int l = 0; l < N; l++)
                    if (A0[j][l] != 0) { r.var[r.n] = k * od + m * os_ + l; r.coef[r.n] = -sc * A0[j][l]; r.n++; }
                md.con.push_back(r);
                nn *= (n - j);
            }
        }
    // ---- corridor boxes, :385-410 (Box::convertToLSCs, collision_constraints.cpp:66-87: faces 2 k (min), 2 k + 1 (max)) ----------
    md.sfc_index.assign((size_t)M * 2 * dim * N, -1);
    if (cfg.world_use_octomap && sfc) {
        for (int m = 0; m < M; m++)
            for (int f = 0; f < 2 * dim; f++) {
                const int k = f / 2;
                const double sg = f % 2 == 0 ? 1.0 : -1.0;
                const double d = f % 2 == 0 ? (double)sfc[m * 6 + k] : -(double)sfc[m * 6 + 3 + k];
                for (int j = 0; j < N; j++) {
                    if (m == 0 && j < phi) continue;
                    md.sfc_index[((size_t)m * 2 * dim + f) * N + j] = (int)md.con.size();
                    model_add(md, DLSCQP_MODEL_SFC, {{k * od + m * os_ + j, sg}}, d, MODEL_INF);
                }
            }
    }
    // ---- LSC rows, :412-450 (the agent's list in populatebyrow's order: obstacle, segment, control point) -------------------------
    md.lsc_index.assign(n_rows, -1);
    for (int r = 0; r < n_rows; r++) {
        const dlscqp_lsc_row &row = rows[r];
        const int m = row.m, i = row.i;
        if (m < 0 || m >= M || i < 0 || i > n) continue;
        if (m == 0 && i < phi) continue;
        const float sq = row.normal[0] * row.normal[0] + row.normal[1] * row.normal[1] + row.normal[2] * row.normal[2];
        if (sqrt((double)sq) < SP_EPSILON_FLOAT) continue;
        ModelRow mr;
        mr.kind = DLSCQP_MODEL_LSC; mr.n = 0;
        double rhs = row.d;
        for (int k = 0; k < dim; k++) {
            mr.var[mr.n] = k * od + m * os_ + i; mr.coef[mr.n] = (double)row.normal[k]; mr.n++;
            rhs += (double)row.normal[k] * (double)row.point[k];
        }
        if (row.slack >= 0 && row.slack < nso) { mr.var[mr.n] = osl + M * row.slack + m; mr.coef[mr.n] = -1.0; mr.n++; }
        mr.lo = rhs; mr.hi = MODEL_INF;
        md.lsc_index[r] = (int)md.con.size();
        md.con.push_back(mr);
    }
    // ---- dynamic limits, :453-487 -------------------------------------------------------------------------
    md.vel_index.assign((size_t)dim * M * N * 2, -1);
    md.acc_index.assign((size_t)dim * M * N * 2, -1);
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++) {
            for (int i = 0; i < n; i++) {
                if (m == 0 && (i == 0 || i == 1)) continue;
                const int v = k * od + m * os_ + i;
                for (int s = 0; s < 2; s++) {
                    const double sg = s == 0 ? 1.0 : -1.0;
                    md.vel_index[(((size_t)k * M + m) * N + i) * 2 + s] = (int)md.con.size();
                    model_add(md, DLSCQP_MODEL_VEL, {{v + 1, sg * c1}, {v, -sg * c1}}, -MODEL_INF, ag.max_vel);
                }
            }
            for (int i = 0; i < n - 1; i++) {
                if (m == 0 && i == 0) continue;
                const int v = k * od + m * os_ + i;
                for (int s = 0; s < 2; s++) {
                    const double sg = s == 0 ? 1.0 : -1.0;
                    md.acc_index[(((size_t)k * M + m) * N + i) * 2 + s] = (int)md.con.size();
                    model_add(md, DLSCQP_MODEL_ACC, {{v + 2, sg * c2}, {v + 1, -2 * sg * c2}, {v, sg * c2}}, -MODEL_INF, ag.max_acc);
                }
            }
        }
    // ---- communication range, :490-513 --------------------------------------------------------------------
    md.comm_index.assign((size_t)dim * M * M * 2, -1);
    md.wp_index.assign((size_t)dim * M * 2, -1);
    if (cfg.communication_range > 0) {
        for (int k = 0; k < dim; k++)
            for (int mi = 0; mi < M; mi++)
                for (int m = mi; m < M; m++)
                    for (int s = 0; s < 2; s++) {
                        const double sg = s == 0 ? 1.0 : -1.0;
                        md.comm_index[(((size_t)k * M + mi) * M + m) * 2 + s] = (int)md.con.size();
                        model_add(md, DLSCQP_MODEL_COMM, {{k * od + m * os_ + n, sg}, {k * od + mi * os_, -sg}}, -MODEL_INF,
                                  0.5 * cfg.communication_range - ag.radius);
                    }
        for (int k = 0; k < dim; k++)
            for (int m = 0; m < M; m++)
                for (int s = 0; s < 2; s++) {
                    const double sg = s == 0 ? 1.0 : -1.0;
                    md.wp_index[((size_t)k * M + m) * 2 + s] = (int)md.con.size();
                    model_add(md, DLSCQP_MODEL_WAYPOINT, {{k * od + m * os_ + n, sg}}, -MODEL_INF,
                              0.5 * cfg.communication_range - SP_EPSILON_FLOAT + sg * (double)ag.next_waypoint[k]);
                }
    }
    // ---- stop at the end of the horizon, :515-524 -----------------------------------------------------------
    if (cfg.planner_mode == DLSCQP_PLANNER_DLSCGC || cfg.planner_mode == DLSCQP_PLANNER_LSC)
        for (int k = 0; k < dim; k++)
            for (int i = 1; i < phi; i++)
                model_add(md, DLSCQP_MODEL_EQ, {{k * od + (M - 1) * os_ + n, 1.0}, {k * od + (M - 1) * os_ + n - i, -1.0}}, 0.0, 0.0);
    return true;
}

// The model as CPLEX-LP text (what cplex.exportModel writes for the reference: same variables, same rows, same order; the row
// names are c0, c1, ... as the conflict refiner prints them).  The objective's constant (terminal_weight |goal|^2 per terminal segment)
// has no place in the format and is written as a comment.
inline bool write_lp(const HostModel &md, const char *path, std::string &err) {
    FILE *fp = fopen(path, "w");
    if (!fp) { err = std::string("cannot open ") + path; return false; }
    const int N = md.n + 1, od = md.M * N, osl = md.dim * od;
    fprintf(fp, "\\ dlsc_gc_planner_amd: TrajOptimizer::populatebyrow of one agent (traj_optimizer.cpp:225-527), written by dlscqp_write_lp\n");
    fprintf(fp, "\\ variables %d (control points %d, slack %d), rows %d, terminal segments %d\n", md.nv, md.n_ctrl, md.n_slack, (int)md.con.size(), md.T);
    double c0 = 0.0;
    fprintf(fp, "Minimize\n obj:");
    int terms = 0;
    auto brk = [&]() { if (++terms % 6 == 0) fprintf(fp, "\n  "); };
    for (int m = md.M - md.T; m < md.M; m++)
        for (int k = 0; k < md.dim; k++) {  // :301-314
            fprintf(fp, " %+.17g %s", -2.0 * md.w_t * md.goal[k], md.var_name[k * od + m * N + md.n].c_str());
            brk();
            c0 += md.w_t * md.goal[k] * md.goal[k];
        }
    fprintf(fp, " + [");
    for (int k = 0; k < md.dim; k++)
        for (int m = 0; m < md.M; m++)
            for (int i = 0; i < N; i++) {
                const int vi = k * od + m * N + i;
                double dii = 2.0 * md.Qb[i * N + i];
                if (i == md.n && m >= md.M - md.T) dii += 2.0 * md.w_t;
                if (dii != 0.0) { fprintf(fp, " %+.17g %s ^2", dii, md.var_name[vi].c_str()); brk(); }
                for (int j = i + 1; j < N; j++) {
                    const double q = 2.0 * (md.Qb[i * N + j] + md.Qb[j * N + i]);
                    if (q != 0.0) { fprintf(fp, " %+.17g %s * %s", q, md.var_name[vi].c_str(), md.var_name[k * od + m * N + j].c_str()); brk(); }
                }
            }
    for (int s = 0; s < md.n_slack; s++) {  // :317-330
        const int m = s % md.M;
        fprintf(fp, " %+.17g %s ^2", 2.0 * md.w_s * ((double)(md.M - m) / md.M), md.var_name[osl + s].c_str());
        brk();
    }
    fprintf(fp, " ] / 2\n\\ objective constant (not part of the format): %.17g\n", c0);
    fprintf(fp, "Subject To\n");
    for (size_t r = 0; r < md.con.size(); r++) {
        const ModelRow &row = md.con[r];
        fprintf(fp, " c%zu:", r);
        for (int t = 0; t < row.n; t++) fprintf(fp, " %+.17g %s", row.coef[t], md.var_name[row.var[t]].c_str());
        if (row.lo == row.hi) fprintf(fp, " = %.17g\n", row.lo);
        else if (row.lo > -MODEL_INF) fprintf(fp, " >= %.17g\n", row.lo);
        else fprintf(fp, " <= %.17g\n", row.hi);
    }
    fprintf(fp, "Bounds\n");
    for (int v = 0; v < md.nv; v++) {
        const bool nl = md.lb[v] <= -MODEL_INF, nu = md.ub[v] >= MODEL_INF;
        if (nl && nu) fprintf(fp, " %s free\n", md.var_name[v].c_str());
        else if (nl) fprintf(fp, " -inf <= %s <= %.17g\n", md.var_name[v].c_str(), md.ub[v]);
        else if (nu) fprintf(fp, " %s >= %.17g\n", md.var_name[v].c_str(), md.lb[v]);
        else fprintf(fp, " %.17g <= %s <= %.17g\n", md.lb[v], md.var_name[v].c_str(), md.ub[v]);
    }
    fprintf(fp, "End\n");
    fclose(fp);
    return true;
}

// One row id of a certificate (gi_core.hpp: conflict_id) -> the row of the reference's model it stands for.  scale_out: the factor
// from the kernel's scaling of that row (metres) to the reference's (n/dt for a velocity row, n(n-1)/dt^2 for an acceleration row):
// the multiplier in the reference's scaling is the kernel's divided by it.
inline bool conflict_member(const dlscqp_config &cfg, const KParams &kp, const Slot *slots, const HostModel &md, const float *sfc, uint32_t id,
                            dlscqp_conflict_member &out, double &scale_out, std::string &err) {
    const int dim = md.dim, M = md.M, n = md.n, N = n + 1, od = M * N, ncon = (int)md.con.size();
    const int type = (int)(id >> 24), index = (int)((id >> 3) & 0x1FFFFF), sub = (int)(id & 7);
    memset(&out, 0, sizeof out);
    scale_out = 1.0;
    int con_i = -1, var_i = -1, var_side = 0;
    auto bound_row = [&](int v, int side) {
        // lo <= x_v <= hi of the kernel = the tighter of the variable's bound (:251-265) and the corridor face (:385-410)
        const int k = v / od, p = v - k * od, m = p / N, i = p - m * N;
        const bool have_sfc = cfg.world_use_octomap && sfc;
        double wb = side == 0 ? md.lb[v] : md.ub[v];
        if (have_sfc) {
            const double sb = side == 0 ? (double)sfc[m * 6 + k] : (double)sfc[m * 6 + 3 + k];
            const bool sfc_binds = side == 0 ? sb > wb : sb < wb;
            if (sfc_binds) { con_i = md.sfc_index[((size_t)m * 2 * dim + 2 * k + side) * N + i]; return; }
        }
        var_i = v; var_side = side;
    };
    if (type == 0) {
        const int kind = sub >> 1, side = sub & 1, v = index;
        if (v < 0 || v >= md.n_ctrl) { err = "certificate: coordinate out of range"; return false; }
        const int k = v / od, p = v - k * od, m = p / N, i = p - m * N;
        if (kind == 0) bound_row(v, side);
        else if (kind == 1) { con_i = md.vel_index[(((size_t)k * M + m) * N + i) * 2 + (side == 0 ? 1 : 0)]; scale_out = 1.0 / kp.vel_scale; }
        else { con_i = md.acc_index[(((size_t)k * M + m) * N + i) * 2 + (side == 0 ? 1 : 0)]; scale_out = 1.0 / kp.acc_scale; }
    } else if (type == 1) {
        if (index < 0 || index >= (int)md.lsc_index.size()) { err = "certificate: LSC row out of range"; return false; }
        con_i = md.lsc_index[index];
    } else if (type == 2) {
        if (index < 0 || index >= kp.n_slots) { err = "certificate: slot out of range"; return false; }
        const Slot &sl = slots[index];
        const int side = sub & 1, v0 = sl.idx[0], k = v0 / od, p0 = v0 - k * od, m0 = p0 / N, i0 = p0 - m0 * N;
        scale_out = (double)sl.scale;
        // (the slots hold the HIGHER control point first: qp_tables.cpp)  side 0: value >= lo, i.e. the reference's "-" row
        if (sl.kind == SLOT_BOUND) bound_row(v0, side);
        else if (sl.kind == SLOT_VEL) con_i = md.vel_index[(((size_t)k * M + m0) * N + (i0 - 1)) * 2 + (side == 0 ? 1 : 0)];
        else if (sl.kind == SLOT_ACC) con_i = md.acc_index[(((size_t)k * M + m0) * N + (i0 - 2)) * 2 + (side == 0 ? 1 : 0)];
        else if (sl.kind == SLOT_COMM) {
            const int p1 = sl.idx[1] - k * od, mi = p1 / N;
            con_i = md.comm_index[(((size_t)k * M + mi) * M + m0) * 2 + (side == 0 ? 1 : 0)];
        } else con_i = md.wp_index[((size_t)k * M + m0) * 2 + (side == 0 ? 1 : 0)];
    } else {
        var_i = md.n_ctrl + index; var_side = 1;
        if (var_i >= md.nv) { err = "certificate: slack variable out of range"; return false; }
    }
    if (con_i >= 0) {
        const ModelRow &r = md.con[con_i];
        out.ref_index = con_i; out.kind = r.kind; out.n = r.n;
        for (int t = 0; t < r.n && t < 4; t++) { out.var[t] = r.var[t]; out.coef[t] = r.coef[t]; }
        out.lo = r.lo; out.hi = r.hi;
    } else if (var_i >= 0) {
        out.ref_index = ncon + 2 * var_i + var_side;  // infeas: con, then IloBound(var, Lower), IloBound(var, Upper) per variable
        out.kind = var_side == 0 ? DLSCQP_MODEL_VAR_LB : DLSCQP_MODEL_VAR_UB;
        out.n = 1; out.var[0] = var_i; out.coef[0] = 1.0;
        out.lo = var_side == 0 ? md.lb[var_i] : -MODEL_INF;
        out.hi = var_side == 1 ? md.ub[var_i] : MODEL_INF;
    } else {
        err = "certificate names a row that is not in the model";
        return false;
    }
    return true;
}

}  // namespace dlscqp
