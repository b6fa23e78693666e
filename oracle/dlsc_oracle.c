/*
 * dlsc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See dlsc_oracle.h.
 *
 * Every function cites the reference lines it restates (paths under /root/reference).
 * Compile with -ffp-contract=off: several float32 expressions mimic octomap::point3d.
 */
#include "dlsc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SP_EPSILON 1e-9       /* sp_const.hpp:3 */
#define SP_EPSILON_FLOAT 1e-5 /* sp_const.hpp:4 */

/* CPLEX 22.1.1 default feasibility tolerance (CPXPARAM_Simplex_Tolerances_Feasibility, 1e-6):
 * a row violated by no more than this in the model's own scaling is accepted by the reference's
 * solver, so an exact method must not report "infeasible" for it. */
#define CPLEX_FEAS_TOL 1e-6
/* violation (distance in x-space) below which a row is considered satisfied by the oracle */
#define ORACLE_VIOL_TOL 1e-9

/* ------------------------------------------------------------------------------------------ */
/* polynomial.hpp:9-20 */
static int n_choose_k(int n, int k) {
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

/* polynomial.hpp:89-99 */
static int coef_derivative(int n, int phi) {
    if (n < phi) return 0;
    int coef = 1;
    for (int i = 0; i < phi; i++) coef *= n - i;
    return coef;
}

/* polynomial.hpp:280-293: B(i,j) = C(n,i) C(n-i,n-j) (-1)^(j-i), j >= i */
static void build_bernstein_basis(int n, double *B) {
    int N = n + 1;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            B[i * N + j] = (j >= i) ? n_choose_k(n, i) * n_choose_k(n - i, n - j) * pow(-1, j - i) : 0.0;
}

/* traj_optimizer.cpp:172-187 */
void oracle_build_q_base(int n, int phi, int phi_n, double dt, double *Qb) {
    int N = n + 1;
    double *B = (double *)calloc((size_t)N * N, sizeof(double));
    double *Z = (double *)calloc((size_t)N * N, sizeof(double));
    double *T = (double *)calloc((size_t)N * N, sizeof(double));
    build_bernstein_basis(n, B);
    memset(Qb, 0, sizeof(double) * N * N);
    for (int k = phi; k > phi - phi_n; k--) {
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                Z[i * N + j] = 0;
                if (i + j - 2 * k + 1 > 0)
                    Z[i * N + j] = (double)coef_derivative(i, k) * coef_derivative(j, k) / (i + j - 2 * k + 1);
            }
        /* T = B * Z */
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int l = 0; l < N; l++) s += B[i * N + l] * Z[l * N + j];
                T[i * N + j] = s;
            }
        /* Z = T * B^T * dt^(-2k+1) */
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int l = 0; l < N; l++) s += T[i * N + l] * B[j * N + l];
                Qb[i * N + j] += s * pow(dt, -2 * k + 1);
            }
    }
    free(B);
    free(Z);
    free(T);
}

/* traj_optimizer.cpp:189-223 */
int oracle_build_aeq_base(int M, int n, int phi, double dt, double *Aeq_base) {
    if (!(n == 5 && phi == 3)) return -1;
    static const double A_0[6][6] = {{1, 0, 0, 0, 0, 0},     {-1, 1, 0, 0, 0, 0},
                                     {1, -2, 1, 0, 0, 0},    {-1, 3, -3, 1, 0, 0},
                                     {1, -4, 6, -4, 1, 0},   {-1, 5, -10, 10, -5, 1}};
    static const double A_T[6][6] = {{0, 0, 0, 0, 0, 1},     {0, 0, 0, 0, -1, 1},
                                     {0, 0, 0, 1, -2, 1},    {0, 0, -1, 3, -3, 1},
                                     {0, 1, -4, 6, -4, 1},   {-1, 5, -10, 10, -5, 1}};
    int N = n + 1, P = M * N;
    int rows = (M - 2) * phi;
    if (rows < 0) rows = 0;
    memset(Aeq_base, 0, sizeof(double) * (size_t)rows * P);
    for (int m = 2; m < M; m++) {
        int nn = 1;
        for (int j = 0; j < phi; j++) {
            for (int l = 0; l < N; l++) {
                Aeq_base[(phi * (m - 2) + j) * P + N * (m - 1) + l] = pow(dt, -j) * nn * A_T[j][l];
                Aeq_base[(phi * (m - 2) + j) * P + N * m + l] = -pow(dt, -j) * nn * A_0[j][l];
            }
            nn = nn * (n - j);
        }
    }
    return rows;
}

/* octomath::Vector3 (float) difference + norm(): float products and float sum, sqrt in double */
static double f32_dist(const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float sq = dx * dx + dy * dy + dz * dz;
    return sqrt((double)sq);
}
static double f32_norm(const float a[3]) {
    float sq = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    return sqrt((double)sq);
}

/* traj_optimizer.cpp:543-551 */
int oracle_terminal_segments(const dlscqp_config *cfg, const dlscqp_agent *ag) {
    double ideal_flight_time = f32_dist(ag->current_goal_point, ag->position) / ag->nominal_velocity;
    int ts = (int)((cfg->M * cfg->dt - ideal_flight_time + SP_EPSILON) / cfg->dt);
    return ts > 1 ? ts : 1;
}

void oracle_model_free(oracle_model *m) {
    if (!m) return;
    free(m->Q); free(m->c); free(m->Aeq); free(m->beq); free(m->Ain);
    free(m->lo); free(m->hi); free(m->lb); free(m->ub);
    free(m);
}

/* populatebyrow, traj_optimizer.cpp:225-527 */
oracle_model *oracle_model_build(const dlscqp_config *cfg, const dlscqp_agent *ag, int n_rows,
                                 const dlscqp_lsc_row *rows, const float *sfc) {
    const int dim = cfg->world_dimension, M = cfg->M, n = cfg->n, phi = cfg->phi;
    if (dim < 2 || dim > 3) return NULL;           /* :248 */
    if (!(n == 5 && phi == 3)) return NULL;        /* :209 */
    if (M < 2) return NULL;
    const double dt = cfg->dt;
    const int offset_seg = n + 1, offset_dim = M * (n + 1);
    const int offset_slack_col = dim * M * (n + 1);
    const int n_slack = M * ag->n_slack_obs;
    const int nv = offset_slack_col + n_slack;
    const int stop = (cfg->planner_mode == DLSCQP_PLANNER_DLSCGC || cfg->planner_mode == DLSCQP_PLANNER_LSC);

    /* count rows */
    int neq = dim * (M > 1 ? 6 : 3) + dim * (M - 2) * phi + (stop ? dim * (phi - 1) : 0);
    int nin = 0;
    if (cfg->world_use_octomap) nin += 2 * dim * (M * (n + 1) - phi);
    nin += n_rows; /* upper bound, zero-normal and initial-state rows are dropped below */
    nin += dim * (2 * (M * n - 2) + 2 * (M * (n - 1) - 1));
    if (cfg->communication_range > 0) nin += dim * (M * (M + 1) + 2 * M);

    oracle_model *md = (oracle_model *)calloc(1, sizeof(oracle_model));
    md->nv = nv;
    md->n_ctrl = offset_slack_col;
    md->n_slack = n_slack;
    md->Q = (double *)calloc((size_t)nv * nv, sizeof(double));
    md->c = (double *)calloc(nv, sizeof(double));
    md->Aeq = (double *)calloc((size_t)neq * nv, sizeof(double));
    md->beq = (double *)calloc(neq, sizeof(double));
    md->Ain = (double *)calloc((size_t)(nin > 0 ? nin : 1) * nv, sizeof(double));
    md->lo = (double *)calloc(nin > 0 ? nin : 1, sizeof(double));
    md->hi = (double *)calloc(nin > 0 ? nin : 1, sizeof(double));
    md->lb = (double *)calloc(nv, sizeof(double));
    md->ub = (double *)calloc(nv, sizeof(double));

    /* variables and bounds, :237-283 */
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                int row = k * offset_dim + m * offset_seg + i;
                double lower = cfg->world_min[k], upper = cfg->world_max[k];
                if (k == 2 && m == 0 && cfg->planner_mode == DLSCQP_PLANNER_RECIPROCALRSFC) {
                    lower = -100;
                    upper = 100;
                }
                if (m == 0 && i < 3) {
                    md->lb[row] = -ORACLE_INF;
                    md->ub[row] = ORACLE_INF;
                } else {
                    md->lb[row] = lower;
                    md->ub[row] = upper;
                }
            }
    for (int s = 0; s < n_slack; s++) {
        md->lb[offset_slack_col + s] = -ORACLE_INF;
        md->ub[offset_slack_col + s] = 0;
    }

    /* cost 1: jerk, :286-299 */
    double Qb[36];
    oracle_build_q_base(n, phi, cfg->phi_n, dt, Qb);
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n + 1; i++) {
                int row = k * offset_dim + m * offset_seg + i;
                for (int j = 0; j < n + 1; j++) {
                    int col = k * offset_dim + m * offset_seg + j;
                    if (Qb[i * (n + 1) + j] != 0 && cfg->control_input_weight != 0)
                        md->Q[(size_t)row * nv + col] += cfg->control_input_weight * Qb[i * (n + 1) + j];
                }
            }
    /* cost 2: error to goal, :301-315 */
    int terminal_segments = oracle_terminal_segments(cfg, ag);
    md->terminal_segments = terminal_segments;
    for (int m = M - terminal_segments; m < M; m++)
        for (int k = 0; k < dim; k++) {
            int idx = k * offset_dim + m * offset_seg + n;
            double g = (double)ag->current_goal_point[k];
            md->Q[(size_t)idx * nv + idx] += cfg->terminal_weight;
            md->c[idx] += -2.0 * cfg->terminal_weight * g;
            md->c0 += cfg->terminal_weight * g * g;
        }
    /* cost 3: slack, :317-330 */
    for (int s = 0; s < ag->n_slack_obs; s++)
        for (int m = 0; m < M; m++) {
            double coef = (double)(M - m) / M;
            int idx = offset_slack_col + M * s + m;
            md->Q[(size_t)idx * nv + idx] += cfg->slack_collision_weight * coef;
        }

    /* equalities, :333-381 */
    int e = 0;
#define AEQ(r, c) md->Aeq[(size_t)(r) * nv + (c)]
    for (int k = 0; k < dim; k++) {
        int b0 = k * offset_dim; /* segment 0 */
        int b1 = k * offset_dim + offset_seg; /* segment 1 */
        AEQ(e, b0 + 0) = 1; md->beq[e] = (double)ag->position[k]; e++;           /* front position */
        if (M > 1) { AEQ(e, b0 + n) = 1; AEQ(e, b1 + 0) = -1; md->beq[e] = 0; e++; } /* back position */
        AEQ(e, b0 + 1) = pow(dt, -1) * n; AEQ(e, b0 + 0) = -pow(dt, -1) * n;
        md->beq[e] = (double)ag->velocity[k]; e++;                               /* front velocity */
        AEQ(e, b0 + 2) = pow(dt, -2) * n * (n - 1); AEQ(e, b0 + 1) = -2 * pow(dt, -2) * n * (n - 1);
        AEQ(e, b0 + 0) = pow(dt, -2) * n * (n - 1);
        md->beq[e] = (double)ag->acceleration[k]; e++;                           /* front acceleration */
        if (M > 1) {
            /* back velocity */
            AEQ(e, b1 + 1) += 1; AEQ(e, b1 + 0) += -1; AEQ(e, b0 + n) += -1; AEQ(e, b0 + n - 1) += 1;
            md->beq[e] = 0; e++;
            /* back acceleration */
            AEQ(e, b1 + 2) += 1; AEQ(e, b1 + 1) += -2; AEQ(e, b1 + 0) += 1;
            AEQ(e, b0 + n) += -1; AEQ(e, b0 + n - 1) += 2; AEQ(e, b0 + n - 2) += -1;
            md->beq[e] = 0; e++;
        }
    }
    {
        int nb = (M - 2) * phi;
        double *Ab = (double *)calloc((size_t)(nb > 0 ? nb : 1) * offset_dim, sizeof(double));
        oracle_build_aeq_base(M, n, phi, dt, Ab);
        for (int k = 0; k < dim; k++)
            for (int i = 0; i < nb; i++) {
                for (int j = 0; j < offset_dim; j++)
                    if (Ab[(size_t)i * offset_dim + j] != 0) AEQ(e, k * offset_dim + j) = Ab[(size_t)i * offset_dim + j];
                md->beq[e] = 0;
                e++;
            }
        free(Ab);
    }
    /* stop at the end of the horizon, :515-524 */
    if (stop)
        for (int k = 0; k < dim; k++) {
            int m = M - 1;
            for (int i = 1; i < phi; i++) {
                AEQ(e, k * offset_dim + m * offset_seg + n) = 1;
                AEQ(e, k * offset_dim + m * offset_seg + n - i) = -1;
                md->beq[e] = 0;
                e++;
            }
        }
#undef AEQ
    md->neq = e;

    /* inequalities */
    int r = 0;
#define AIN(rr, c) md->Ain[(size_t)(rr) * nv + (c)]
    /* SFC, :385-410 with Box::convertToLSCs, collision_constraints.cpp:66-87 */
    if (cfg->world_use_octomap && sfc) {
        for (int m = 0; m < M; m++) {
            const float *bmin = sfc + (size_t)m * 6, *bmax = bmin + 3;
            for (int f = 0; f < 2 * dim; f++) {
                int axis = f / 2;
                double nrm[3] = {0, 0, 0};
                double d;
                if ((f & 1) == 0) { nrm[axis] = 1; d = (double)bmin[axis]; }
                else { nrm[axis] = -1; d = -(double)bmax[axis]; }
                for (int j = 0; j < n + 1; j++) {
                    if (m == 0 && j < phi) continue;
                    for (int k = 0; k < dim; k++) AIN(r, k * offset_dim + m * offset_seg + j) = nrm[k];
                    md->lo[r] = d; /* obs_control_point = 0 */
                    md->hi[r] = ORACLE_INF;
                    r++;
                }
            }
        }
    }
    /* LSC, :413-450 */
    for (int ri = 0; ri < n_rows; ri++) {
        const dlscqp_lsc_row *L = &rows[ri];
        int m = L->m, i = L->i;
        if (m == 0 && i < phi) continue;
        if (f32_norm(L->normal) < SP_EPSILON_FLOAT) continue;
        double rhs = L->d;
        for (int k = 0; k < dim; k++) {
            AIN(r, k * offset_dim + m * offset_seg + i) = (double)L->normal[k];
            rhs += (double)L->normal[k] * (double)L->point[k];
        }
        if (L->slack >= 0) AIN(r, offset_slack_col + M * L->slack + m) = -1.0;
        md->lo[r] = rhs;
        md->hi[r] = ORACLE_INF;
        r++;
    }
    /* dynamic feasibility, :453-487 */
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++) {
            for (int i = 0; i < n; i++) {
                if (m == 0 && (i == 0 || i == 1)) continue;
                for (int sgn = 1; sgn >= -1; sgn -= 2) {
                    AIN(r, k * offset_dim + m * offset_seg + i + 1) = sgn * pow(dt, -1) * n;
                    AIN(r, k * offset_dim + m * offset_seg + i) = -sgn * pow(dt, -1) * n;
                    md->lo[r] = -ORACLE_INF;
                    md->hi[r] = ag->max_vel;
                    r++;
                }
            }
            for (int i = 0; i < n - 1; i++) {
                if (m == 0 && i == 0) continue;
                for (int sgn = 1; sgn >= -1; sgn -= 2) {
                    double cf = sgn * pow(dt, -2) * n * (n - 1);
                    AIN(r, k * offset_dim + m * offset_seg + i + 2) = cf;
                    AIN(r, k * offset_dim + m * offset_seg + i + 1) = -2 * cf;
                    AIN(r, k * offset_dim + m * offset_seg + i) = cf;
                    md->lo[r] = -ORACLE_INF;
                    md->hi[r] = ag->max_acc;
                    r++;
                }
            }
        }
    /* communication range, :490-513 */
    if (cfg->communication_range > 0) {
        for (int k = 0; k < dim; k++)
            for (int mi = 0; mi < M; mi++)
                for (int m = mi; m < M; m++)
                    for (int sgn = 1; sgn >= -1; sgn -= 2) {
                        AIN(r, k * offset_dim + m * offset_seg + n) += sgn;
                        AIN(r, k * offset_dim + mi * offset_seg + 0) += -sgn;
                        md->lo[r] = -ORACLE_INF;
                        md->hi[r] = 0.5 * cfg->communication_range - ag->radius;
                        r++;
                    }
        for (int k = 0; k < dim; k++)
            for (int m = 0; m < M; m++)
                for (int sgn = 1; sgn >= -1; sgn -= 2) {
                    AIN(r, k * offset_dim + m * offset_seg + n) = sgn;
                    md->lo[r] = -ORACLE_INF;
                    md->hi[r] = 0.5 * cfg->communication_range - SP_EPSILON_FLOAT + sgn * (double)ag->next_waypoint[k];
                    r++;
                }
    }
#undef AIN
    md->nin = r;
    return md;
}

/* ------------------------------------------------------------------------------------------ */
/* Generic dense strictly-convex QP:  min x'Qx + c'x  s.t. Aeq x = beq, lo <= Ain x <= hi,
 * lb <= x <= ub.   Householder null-space elimination + Goldfarb-Idnani (Math. Prog. 27, 1983)
 * dual active set with Givens updates. */

typedef struct {
    int r;      /* reduced dimension */
    int nc;     /* number of one-sided reduced rows */
    double *C;  /* nc*r  rows c_i, constraint c_i.y >= b_i (x-space normal has unit 2-norm) */
    double *b;  /* nc */
    double *sc; /* nc: norm of the original row (violation in model scale = viol * sc) */
} reduced_rows;

static int cholesky_lower(double *A, int n) { /* in place, lower triangle */
    for (int j = 0; j < n; j++) {
        double s = A[j * n + j];
        for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0)) return -1;
        double d = sqrt(s);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[i * n + j];
            for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / d;
        }
    }
    return 0;
}

int oracle_qp_solve(const oracle_model *m, double *x, double *obj, int *iters_out) {
    const int nv = m->nv, neq = m->neq;
    int status = DLSCQP_OK;
    if (iters_out) *iters_out = 0;
    /* --- Householder QR of Aeq^T (nv x neq) ------------------------------------------- */
    double *At = (double *)calloc((size_t)nv * neq, sizeof(double)); /* column k at At[k*nv..] */
    double *V = (double *)calloc((size_t)nv * neq, sizeof(double));
    for (int e = 0; e < neq; e++)
        for (int j = 0; j < nv; j++) At[(size_t)e * nv + j] = m->Aeq[(size_t)e * nv + j];
    double rmax = 0;
    for (int k = 0; k < neq; k++) {
        double *col = At + (size_t)k * nv;
        double nrm = 0;
        for (int j = k; j < nv; j++) nrm += col[j] * col[j];
        nrm = sqrt(nrm);
        if (nrm > rmax) rmax = nrm;
        if (nrm < 1e-12 * (rmax > 1 ? rmax : 1)) { status = DLSCQP_NUMERICAL; goto done_qr; }
        double alpha = col[k] > 0 ? -nrm : nrm;
        double *v = V + (size_t)k * nv;
        for (int j = k; j < nv; j++) v[j] = col[j];
        v[k] -= alpha;
        double vn = 0;
        for (int j = k; j < nv; j++) vn += v[j] * v[j];
        vn = sqrt(vn);
        for (int j = k; j < nv; j++) v[j] /= vn;
        for (int c = k; c < neq; c++) {
            double *cc = At + (size_t)c * nv;
            double dot = 0;
            for (int j = k; j < nv; j++) dot += v[j] * cc[j];
            for (int j = k; j < nv; j++) cc[j] -= 2 * dot * v[j];
        }
    }
done_qr:;
    double *Qf = NULL, *Z = NULL, *xp = NULL, *Hr = NULL, *gr = NULL, *L = NULL, *J = NULL;
    double *Rm = NULL, *y = NULL, *d = NULL, *z = NULL, *rv = NULL, *u = NULL, *s = NULL;
    int *A = NULL;
    char *flag = NULL;
    reduced_rows rr = {0, 0, NULL, NULL, NULL};
    if (status != DLSCQP_OK) goto cleanup;

    const int r = nv - neq;
    /* Qfull = H_0 H_1 ... H_{neq-1} */
    Qf = (double *)calloc((size_t)nv * nv, sizeof(double)); /* row-major */
    for (int i = 0; i < nv; i++) Qf[(size_t)i * nv + i] = 1;
    for (int k = neq - 1; k >= 0; k--) {
        double *v = V + (size_t)k * nv;
        for (int c = 0; c < nv; c++) {
            double dot = 0;
            for (int j = k; j < nv; j++) dot += v[j] * Qf[(size_t)j * nv + c];
            if (dot != 0)
                for (int j = k; j < nv; j++) Qf[(size_t)j * nv + c] -= 2 * dot * v[j];
        }
    }
    /* xp = Q1 y1, R^T y1 = beq (R = upper part of At) */
    xp = (double *)calloc(nv, sizeof(double));
    {
        double *y1 = (double *)calloc(neq > 0 ? neq : 1, sizeof(double));
        for (int i = 0; i < neq; i++) {
            double t = m->beq[i];
            for (int k = 0; k < i; k++) t -= At[(size_t)i * nv + k] * y1[k]; /* R[k][i] = At col i row k */
            y1[i] = t / At[(size_t)i * nv + i];
        }
        for (int j = 0; j < nv; j++) {
            double t = 0;
            for (int k = 0; k < neq; k++) t += Qf[(size_t)j * nv + k] * y1[k];
            xp[j] = t;
        }
        free(y1);
    }
    /* Z = Qf[:, neq:]  (nv x r) */
    Z = (double *)calloc((size_t)nv * r, sizeof(double));
    for (int j = 0; j < nv; j++)
        for (int k = 0; k < r; k++) Z[(size_t)j * r + k] = Qf[(size_t)j * nv + neq + k];

    /* Hr = Z'(2Q)Z, gr = Z'(2Q xp + c) */
    Hr = (double *)calloc((size_t)r * r, sizeof(double));
    gr = (double *)calloc(r, sizeof(double));
    {
        double *QZ = (double *)calloc((size_t)nv * r, sizeof(double));
        double *g = (double *)calloc(nv, sizeof(double));
        for (int i = 0; i < nv; i++) {
            double t = m->c[i];
            for (int j = 0; j < nv; j++) {
                double qij = m->Q[(size_t)i * nv + j] + m->Q[(size_t)j * nv + i]; /* (Q+Q') = 2Q sym */
                if (qij != 0) {
                    t += qij * xp[j];
                    for (int k = 0; k < r; k++) QZ[(size_t)i * r + k] += qij * Z[(size_t)j * r + k];
                }
            }
            g[i] = t;
        }
        for (int a = 0; a < r; a++) {
            double t = 0;
            for (int i = 0; i < nv; i++) t += Z[(size_t)i * r + a] * g[i];
            gr[a] = t;
            for (int b = 0; b < r; b++) {
                double h = 0;
                for (int i = 0; i < nv; i++) h += Z[(size_t)i * r + a] * QZ[(size_t)i * r + b];
                Hr[(size_t)a * r + b] = h;
            }
        }
        free(QZ);
        free(g);
    }
    L = (double *)malloc(sizeof(double) * r * r);
    for (int a = 0; a < r; a++)
        for (int b = 0; b < r; b++) L[a * r + b] = 0.5 * (Hr[a * r + b] + Hr[b * r + a]);
    if (cholesky_lower(L, r) != 0) { status = DLSCQP_NUMERICAL; goto cleanup; }

    /* --- one-sided reduced rows ----------------------------------------------------------- */
    {
        int cap = 2 * m->nin + 2 * nv;
        rr.r = r;
        rr.C = (double *)calloc((size_t)cap * r, sizeof(double));
        rr.b = (double *)calloc(cap, sizeof(double));
        rr.sc = (double *)calloc(cap, sizeof(double));
        int nc = 0;
        for (int i = 0; i < m->nin; i++) {
            const double *a = m->Ain + (size_t)i * nv;
            double nrm = 0, axp = 0;
            for (int j = 0; j < nv; j++) { nrm += a[j] * a[j]; axp += a[j] * xp[j]; }
            nrm = sqrt(nrm);
            if (nrm == 0) continue;
            for (int side = 0; side < 2; side++) {
                double bound = side == 0 ? m->lo[i] : m->hi[i];
                if (fabs(bound) >= ORACLE_INF) continue;
                double sg = side == 0 ? 1.0 : -1.0; /* a.x >= lo   |  -a.x >= -hi */
                for (int k = 0; k < r; k++) {
                    double t = 0;
                    for (int j = 0; j < nv; j++)
                        if (a[j] != 0) t += a[j] * Z[(size_t)j * r + k];
                    rr.C[(size_t)nc * r + k] = sg * t / nrm;
                }
                rr.b[nc] = sg * (bound - axp) / nrm;
                rr.sc[nc] = nrm;
                nc++;
            }
        }
        for (int j = 0; j < nv; j++)
            for (int side = 0; side < 2; side++) {
                double bound = side == 0 ? m->lb[j] : m->ub[j];
                if (fabs(bound) >= ORACLE_INF) continue;
                double sg = side == 0 ? 1.0 : -1.0;
                for (int k = 0; k < r; k++) rr.C[(size_t)nc * r + k] = sg * Z[(size_t)j * r + k];
                rr.b[nc] = sg * (bound - xp[j]);
                rr.sc[nc] = 1.0;
                nc++;
            }
        rr.nc = nc;
    }

    /* --- Goldfarb-Idnani ------------------------------------------------------------------- */
    y = (double *)calloc(r, sizeof(double));
    J = (double *)calloc((size_t)r * r, sizeof(double));
    Rm = (double *)calloc((size_t)r * r, sizeof(double));
    d = (double *)calloc(r, sizeof(double));
    z = (double *)calloc(r, sizeof(double));
    rv = (double *)calloc(r + 1, sizeof(double));
    u = (double *)calloc(r + 1, sizeof(double));
    A = (int *)calloc(r + 1, sizeof(int));
    s = (double *)calloc(rr.nc > 0 ? rr.nc : 1, sizeof(double));
    flag = (char *)calloc(rr.nc > 0 ? rr.nc : 1, 1); /* 1 active, 2 ignored (within CPLEX tol) */

    /* y0 = -Hr^{-1} gr */
    for (int i = 0; i < r; i++) {
        double t = -gr[i];
        for (int k = 0; k < i; k++) t -= L[i * r + k] * y[k];
        y[i] = t / L[i * r + i];
    }
    for (int i = r - 1; i >= 0; i--) {
        double t = y[i];
        for (int k = i + 1; k < r; k++) t -= L[k * r + i] * y[k];
        y[i] = t / L[i * r + i];
    }
    /* J = L^{-T}: column c solves L^T J[:,c] = e_c */
    for (int c = 0; c < r; c++)
        for (int i = r - 1; i >= 0; i--) {
            double t = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k < r; k++) t -= L[k * r + i] * J[(size_t)k * r + c];
            J[(size_t)i * r + c] = t / L[i * r + i];
        }

    int q = 0, iters = 0;
    const int max_iters = 50 * (r + 10) + rr.nc;
    for (;;) {
        /* most violated inactive row */
        int ip = -1;
        double worst = -ORACLE_VIOL_TOL;
        for (int i = 0; i < rr.nc; i++) {
            if (flag[i]) continue;
            double t = -rr.b[i];
            const double *ci = rr.C + (size_t)i * r;
            for (int k = 0; k < r; k++) t += ci[k] * y[k];
            s[i] = t;
            if (t < worst) { worst = t; ip = i; }
        }
        if (ip < 0) break;
        const double *np = rr.C + (size_t)ip * r;
        double sp = s[ip];
        double uq = 0;
        int added = 0;
        while (!added) {
            if (++iters > max_iters) { status = DLSCQP_ITERATION_LIMIT; goto finish; }
            /* d = J' np */
            for (int j = 0; j < r; j++) {
                double t = 0;
                for (int i = 0; i < r; i++) t += J[(size_t)i * r + j] * np[i];
                d[j] = t;
            }
            /* z = J2 d2 */
            double dd2 = 0;
            for (int j = q; j < r; j++) dd2 += d[j] * d[j];
            for (int i = 0; i < r; i++) {
                double t = 0;
                for (int j = q; j < r; j++) t += J[(size_t)i * r + j] * d[j];
                z[i] = t;
            }
            /* rv = R^{-1} d1 */
            for (int i = q - 1; i >= 0; i--) {
                double t = d[i];
                for (int k = i + 1; k < q; k++) t -= Rm[(size_t)i * r + k] * rv[k];
                rv[i] = t / Rm[(size_t)i * r + i];
            }
            double t1 = ORACLE_INF, t2 = ORACLE_INF;
            int l = -1;
            for (int k = 0; k < q; k++)
                if (rv[k] > 0 && u[k] / rv[k] < t1) { t1 = u[k] / rv[k]; l = k; }
            double znp = 0;
            for (int i = 0; i < r; i++) znp += z[i] * np[i];
            /* linear dependence on the active set: ||d2||^2 relative to ||d||^2 */
            double dall = 0;
            for (int j = 0; j < r; j++) dall += d[j] * d[j];
            int zzero = !(dd2 > 1e-20 * dall) || !(znp > 0);
            if (!zzero) t2 = -sp / znp;
            if (t1 >= ORACLE_INF && t2 >= ORACLE_INF) {
                /* infeasible -- unless the row is satisfied within CPLEX's feasibility tolerance */
                if (-sp * rr.sc[ip] <= CPLEX_FEAS_TOL) { flag[ip] = 2; break; }
                status = DLSCQP_INFEASIBLE;
                goto finish;
            }
            double t = t1 < t2 ? t1 : t2;
            if (t2 < ORACLE_INF) {
                for (int i = 0; i < r; i++) y[i] += t * z[i];
                sp += t * znp;
            }
            for (int k = 0; k < q; k++) u[k] -= t * rv[k];
            uq += t;
            if (t2 < ORACLE_INF && t2 <= t1) {
                /* add constraint ip: Givens from the bottom up to zero d[q+1..r-1] */
                for (int j = r - 1; j > q; j--) {
                    double a = d[j - 1], b = d[j];
                    if (b == 0) continue;
                    double rho = hypot(a, b), cs = a / rho, sn = b / rho;
                    d[j - 1] = rho;
                    d[j] = 0;
                    for (int i = 0; i < r; i++) {
                        double j0 = J[(size_t)i * r + j - 1], j1 = J[(size_t)i * r + j];
                        J[(size_t)i * r + j - 1] = cs * j0 + sn * j1;
                        J[(size_t)i * r + j] = -sn * j0 + cs * j1;
                    }
                }
                for (int i = 0; i <= q; i++) Rm[(size_t)i * r + q] = d[i];
                u[q] = uq;
                A[q] = ip;
                flag[ip] = 1;
                q++;
                added = 1;
            } else {
                /* drop active constraint l */
                flag[A[l]] = 0;
                for (int c = l; c < q - 1; c++) {
                    for (int i = 0; i <= c + 1; i++) Rm[(size_t)i * r + c] = Rm[(size_t)i * r + c + 1];
                    u[c] = u[c + 1];
                    A[c] = A[c + 1];
                }
                for (int i = 0; i < q; i++) Rm[(size_t)i * r + q - 1] = 0;
                q--;
                for (int tt = l; tt < q; tt++) {
                    double a = Rm[(size_t)tt * r + tt], b = Rm[(size_t)(tt + 1) * r + tt];
                    if (b == 0) continue;
                    double rho = hypot(a, b), cs = a / rho, sn = b / rho;
                    for (int c = tt; c < q; c++) {
                        double r0 = Rm[(size_t)tt * r + c], r1 = Rm[(size_t)(tt + 1) * r + c];
                        Rm[(size_t)tt * r + c] = cs * r0 + sn * r1;
                        Rm[(size_t)(tt + 1) * r + c] = -sn * r0 + cs * r1;
                    }
                    for (int i = 0; i < r; i++) {
                        double j0 = J[(size_t)i * r + tt], j1 = J[(size_t)i * r + tt + 1];
                        J[(size_t)i * r + tt] = cs * j0 + sn * j1;
                        J[(size_t)i * r + tt + 1] = -sn * j0 + cs * j1;
                    }
                }
            }
        }
    }
finish:
    if (iters_out) *iters_out = iters;
    if (status == DLSCQP_OK) {
        for (int j = 0; j < nv; j++) {
            double t = xp[j];
            for (int k = 0; k < r; k++) t += Z[(size_t)j * r + k] * y[k];
            x[j] = t;
        }
        if (obj) {
            double o = m->c0;
            for (int i = 0; i < nv; i++) {
                o += m->c[i] * x[i];
                for (int j = 0; j < nv; j++) {
                    double qij = m->Q[(size_t)i * nv + j];
                    if (qij != 0) o += qij * x[i] * x[j];
                }
            }
            *obj = o;
        }
    }
cleanup:
    free(At); free(V); free(Qf); free(Z); free(xp); free(Hr); free(gr); free(L); free(J);
    free(Rm); free(y); free(d); free(z); free(rv); free(u); free(A); free(s); free(flag);
    free(rr.C); free(rr.b); free(rr.sc);
    return status;
}

/* TrajOptimizer::solve, traj_optimizer.cpp:18-165 */
int oracle_traj_solve(const dlscqp_config *cfg, const dlscqp_agent *ag, int n_rows,
                      const dlscqp_lsc_row *rows, const float *sfc, float *ctrl, double *x_f64,
                      double *cost, double *slack_cost, int *iters) {
    oracle_model *md = oracle_model_build(cfg, ag, n_rows, rows, sfc);
    if (!md) return DLSCQP_NUMERICAL;
    double *x = (double *)calloc(md->nv, sizeof(double));
    double obj = 0;
    int st = oracle_qp_solve(md, x, &obj, iters);
    if (st == DLSCQP_OK) {
        const int dim = cfg->world_dimension, M = cfg->M, n = cfg->n;
        const int offset_seg = n + 1, offset_dim = M * (n + 1);
        /* :69-83 */
        if (ctrl)
            for (int m = 0; m < M; m++)
                for (int i = 0; i < n + 1; i++) {
                    float *p = ctrl + ((size_t)m * (n + 1) + i) * 3;
                    p[0] = (float)x[0 * offset_dim + m * offset_seg + i];
                    p[1] = (float)x[1 * offset_dim + m * offset_seg + i];
                    p[2] = dim == 3 ? (float)x[2 * offset_dim + m * offset_seg + i] : (float)cfg->world_z_2d;
                }
        /* :85-106 */
        if (slack_cost)
            for (int s = 0; s < ag->n_slack_obs; s++) {
                double sc = 0;
                for (int m = 0; m < M; m++) sc += fabs(x[md->n_ctrl + M * s + m]);
                slack_cost[s] = sc;
            }
        if (cost) *cost = obj; /* :109 */
        if (x_f64) memcpy(x_f64, x, sizeof(double) * md->nv);
    }
    free(x);
    oracle_model_free(md);
    return st;
}

/* GoalOptimizer::solve + populatebyrow, goal_optimizer.cpp:7-198.
 * The LP  min t, 0 <= t <= 1+1e-5, a_r t + b_r >= 0  is solved exactly by bound propagation
 * (what a simplex code's presolve does for a single column). */
int oracle_goal_solve(const dlscqp_config *cfg, const dlscqp_agent *ag, int n_rows,
                      const dlscqp_lsc_row *rows, const float *sfc, float goal_out[3], double *t_out) {
    const int dim = cfg->world_dimension, M = cfg->M, n = cfg->n;
    const float *cg = ag->current_goal_point, *wp = ag->next_waypoint;
    if (t_out) *t_out = 0;
    /* :12-14 */
    if (f32_dist(cg, wp) < SP_EPSILON_FLOAT) {
        for (int k = 0; k < 3; k++) goal_out[k] = wp[k];
        return DLSCQP_OK;
    }
    double t_lo = 0, t_hi = 1 + SP_EPSILON_FLOAT; /* :147 */
    int infeasible = 0;
    float diff[3];
    for (int k = 0; k < 3; k++) diff[k] = cg[k] - wp[k]; /* float32, as point3d(k) - point3d(k) */
#define LP_ROW(a_, b_)                                                          \
    do {                                                                        \
        double lp_a = (a_), lp_b = (b_);                                        \
        if (lp_a > 0) { if (-lp_b / lp_a > t_lo) t_lo = -lp_b / lp_a; }         \
        else if (lp_a < 0) { if (-lp_b / lp_a < t_hi) t_hi = -lp_b / lp_a; }    \
        else if (lp_b < -CPLEX_FEAS_TOL) infeasible = 1;                        \
    } while (0)
    /* SFC rows, :158-173 */
    if (cfg->world_use_octomap && sfc) {
        const float *bmin = sfc + (size_t)(M - 1) * 6, *bmax = bmin + 3;
        for (int f = 0; f < 2 * dim; f++) {
            int axis = f / 2;
            double nrm = (f & 1) == 0 ? 1.0 : -1.0;
            double dd = (f & 1) == 0 ? (double)bmin[axis] : -(double)bmax[axis];
            float off = wp[axis] - 0.0f;
            LP_ROW(nrm * (double)diff[axis], nrm * (double)off - dd);
        }
    }
    /* LSC rows, :176-195 */
    for (int ri = 0; ri < n_rows; ri++) {
        const dlscqp_lsc_row *L = &rows[ri];
        if (L->flags & DLSCQP_ROW_DYNAMIC) continue;
        if (L->m != M - 1 || L->i != n) continue;
        if (f32_norm(L->normal) < SP_EPSILON_FLOAT) continue;
        double a = 0, b = -L->d;
        for (int k = 0; k < dim; k++) {
            float off = wp[k] - L->point[k];
            a += (double)L->normal[k] * (double)diff[k];
            b += (double)L->normal[k] * (double)off;
        }
        LP_ROW(a, b);
    }
#undef LP_ROW
    if (t_lo > t_hi + CPLEX_FEAS_TOL) infeasible = 1;
    if (!infeasible) {
        double t = t_lo;
        if (t_out) *t_out = t;
        float tf = (float)t; /* Vector3::operator*(float) */
        for (int k = 0; k < 3; k++) goal_out[k] = diff[k] * tf + wp[k]; /* :51 */
        return DLSCQP_OK;
    }
    /* exception branch, :54-81 */
    int numerical_error = 1;
    if (cfg->world_use_octomap && sfc) {
        const float *bmin = sfc + (size_t)(M - 1) * 6, *bmax = bmin + 3;
        /* Box::isPointInBox, collision_constraints.cpp:109-116: all three axes, float compare
         * against double-promoted bounds */
        for (int k = 0; k < 3; k++)
            if (!((double)cg[k] > (double)bmin[k] - SP_EPSILON_FLOAT && (double)cg[k] < (double)bmax[k] + SP_EPSILON_FLOAT))
                numerical_error = 0;
    }
    for (int ri = 0; ri < n_rows; ri++) {
        const dlscqp_lsc_row *L = &rows[ri];
        if (L->flags & DLSCQP_ROW_DYNAMIC) continue;
        if (L->m != M - 1 || L->i != n) continue;
        if (f32_norm(L->normal) < SP_EPSILON_FLOAT) continue;
        /* normal.dot(cg - point): float32 dot product of float32 differences (all 3 axes) */
        float e0 = cg[0] - L->point[0], e1 = cg[1] - L->point[1], e2 = cg[2] - L->point[2];
        float dot = L->normal[0] * e0 + L->normal[1] * e1 + L->normal[2] * e2;
        double delta = (double)dot - L->d;
        if (delta < -SP_EPSILON_FLOAT) numerical_error = 0;
    }
    if (numerical_error) {
        for (int k = 0; k < 3; k++) goal_out[k] = cg[k];
        if (t_out) *t_out = 1.0;
        return DLSCQP_GOAL_FALLBACK;
    }
    return DLSCQP_INFEASIBLE;
}

int oracle_plan_batch(const dlscqp_config *cfg, int n_agents, const dlscqp_agent *agents,
                      const int64_t *row_offsets, const dlscqp_lsc_row *rows, const float *sfc,
                      int max_slack_obs, float *out_goal, int *out_goal_status, float *out_ctrl,
                      double *out_x, double *out_cost, double *out_slack_cost, int *out_status,
                      int do_goal) {
    const int M = cfg->M, n = cfg->n, dim = cfg->world_dimension;
    const size_t ctrl_stride = (size_t)M * (n + 1) * 3;
    const size_t x_stride = (size_t)dim * M * (n + 1) + (size_t)M * max_slack_obs;
    for (int a = 0; a < n_agents; a++) {
        dlscqp_agent ag = agents[a];
        const dlscqp_lsc_row *r0 = rows + row_offsets[a];
        int nr = (int)(row_offsets[a + 1] - row_offsets[a]);
        const float *box = sfc ? sfc + (size_t)a * M * 6 : NULL;
        if (do_goal) {
            float g[3];
            int gs = oracle_goal_solve(cfg, &ag, nr, r0, box, g, NULL);
            if (out_goal_status) out_goal_status[a] = gs;
            if (gs == DLSCQP_INFEASIBLE) {
                if (out_status) out_status[a] = DLSCQP_INFEASIBLE;
                continue; /* the reference would have thrown before trajOptimization */
            }
            for (int k = 0; k < 3; k++) {
                ag.current_goal_point[k] = g[k]; /* traj_planner.cpp:494 */
                if (out_goal) out_goal[(size_t)a * 3 + k] = g[k];
            }
        }
        int st = oracle_traj_solve(cfg, &ag, nr, r0, box, out_ctrl ? out_ctrl + a * ctrl_stride : NULL,
                                   out_x ? out_x + a * x_stride : NULL, out_cost ? out_cost + a : NULL,
                                   out_slack_cost ? out_slack_cost + (size_t)a * max_slack_obs : NULL, NULL);
        if (out_status) out_status[a] = st;
    }
    return 0;
}

/* Trajectory<point3d>::getPointAt / derivative / getStateAt, trajectory.cpp:110-199, with
 * octomath::Vector3 float32 arithmetic (operator*(float), operator+, operator-). */
static void traj_point_at(const float *cp, int M, int n, double dt, double time, float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    if (time < 0) return;
    int m = -1;
    double t_normalized = 0, segment_end_time = 0;
    for (int idx = 0; idx < M; idx++) {
        segment_end_time += dt;
        if (time < segment_end_time) {
            m = idx;
            t_normalized = 1 - (segment_end_time - time) / dt;
            break;
        }
    }
    if (m == -1) {
        if (time < segment_end_time + SP_EPSILON_FLOAT) { m = M - 1; t_normalized = 1.0; }
        else return;
    }
    for (int i = 0; i < n + 1; i++) {
        double b = n_choose_k(n, i) * pow(t_normalized, i) * pow(1 - t_normalized, n - i);
        float bf = (float)b;
        const float *c = cp + ((size_t)m * (n + 1) + i) * 3;
        for (int k = 0; k < 3; k++) out[k] = out[k] + c[k] * bf;
    }
}

void oracle_state_at(const float *ctrl, int M, int n, double dt, double time, float state_out[9]) {
    /* derivative trajectories keep n+1 slots per segment with degree n-1, n-2 */
    size_t sz = (size_t)M * (n + 1) * 3;
    float *d1 = (float *)calloc(sz, sizeof(float));
    float *d2 = (float *)calloc(sz, sizeof(float));
    traj_point_at(ctrl, M, n, dt, time, state_out);
    float s1 = (float)((double)n / dt);
    for (int m = 0; m < M; m++)
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++)
                d1[((size_t)m * (n + 1) + i) * 3 + k] =
                    (ctrl[((size_t)m * (n + 1) + i + 1) * 3 + k] - ctrl[((size_t)m * (n + 1) + i) * 3 + k]) * s1;
    /* the derivative trajectory has degree n-1 but the same (n+1)-slot stride */
    {
        /* evaluate with degree n-1: repack into stride n */
        float *p1 = (float *)calloc((size_t)M * n * 3, sizeof(float));
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) p1[((size_t)m * n + i) * 3 + k] = d1[((size_t)m * (n + 1) + i) * 3 + k];
        traj_point_at(p1, M, n - 1, dt, time, state_out + 3);
        float s2 = (float)((double)(n - 1) / dt);
        float *p2 = (float *)calloc((size_t)M * (n - 1) * 3, sizeof(float));
        for (int m = 0; m < M; m++)
            for (int i = 0; i < n - 1; i++)
                for (int k = 0; k < 3; k++)
                    p2[((size_t)m * (n - 1) + i) * 3 + k] =
                        (p1[((size_t)m * n + i + 1) * 3 + k] - p1[((size_t)m * n + i) * 3 + k]) * s2;
        traj_point_at(p2, M, n - 2, dt, time, state_out + 6);
        free(p1);
        free(p2);
    }
    free(d1);
    free(d2);
}
