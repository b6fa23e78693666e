/*
 * dlsc_lsc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for SURVEY.md row N1:
 * DLSC-GC half-space generation between agents.  See the declarations in dlsc_oracle.h for the
 * reference lines each function restates.  octomath::Vector3 is three float32; its operators work
 * in float32, dot()/norm()/distance() return double as noted at each use.
 * Compile with -ffp-contract=off.
 */
#include "dlsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SP_EPSILON_FLOAT 1e-5

typedef struct { float x, y, z; } v3;

static v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 v3_mulf(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
/* Vector3::dot: float32 products and sums, returned as double */
static double v3_dot(v3 a, v3 b) { float s = a.x * b.x + a.y * b.y + a.z * b.z; return (double)s; }
static double v3_norm(v3 a) { float s = a.x * a.x + a.y * a.y + a.z * a.z; return sqrt((double)s); }
/* Vector3::distance: float32 differences, squared and summed in double */
static double v3_dist(v3 a, v3 b) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrt(dx * dx + dy * dy + dz * dz);
}
static v3 v3_cross(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
/* Vector3::normalized: divide by (float) norm when it is positive */
static v3 v3_normalized(v3 a) {
    double len = v3_norm(a);
    if (len > 0) { float f = (float)len; a.x /= f; a.y /= f; a.z /= f; }
    return a;
}
static int v3_eq(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

/* ---- point-to-hull: exact enumeration (what GJK converges to) --------------------------------- */
static oracle_gjk_fn g_gjk = 0;
void oracle_set_gjk(oracle_gjk_fn fn) { g_gjk = fn; }

static void consider(const double c[3], double best[3], double *bd) {
    double d = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    if (d < *bd) { *bd = d; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
}

static double det3(const double a[3], const double b[3], const double c[3]) {
    return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
}

double oracle_hull_closest_point(const double *p, int n, double v[3]) {
    double bd = 1e300;
    /* origin inside the hull <=> inside one of the tetrahedra spanned by its points (Caratheodory): distance 0 */
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            for (int c = b + 1; c < n; c++)
                for (int d = c + 1; d < n; d++) {
                    const double *A = p + 3 * a, *B = p + 3 * b, *Cc = p + 3 * c, *D = p + 3 * d;
                    double ab[3], ac[3], ad[3], oa[3];
                    for (int k = 0; k < 3; k++) { ab[k] = B[k] - A[k]; ac[k] = Cc[k] - A[k]; ad[k] = D[k] - A[k]; oa[k] = -A[k]; }
                    double vol = det3(ab, ac, ad);
                    if (fabs(vol) < 1e-300) continue;
                    double l1 = det3(oa, ac, ad) / vol, l2 = det3(ab, oa, ad) / vol, l3 = det3(ab, ac, oa) / vol;
                    if (l1 >= 0 && l2 >= 0 && l3 >= 0 && l1 + l2 + l3 <= 1) { v[0] = v[1] = v[2] = 0; return 0.0; }
                }
    for (int a = 0; a < n; a++) consider(p + 3 * a, v, &bd);
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++) {
            double ab[3], den = 0, num = 0;
            for (int k = 0; k < 3; k++) { ab[k] = p[3 * b + k] - p[3 * a + k]; den += ab[k] * ab[k]; num -= p[3 * a + k] * ab[k]; }
            if (den <= 0) continue;
            double t = num / den;
            if (t <= 0 || t >= 1) continue;
            double c[3];
            for (int k = 0; k < 3; k++) c[k] = p[3 * a + k] + t * ab[k];
            consider(c, v, &bd);
        }
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            for (int cidx = b + 1; cidx < n; cidx++) {
                double ab[3], ac[3], g11 = 0, g12 = 0, g22 = 0, r1 = 0, r2 = 0;
                for (int k = 0; k < 3; k++) {
                    ab[k] = p[3 * b + k] - p[3 * a + k];
                    ac[k] = p[3 * cidx + k] - p[3 * a + k];
                    g11 += ab[k] * ab[k]; g12 += ab[k] * ac[k]; g22 += ac[k] * ac[k];
                    r1 -= p[3 * a + k] * ab[k]; r2 -= p[3 * a + k] * ac[k];
                }
                double det = g11 * g22 - g12 * g12;
                if (!(det > 1e-14 * g11 * g22)) continue;
                double u = (r1 * g22 - r2 * g12) / det, w = (g11 * r2 - g12 * r1) / det;
                if (u <= 0 || w <= 0 || u + w >= 1) continue;
                double c[3];
                for (int k = 0; k < 3; k++) c[k] = p[3 * a + k] + u * ab[k] + w * ac[k];
                consider(c, v, &bd);
            }
    return sqrt(bd);
}

/* ---- trajectories ------------------------------------------------------------------------------ */
void oracle_shift_plan(const float *prev, int M, int n, float *out) { /* traj_planner.cpp:412-421, 304-314 */
    int N = n + 1;
    for (int m = 0; m < M; m++)
        for (int i = 0; i < N; i++)
            for (int k = 0; k < 3; k++)
                out[(m * N + i) * 3 + k] = m == M - 1 ? prev[((M - 1) * N + n) * 3 + k] : prev[((m + 1) * N + i) * 3 + k];
}

void oracle_const_vel_plan(const float pos[3], const float vel[3], int M, int n, double dt, float *out) {
    /* trajectory.cpp:78-91: time advances by segment_time / n after EVERY control point; T = point3d, so
       `velocity * time` is Vector3::operator*(float) */
    double time = 0;
    for (int m = 0; m < M; m++)
        for (int i = 0; i < n + 1; i++) {
            float tf = (float)time;
            for (int k = 0; k < 3; k++) out[(m * (n + 1) + i) * 3 + k] = pos[k] + vel[k] * tf;
            time += dt / n;
        }
}

/* ---- geometry.hpp ------------------------------------------------------------------------------ */
typedef struct { double dist; v3 p1, p2; } closest_t;

/* geometry.hpp:76-111 */
static closest_t point_segment(v3 point, v3 ls, v3 le) {
    v3 a = v3_sub(ls, point), b = v3_sub(le, point), rel;
    double dist, dist_min;
    if (v3_eq(a, b)) {
        dist_min = v3_norm(a);
        rel = a;
    } else {
        dist_min = v3_norm(a);
        rel = a;
        dist = v3_norm(b);
        if (dist_min > dist) { dist_min = dist; rel = b; }
        v3 n_line = v3_normalized(v3_sub(b, a));
        v3 c = v3_sub(a, v3_mulf(n_line, (float)v3_dot(a, n_line)));
        dist = v3_norm(c);
        if (v3_dot(v3_sub(c, a), v3_sub(c, b)) < 0 && dist_min > dist) { dist_min = dist; rel = c; }
    }
    closest_t r;
    r.dist = dist_min;
    r.p1 = point;
    r.p2 = v3_add(rel, point);
    return r;
}

/* geometry.hpp:139-182: Eigen::Matrix3f inverse (cofactors / determinant) times the offset */
static closest_t lines(v3 s1, v3 e1, v3 s2, v3 e2) {
    closest_t r;
    v3 n1 = v3_normalized(v3_sub(e1, s1)), n2 = v3_normalized(v3_sub(e2, s2));
    v3 mn2 = {-n2.x, -n2.y, -n2.z};
    if (v3_dist(n1, n2) < SP_EPSILON_FLOAT || v3_dist(n1, mn2) < SP_EPSILON_FLOAT) {
        v3 delta = v3_sub(s2, s1);
        delta = v3_sub(delta, v3_mulf(n1, (float)v3_dot(delta, n1)));
        r.dist = v3_norm(delta);
        r.p1 = s1;
        r.p2 = v3_add(s1, delta);
    } else {
        v3 delta = v3_sub(s2, s1);
        v3 n3 = v3_normalized(v3_cross(n2, n1));
        float A[3][3] = {{n1.x, -n2.x, n3.x}, {n1.y, -n2.y, n3.y}, {n1.z, -n2.z, n3.z}};
        float C[3][3]; /* cofactors: inverse = C^T / det */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
                C[i][j] = A[i1][j1] * A[i2][j2] - A[i1][j2] * A[i2][j1];
            }
        float det = A[0][0] * C[0][0] + A[1][0] * C[1][0] + A[2][0] * C[2][0];
        float invdet = 1.0f / det;
        float b[3] = {delta.x, delta.y, delta.z}, al[3];
        for (int i = 0; i < 3; i++) {
            /* inverse(i, j) = C[j][i] * invdet */
            al[i] = (C[0][i] * invdet) * b[0] + (C[1][i] * invdet) * b[1] + (C[2][i] * invdet) * b[2];
        }
        r.dist = fabs((double)al[2]);
        r.p1 = v3_add(s1, v3_mulf(n1, al[0]));
        r.p2 = v3_add(s2, v3_mulf(n2, al[1]));
    }
    return r;
}

/* geometry.hpp:184-274 */
static closest_t segments(v3 s1, v3 e1, v3 s2, v3 e2) {
    closest_t r;
    if (v3_dist(s1, e1) < SP_EPSILON_FLOAT) {
        r = point_segment(s1, s2, e2);
    } else if (v3_dist(s2, e2) < SP_EPSILON_FLOAT) {
        r = point_segment(s2, s1, e1);
        v3 t = r.p1; r.p1 = r.p2; r.p2 = t;
    } else {
        v3 v1 = v3_sub(e1, s1), v2 = v3_sub(e2, s2);
        double l1 = v3_norm(v1), l2 = v3_norm(v2);
        v3 n1 = v3_mulf(v1, (float)(1 / l1)), n2 = v3_mulf(v2, (float)(1 / l2));
        if (v3_norm(v3_cross(n1, n2)) < SP_EPSILON_FLOAT) {
            double bound_min = v3_dot(v3_sub(s2, s1), n1), bound_max = v3_dot(v3_sub(e2, s1), n1);
            v3 p2_min = s2, p2_max = e2;
            if (bound_max < bound_min) {
                double t = bound_min; bound_min = bound_max; bound_max = t;
                v3 tv = p2_min; p2_min = p2_max; p2_max = tv;
            }
            v3 delta = v3_sub(s2, s1);
            delta = v3_sub(delta, v3_mulf(n1, (float)v3_dot(delta, n1)));
            if (l1 < bound_min) { r.p1 = e1; r.p2 = p2_min; }
            else if (bound_max < 0) { r.p1 = s1; r.p2 = p2_max; }
            else if (bound_min < 0) { r.p1 = s1; r.p2 = v3_add(s1, delta); }
            else { r.p1 = v3_sub(p2_min, delta); r.p2 = p2_min; }
            r.dist = v3_dist(r.p1, r.p2);
        } else {
            r = lines(s1, e1, s2, e2);
            double alpha1 = v3_dot(v3_sub(r.p1, s1), n1) / l1, alpha2 = v3_dot(v3_sub(r.p2, s2), n2) / l2;
            if (alpha1 < 0) r.p1 = s1; else if (alpha1 > 1) r.p1 = e1;
            if (alpha2 < 0) r.p2 = s2; else if (alpha2 > 1) r.p2 = e2;
            if (alpha1 < 0 || alpha1 > 1) {
                double dot = v3_dot(n2, v3_sub(r.p1, s2));
                if (dot < 0) dot = 0; else if (dot > l2) dot = l2;
                r.p2 = v3_add(s2, v3_mulf(n2, (float)dot));
            }
            if (alpha2 < 0 || alpha2 > 1) {
                double dot = v3_dot(n1, v3_sub(r.p2, s1));
                if (dot < 0) dot = 0; else if (dot > l1) dot = l1;
                r.p1 = v3_add(s1, v3_mulf(n1, (float)dot));
            }
            r.dist = v3_dist(r.p1, r.p2);
        }
    }
    return r;
}

static v3 ld(const float *p) { v3 r = {p[0], p[1], p[2]}; return r; }

/* traj_planner.cpp:603-666, agents only */
int oracle_generate_dlscgc(const dlscqp_config *cfg, const float *init_self, const float goal_self[3],
                           double radius_self, double downwash_self, int n_nbr, const float *const *init_nbr,
                           const float *goal_nbr, const double *radius_nbr, const double *downwash_nbr,
                           dlscqp_lsc_row *rows_out) {
    const int M = cfg->M, n = cfg->n, N = n + 1, phi = cfg->phi;
    int nr = 0;
    for (int oi = 0; oi < n_nbr; oi++) {
        const double collision_dist = radius_nbr[oi] + radius_self;                               /* :605 */
        const double downwash = (downwash_self * radius_self + downwash_nbr[oi] * radius_nbr[oi]) /
                                (radius_self + radius_nbr[oi]);                                    /* :1150-1153 */
        const float *pred = init_nbr[oi];
        /* Trajectory::coordinateTransform: z /= (float)downwash, trajectory.cpp:206-219 */
        const float dwf = (float)downwash;
        /* TrajPlanner::coordinateTransform of a point: z = z / downwash in double, :1183-1187 */
        v3 obs_goal_t = ld(goal_nbr + 3 * oi), agent_goal_t = ld(goal_self);
        obs_goal_t.z = (float)((double)obs_goal_t.z / downwash);
        agent_goal_t.z = (float)((double)agent_goal_t.z / downwash);
        for (int m = 0; m < M; m++) {
            if (m < M - 1) {
                /* normalVectorBetweenPolys, :1102-1127 */
                v3 rel[8];
                double pts[8 * 3];
                for (int i = 0; i < N; i++) {
                    v3 a = ld(init_self + (m * N + i) * 3), b = ld(pred + (m * N + i) * 3);
                    a.z = a.z / dwf;
                    b.z = b.z / dwf;
                    rel[i] = v3_sub(a, b);
                    pts[3 * i] = rel[i].x; pts[3 * i + 1] = rel[i].y; pts[3 * i + 2] = rel[i].z;
                }
                double v[3];
                if (g_gjk) g_gjk(pts, N, v); else oracle_hull_closest_point(pts, N, v);
                v3 cp = {(float)v[0], (float)v[1], (float)v[2]}; /* point3d(0,0,0) + point3d(v) */
                v3 zero = {0, 0, 0};
                cp = v3_add(zero, cp);
                v3 nt = v3_normalized(cp);
                v3 nrm = {nt.x, nt.y, (float)((double)nt.z / downwash)}; /* :634-636 */
                for (int i = 0; i < N; i++) {
                    if (m == 0 && i < phi) continue;
                    double d = 0.5 * (collision_dist + v3_dot(rel[i], nt)); /* :639-640 */
                    dlscqp_lsc_row *r = &rows_out[nr++];
                    memset(r, 0, sizeof *r);
                    r->normal[0] = nrm.x; r->normal[1] = nrm.y; r->normal[2] = nrm.z;
                    for (int k = 0; k < 3; k++) r->point[k] = pred[(m * N + i) * 3 + k];
                    r->d = d;
                    r->m = (int16_t)m; r->i = (int8_t)i; r->obs = (int16_t)oi; r->slack = -1; r->flags = 0;
                }
            } else {
                /* last segment, :641-663 */
                v3 l1s = ld(pred + ((M - 1) * N + n) * 3), l2s = ld(init_self + ((M - 1) * N + n) * 3);
                l1s.z = l1s.z / dwf;
                l2s.z = l2s.z / dwf;
                closest_t c = segments(l1s, obs_goal_t, l2s, agent_goal_t);
                v3 nt = v3_normalized(v3_sub(c.p2, c.p1));
                double d = 0.5 * (collision_dist + c.dist);
                v3 nrm = {nt.x, nt.y, (float)((double)nt.z / downwash)};
                v3 pt = c.p1;
                pt.z = (float)((double)pt.z * downwash);
                for (int i = 0; i < N; i++) {
                    dlscqp_lsc_row *r = &rows_out[nr++];
                    memset(r, 0, sizeof *r);
                    r->normal[0] = nrm.x; r->normal[1] = nrm.y; r->normal[2] = nrm.z;
                    r->point[0] = pt.x; r->point[1] = pt.y; r->point[2] = pt.z;
                    r->d = d;
                    r->m = (int16_t)m; r->i = (int8_t)i; r->obs = (int16_t)oi; r->slack = -1; r->flags = 0;
                }
            }
        }
    }
    return nr;
}


/* ---- dynamic (non-agent) obstacles: traj_planner.cpp:616-627 ------------------------------------------------ */

/* closestPointsBetweenLinePaths, geometry.hpp:113-137 */
static closest_t line_paths(v3 s1, v3 e1, v3 s2, v3 e2) {
    v3 rs = v3_sub(s2, s1), re = v3_sub(e2, e1); /* Line operator-, geometry.hpp:32-34 */
    v3 origin = {0, 0, 0};
    closest_t rel = point_segment(origin, rs, re);
    double len = v3_dist(rs, re); /* Line::length */
    double alpha = len > 0 ? v3_norm(v3_sub(rel.p2, rs)) / len : 0.0;
    closest_t c;
    c.dist = rel.dist;
    c.p1 = v3_add(s1, v3_mulf(v3_sub(e1, s1), (float)alpha));
    c.p2 = v3_add(s2, v3_mulf(v3_sub(e2, s2), (float)alpha));
    return c;
}

int oracle_generate_dlscgc_dynamic(const dlscqp_config *cfg, const float *init_self, double radius_self, int n_dyn,
                                   const float *dyn_position, const float *dyn_velocity, const double *dyn_radius,
                                   const double *dyn_max_acc, const double *dyn_downwash, int size_prediction,
                                   double uncertainty_horizon, int obs_base, dlscqp_lsc_row *rows_out) {
    const int M = cfg->M, n = cfg->n, N = n + 1, phi = cfg->phi;
    float *pred = (float *)malloc(sizeof(float) * M * N * 3);
    double *size = (double *)malloc(sizeof(double) * M * N);
    int nr = 0;
    /* B_inv of polynomial.hpp:280-293 restated: power basis -> Bernstein, B_inv(k, i) = C(i,k) / C(n,k) */
    for (int o = 0; o < n_dyn; o++) {
        const double downwash = (radius_self + dyn_downwash[o] * dyn_radius[o]) / (radius_self + dyn_radius[o]); /* :1155-1157 */
        oracle_const_vel_plan(dyn_position + 3 * o, dyn_velocity + 3 * o, M, n, cfg->dt, pred); /* :293-303 */
        /* obstacleSizePredictionWithConstAcc, :338-368 */
        int Mu = (int)((uncertainty_horizon + 1e-9) / cfg->dt);
        if (Mu > M) Mu = M;
        for (int m = 0; m < M; m++)
            for (int i = 0; i < N; i++) {
                double sz = dyn_radius[o];
                if (size_prediction) {
                    if (m < Mu) {
                        double coef[3] = {0.5 * dyn_max_acc[o] * pow(m * cfg->dt, 2), dyn_max_acc[o] * m * cfg->dt * cfg->dt,
                                          0.5 * dyn_max_acc[o] * pow(cfg->dt, 2)};
                        double cp = 0, cik = 1, cnk = 1; /* C(i,k), C(n,k) */
                        for (int k = 0; k < 3; k++) {
                            if (k > 0) { cik = cik * (i - k + 1) / k; cnk = cnk * (n - k + 1) / k; }
                            cp += coef[k] * (i >= k ? cik / cnk : 0.0);
                        }
                        sz = dyn_radius[o] + cp;
                    } else {
                        sz = dyn_radius[o] + 0.5 * dyn_max_acc[o] * pow(Mu * cfg->dt, 2);
                    }
                }
                size[m * N + i] = sz;
            }
        for (int m = 0; m < M; m++) {
            /* normalVectorDynamicObs :1129-1148 -> normalVectorBetweenLines :1080-1100 */
            v3 os = ld(pred + (m * N) * 3), oe = ld(pred + (m * N + n) * 3);
            v3 as = ld(init_self + (m * N) * 3), ae = ld(init_self + (m * N + n) * 3);
            closest_t c = line_paths(os, oe, as, ae);
            v3 nt = v3_normalized(v3_sub(c.p2, c.p1));
            if (v3_norm(nt) == 0) {
                v3 a = v3_sub(as, os), b = v3_sub(ae, oe), ez = {0, 0, 1}, ex = {1, 0, 0};
                nt = (v3_norm(a) == 0 && v3_norm(b) == 0) ? ex : v3_cross(v3_sub(b, a), ez);
            }
            v3 nrm = {nt.x, nt.y, (float)((double)nt.z / downwash)}; /* :618-620 */
            for (int i = 0; i < N; i++) {
                if (m == 0 && i < phi) continue;
                dlscqp_lsc_row *r = &rows_out[nr++];
                memset(r, 0, sizeof *r);
                r->normal[0] = nrm.x; r->normal[1] = nrm.y; r->normal[2] = nrm.z;
                for (int k = 0; k < 3; k++) r->point[k] = pred[(m * N + i) * 3 + k];
                r->d = size[m * N + i] + radius_self; /* :623 */
                r->m = (int16_t)m; r->i = (int8_t)i; r->obs = (int16_t)(obs_base + o); r->slack = (int16_t)o;
                r->flags = DLSCQP_ROW_DYNAMIC;
            }
        }
    }
    free(pred);
    free(size);
    return nr;
}

/* ---- TrajPlanner::checkObstacleDisturbance (traj_planner.cpp:329-336) ------------------------------------------
 * pred: a neighbour's predicted trajectory [M][n+1][3]; if it starts farther than reset_threshold from the observed
 * position the obstacle is predicted to stand still there (planConstVelTraj(position, 0)).  Returns 1 if reset. */
int oracle_obstacle_disturbance(float *pred, int M, int n, const float obs_position[3], double reset_threshold) {
    float dx = pred[0] - obs_position[0], dy = pred[1] - obs_position[1], dz = pred[2] - obs_position[2];
    float sq = dx * dx + dy * dy + dz * dz; /* octomath::Vector3::norm */
    if (!(sqrt((double)sq) > reset_threshold)) return 0;
    for (int p = 0; p < M * (n + 1); p++)
        for (int k = 0; k < 3; k++) pred[3 * p + k] = obs_position[k];
    return 1;
}

/* ---- TrajPlanner::checkWaypointTrap (traj_planner.cpp:708-735) ---------------------------------------------------
 * rows: the agent's rows in populatebyrow order -- n_nbr neighbour agents, then n_dyn dynamic obstacles, `per` =
 * M (n+1) - phi rows each (rows (0, i < phi) are not stored).  sfc_last: {min xyz, max xyz} of corridor box M-1 or NULL.
 * Rows of the dynamic obstacles that can reach the waypoint are replaced by the default LSC (zero normal) if the sub-goal
 * or the waypoint is outside the feasible region of the last control point.  Returns the number of released obstacles. */
static int oracle_lsc_contains(const dlscqp_lsc_row *r, const float pt[3]) { /* LSC::isPointInLSC, collision_constraints.cpp:35-37 */
    float e0 = pt[0] - r->point[0], e1 = pt[1] - r->point[1], e2 = pt[2] - r->point[2];
    float dotv = e0 * r->normal[0] + e1 * r->normal[1] + e2 * r->normal[2];
    return (double)dotv - r->d > -1e-9;
}
static int oracle_box_contains(const float bmin[3], const float bmax[3], const float pt[3]) { /* Box::isPointInBox, :109-116 */
    for (int k = 0; k < 3; k++)
        if (!((double)pt[k] > (double)bmin[k] - 1e-5 && (double)pt[k] < (double)bmax[k] + 1e-5)) return 0;
    return 1;
}
static int oracle_feasible_last(const dlscqp_config *cfg, const dlscqp_lsc_row *rows, int n_nbr, int per, const float *sfc_last,
                                const float wp[3], const float pt[3], int first_plan) { /* isPointInFeasibleRegion(point, M-1, n), :586-598 */
    for (int oi = 0; oi < n_nbr; oi++) {
        const dlscqp_lsc_row *last = rows + (size_t)oi * per + (per - 1); /* (M-1, n) is the last row of an obstacle */
        if (!oracle_lsc_contains(last, pt)) return 0;
    }
    if (cfg->world_use_octomap && sfc_last && !oracle_box_contains(sfc_last, sfc_last + 3, pt)) return 0;
    float cmin[3] = {0, 0, 0}, cmax[3] = {0, 0, 0}; /* Box() until constructCommunicationRange sets it (:538-546), which only the
                                                      * non-initialising branch of generateSFC calls (traj_planner.cpp:692-705) */
    if (cfg->communication_range > 0 && cfg->world_use_octomap && !first_plan) {
        float dl = (float)(0.5 * cfg->communication_range);
        for (int k = 0; k < 3; k++) { cmin[k] = wp[k] - dl; cmax[k] = wp[k] + dl; }
    }
    return oracle_box_contains(cmin, cmax, pt);
}
int oracle_waypoint_trap(const dlscqp_config *cfg, dlscqp_lsc_row *rows, int n_nbr, int n_dyn, const dlscqp_agent *agent,
                         const float *sfc_last, const float *dyn_position, const float *dyn_velocity, const double *dyn_radius,
                         const double *dyn_max_acc, double uncertainty_horizon, int first_plan) {
    const int per = cfg->M * (cfg->n + 1) - cfg->phi;
    if (n_dyn <= 0) return 0;
    const float *cg = agent->current_goal_point, *wp = agent->next_waypoint;
    if (oracle_feasible_last(cfg, rows, n_nbr, per, sfc_last, wp, cg, first_plan) && oracle_feasible_last(cfg, rows, n_nbr, per, sfc_last, wp, wp, first_plan)) return 0;
    int released = 0;
    const double horizon = cfg->M * cfg->dt;
    for (int o = 0; o < n_dyn; o++) {
        int hit = 0; /* Obstacle::isCollided, obstacle.hpp:26-36 */
        const double step = 0.1 * horizon < 0.1 ? 0.1 * horizon : 0.1;
        for (double t = 0; t <= horizon && !hit; t += step) {
            float tf = (float)t;
            float ox = dyn_position[3 * o] + dyn_velocity[3 * o] * tf, oy = dyn_position[3 * o + 1] + dyn_velocity[3 * o + 1] * tf,
                  oz = dyn_position[3 * o + 2] + dyn_velocity[3 * o + 2] * tf;
            double tm = t < uncertainty_horizon ? t : uncertainty_horizon;
            float dx = ox - wp[0], dy = oy - wp[1], dz = oz - wp[2];
            double dist = sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz); /* Vector3::distance: double squares */
            if (dist < agent->radius + dyn_radius[o] + 0.5 * dyn_max_acc[o] * tm * tm) hit = 1;
        }
        if (!hit) continue;
        released++;
        for (int j = 0; j < per; j++) {
            dlscqp_lsc_row *r = rows + (size_t)(n_nbr + o) * per + j;
            for (int k = 0; k < 3; k++) { r->normal[k] = 0; r->point[k] = 0; }
            r->d = 0;
        }
    }
    return released;
}
