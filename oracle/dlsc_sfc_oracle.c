/*
 * dlsc_sfc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code): row N2 of SURVEY.md section 8(f),
 * safe-flight-corridor (SFC) boxes.
 *
 * Plain-C restatement of CollisionConstraints::initializeSFC / constructSFCFromInitialTraj /
 * expandSFCFromConvexHull / expandSFCIncrementally / isObstacleInSFC / isSFCInBoundary
 * (/root/reference/src/collision_constraints.cpp:435-452, 502-536, 781-901, 1023-1093) and of the world
 * MapManager::updateOctreeFromCSV builds (map_manager.cpp:264-316), on a dense occupancy grid.
 *
 * PARITY UNPINNED.  The reference asks octomap + DynamicEDT3D (third-party, absent here, versions = whatever
 * ros-*-octomap / dynamicEDT3D provide) for "the closest occupied cell" of a query point
 * (DynamicEDTOctomap::getDistanceAndClosestObstacle, maxdist 1.0 m, map_manager.cpp:14).  Restated here as: the
 * occupied cell with the smallest Euclidean distance between cell indices (query cell = floor(point / resolution)
 * as OcTreeBaseImpl::coordToKey computes it), distance capped at 1 m.  When several cells tie, DynamicEDT3D returns
 * whichever its wavefront propagation reached first; here the tie goes to the cell with the smallest L-infinity
 * distance to the query point (the conservative choice for the test at collision_constraints.cpp:880-886).
 * There are no golden vectors for this row in the reference; tests/test_sfc.py checks properties.
 *
 * point3d is octomath::Vector3: float32 components; Param::world_resolution etc. are double.
 */
#include "dlsc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SFC_EPS_F 1e-5 /* SP_EPSILON_FLOAT, sp_const.hpp:4 */

struct oracle_world {
    double res;
    float wmin[3], wmax[3];
    int lo[3], n[3]; /* cell index range [lo, lo+n) per axis, cell i covers [i res, (i+1) res) */
    unsigned char *occ;
    /* per query cell, computed on first use: squared index distance to the nearest occupied cell (-1: none within
     * maxdist, -2: not computed yet) and up to 8 of the cells at that distance (offsets packed 3 x int8) */
    int *near_d2;
    signed char *near_off; /* [cell][8][3] */
    unsigned char *near_n;
};

static int w_idx(const oracle_world *w, int i, int j, int k) {
    i -= w->lo[0]; j -= w->lo[1]; k -= w->lo[2];
    if (i < 0 || j < 0 || k < 0 || i >= w->n[0] || j >= w->n[1] || k >= w->n[2]) return -1;
    return (i * w->n[1] + j) * w->n[2] + k;
}

/* map_manager.cpp:264-316: every CSV row (centre x,y,z, size x,y,z) marks the cells [round((c - s/2)/res), round((c + s/2)/res)) */
oracle_world *oracle_world_from_boxes(const double *boxes, int n_boxes, double res, const float wmin[3], const float wmax[3]) {
    oracle_world *w = (oracle_world *)calloc(1, sizeof *w);
    w->res = res;
    for (int k = 0; k < 3; k++) {
        w->wmin[k] = wmin[k]; w->wmax[k] = wmax[k];
        w->lo[k] = (int)floor(wmin[k] / res) - 12; /* 1 m (maxdist) of slack around the world */
        w->n[k] = (int)ceil(wmax[k] / res) + 12 - w->lo[k];
    }
    const size_t ncell = (size_t)w->n[0] * w->n[1] * w->n[2];
    w->occ = (unsigned char *)calloc(ncell, 1);
    w->near_d2 = (int *)malloc(ncell * sizeof(int));
    for (size_t c = 0; c < ncell; c++) w->near_d2[c] = -2;
    w->near_off = (signed char *)calloc(ncell * 24, 1);
    w->near_n = (unsigned char *)calloc(ncell, 1);
    for (int b = 0; b < n_boxes; b++) {
        const double *c = boxes + 6 * b, *s = c + 3;
        int a0[3], a1[3];
        for (int k = 0; k < 3; k++) {
            a0[k] = (int)round((c[k] - 0.5 * s[k]) / res);
            a1[k] = (int)round((c[k] + 0.5 * s[k]) / res);
        }
        for (int i = a0[0]; i < a1[0]; i++)
            for (int j = a0[1]; j < a1[1]; j++)
                for (int k = a0[2]; k < a1[2]; k++) {
                    int id = w_idx(w, i, j, k);
                    if (id >= 0) w->occ[id] = 1;
                }
    }
    return w;
}

void oracle_world_free(oracle_world *w) {
    if (w) { free(w->occ); free(w->near_d2); free(w->near_off); free(w->near_n); free(w); }
}

/* getDistanceAndClosestObstacle + collision_constraints.cpp:878-886 for one search point: 1 if an obstacle is within
 * `margin` (L-infinity, to the cube of the closest occupied cell). */
static void nearest_cells(oracle_world *w, int q, const int c[3]) {
    const int R = (int)ceil(1.0 / w->res); /* maxdist 1.0 m */
    long best = -1;
    int n = 0;
    for (int di = -R; di <= R; di++)
        for (int dj = -R; dj <= R; dj++)
            for (int dk = -R; dk <= R; dk++) {
                int id = w_idx(w, c[0] + di, c[1] + dj, c[2] + dk);
                if (id < 0 || !w->occ[id]) continue;
                long d2 = (long)di * di + (long)dj * dj + (long)dk * dk;
                if (best >= 0 && d2 > best) continue;
                if (best < 0 || d2 < best) { best = d2; n = 0; }
                if (n < 8) {
                    w->near_off[(size_t)q * 24 + 3 * n] = (signed char)di;
                    w->near_off[(size_t)q * 24 + 3 * n + 1] = (signed char)dj;
                    w->near_off[(size_t)q * 24 + 3 * n + 2] = (signed char)dk;
                    n++;
                }
            }
    w->near_d2[q] = (int)best;
    w->near_n[q] = (unsigned char)n;
}

static int point_blocked(const oracle_world *cw, const float sp[3], double margin) {
    oracle_world *w = (oracle_world *)cw; /* the cache is logically const */
    int c[3];
    for (int k = 0; k < 3; k++) c[k] = (int)floor((1.0 / w->res) * (double)sp[k]); /* coordToKey */
    const int q = w_idx(w, c[0], c[1], c[2]);
    if (q < 0) return 0;
    if (w->near_d2[q] == -2) nearest_cells(w, q, c);
    if (w->near_d2[q] < 0) return 0;
    double best_linf = 1e30;
    for (int t = 0; t < w->near_n[q]; t++) {
        /* cube of that cell: centre (float) -+ 0.5 res (float), closest point of the cube to sp, L-infinity distance */
        double linf = 0;
        for (int k = 0; k < 3; k++) {
            const int cell = c[k] + w->near_off[(size_t)q * 24 + 3 * t + k];
            float centre = (float)((cell + 0.5) * w->res); /* keyToCoord */
            float delta = (float)(0.5 * w->res);
            float lo = centre - delta, hi = centre + delta;
            float cp = sp[k] < lo ? lo : (sp[k] > hi ? hi : sp[k]);
            double dd = fabsf(cp - sp[k]);
            if (linf < dd) linf = dd;
        }
        if (linf < best_linf) best_linf = linf;
    }
    const float dist = (float)(sqrt((double)w->near_d2[q]) * w->res);
    return dist < 1 && best_linf < margin + SFC_EPS_F;
}

/* isObstacleInSFC, collision_constraints.cpp:859-892 */
int oracle_sfc_obstacle_in_box(const oracle_world *w, const float bmin[3], const float bmax[3], double margin) {
    int mi[3];
    for (int k = 0; k < 3; k++) mi[k] = (int)floor((bmax[k] - bmin[k] + SFC_EPS_F) / w->res) + 1;
    for (int a = 0; a < mi[0]; a++)
        for (int b = 0; b < mi[1]; b++)
            for (int c = 0; c < mi[2]; c++) {
                int it[3] = {a, b, c};
                float sp[3];
                for (int k = 0; k < 3; k++) sp[k] = (float)(bmin[k] + it[k] * w->res);
                if (point_blocked(w, sp, margin)) return 1;
            }
    return 0;
}

/* isSFCInBoundary, collision_constraints.cpp:894-901 */
static int in_boundary(const oracle_world *w, const float bmin[3], const float bmax[3], double margin) {
    for (int k = 0; k < 3; k++)
        if (!(bmin[k] > w->wmin[k] + margin - SFC_EPS_F && bmax[k] < w->wmax[k] - margin + SFC_EPS_F)) return 0;
    return 1;
}

/* expandSFCIncrementally, collision_constraints.cpp:1023-1093.  box = {min[3], max[3]}. */
int oracle_sfc_expand(const oracle_world *w, const float init[6], double margin, double grid_resolution, double agent_max_vel,
                      double dt, float out[6]) {
    if (oracle_sfc_obstacle_in_box(w, init, init + 3, margin)) return 0;
    float sfc[6], cand[6], upd[6];
    int axis_cand[6] = {0, 1, 2, 3, 4, 5}, n_cand = 6; /* -x -y -z +x +y +z */
    const double mx = 2 * grid_resolution > agent_max_vel * dt ? 2 * grid_resolution : agent_max_vel * dt;
    const int max_iter = (int)round(mx / w->res) + 1;
    int axis_iter[6] = {0, 0, 0, 0, 0, 0};
    int i = -1, axis;
    memcpy(sfc, init, sizeof sfc);
    while (n_cand > 0) {
        memcpy(cand, sfc, sizeof cand);
        memcpy(upd, sfc, sizeof upd);
        while (in_boundary(w, upd, upd + 3, 0) && !oracle_sfc_obstacle_in_box(w, upd, upd + 3, margin)) {
            i++;
            if (i >= n_cand) i = 0;
            axis = axis_cand[i];
            memcpy(sfc, cand, sizeof sfc);
            memcpy(upd, cand, sizeof upd);
            if (axis < 3) {
                upd[3 + axis] = cand[axis];
                cand[axis] = (float)(cand[axis] - w->res);
                upd[axis] = cand[axis];
            } else {
                upd[axis - 3] = cand[axis];
                cand[axis] = (float)(cand[axis] + w->res);
                upd[axis] = cand[axis];
            }
            axis_iter[axis]++;
            if (axis_iter[axis] > max_iter) break;
        }
        /* the current axis direction is blocked (or exhausted): drop it */
        for (int t = i; t + 1 < n_cand; t++) axis_cand[t] = axis_cand[t + 1];
        n_cand--;
        if (i > 0) i--; else i = n_cand - 1;
    }
    /* SFC margin compensation */
    const double delta = margin - ((int)(margin / w->res) * w->res);
    for (int k = 0; k < 3; k++) {
        if (sfc[k] > w->wmin[k] + SFC_EPS_F) sfc[k] = (float)(sfc[k] - delta);
        if (sfc[3 + k] < w->wmax[k] - SFC_EPS_F) sfc[3 + k] = (float)(sfc[3 + k] + delta);
    }
    memcpy(out, sfc, sizeof sfc);
    return 1;
}

static int point_in_box(const float b[6], const float p[3]) { /* Box::isPointInBox, :109-116 */
    for (int k = 0; k < 3; k++)
        if (!(p[k] > b[k] - SFC_EPS_F && p[k] < b[3 + k] + SFC_EPS_F)) return 0;
    return 1;
}

static int superset_of_hull(const float b[6], const float *pts, int n) { /* Box::isSuperSetOfConvexHull, :170-185 */
    for (int k = 0; k < 3; k++) {
        float mn = pts[k], mxv = pts[k];
        for (int p = 1; p < n; p++) {
            if (pts[3 * p + k] < mn) mn = pts[3 * p + k];
            if (pts[3 * p + k] > mxv) mxv = pts[3 * p + k];
        }
        if (mn < b[k] - SFC_EPS_F || mxv > b[3 + k] + SFC_EPS_F) return 0;
    }
    return 1;
}

/* expandSFCFromConvexHull, both overloads (collision_constraints.cpp:781-857): prev == NULL -> the first one */
int oracle_sfc_from_hull(const oracle_world *w, const float *pts, int n_pts, const float *prev, double margin,
                         double grid_resolution, double agent_max_vel, double dt, float out[6]) {
    if (n_pts <= 0) return 0;
    float b[6];
    for (int k = 0; k < 3; k++) b[k] = b[3 + k] = pts[k];
    for (int p = 0; p < n_pts; p++)
        for (int k = 0; k < 3; k++) {
            if (pts[3 * p + k] < b[k]) b[k] = pts[3 * p + k];
            if (pts[3 * p + k] > b[3 + k]) b[3 + k] = pts[3 * p + k];
        }
    if (!prev) {
        for (int k = 0; k < 3; k++) {
            b[k] = (float)(round(b[k] / w->res) * w->res);
            b[3 + k] = (float)(round(b[3 + k] / w->res) * w->res);
        }
        int ok = oracle_sfc_expand(w, b, margin, grid_resolution, agent_max_vel, dt, out);
        if (ok && !superset_of_hull(out, pts, n_pts)) ok = 0;
        return ok;
    }
    for (int k = 0; k < 3; k++) {
        b[k] = (float)(floor(b[k] / w->res) * w->res);
        b[3 + k] = (float)(ceil(b[3 + k] / w->res) * w->res);
    }
    if (!(point_in_box(prev, b) && point_in_box(prev, b + 3))) { /* not sfc_prev.include(sfc_initial) */
        for (int k = 0; k < 3; k++) {
            float lo = prev[k] > b[k] ? prev[k] : b[k], hi = prev[3 + k] < b[3 + k] ? prev[3 + k] : b[3 + k];
            b[k] = (float)(ceil((lo - SFC_EPS_F) / w->res) * w->res);
            b[3 + k] = (float)(floor((hi + SFC_EPS_F) / w->res) * w->res);
        }
    }
    return oracle_sfc_expand(w, b, margin, grid_resolution, agent_max_vel, dt, out);
}

/* initializeSFC, collision_constraints.cpp:435-452: sfcs [M][6] */
int oracle_sfc_initialize(const oracle_world *w, const float pos[3], int M, double margin, double grid_resolution,
                          double agent_max_vel, double dt, float *sfcs) {
    float b[6], e[6];
    for (int k = 0; k < 3; k++) {
        b[k] = (float)(floor(pos[k] / w->res) * w->res);
        b[3 + k] = (float)(ceil(pos[k] / w->res) * w->res);
    }
    if (!oracle_sfc_expand(w, b, margin, grid_resolution, agent_max_vel, dt, e)) return 0;
    for (int m = 0; m < M; m++) memcpy(sfcs + 6 * m, e, sizeof e);
    return 1;
}

/* constructSFCFromInitialTraj, collision_constraints.cpp:502-536.  initial_traj [M][n+1][3]; sfcs [M][6] in/out.
 * Returns 0: greedy box (hull incl. waypoint), 1: box inside the previous one, 2: previous box reused. */
int oracle_sfc_construct(const oracle_world *w, const float *initial_traj, int M, int n, const float goal[3],
                         const float waypoint[3], double margin, double grid_resolution, double agent_max_vel, double dt,
                         float *sfcs) {
    const int N = n + 1;
    for (int m = 0; m < M - 1; m++) memcpy(sfcs + 6 * m, sfcs + 6 * (m + 1), 6 * sizeof(float));
    for (int m = 0; m < M - 2; m++) /* minor refinement */
        if (superset_of_hull(sfcs + 6 * (m + 1), initial_traj + (size_t)m * N * 3, N))
            memcpy(sfcs + 6 * m, sfcs + 6 * (m + 1), 6 * sizeof(float));
    float hull[9];
    memcpy(hull, initial_traj + ((size_t)(M - 1) * N + n) * 3, 3 * sizeof(float)); /* lastPoint */
    memcpy(hull + 3, goal, 3 * sizeof(float));
    memcpy(hull + 6, waypoint, 3 * sizeof(float));
    float upd[6];
    int code = 0;
    if (!oracle_sfc_from_hull(w, hull, 3, NULL, margin, grid_resolution, agent_max_vel, dt, upd)) {
        code = 1;
        if (!oracle_sfc_from_hull(w, hull, 2, sfcs + 6 * (M - 1), margin, grid_resolution, agent_max_vel, dt, upd)) {
            code = 2;
            memcpy(upd, sfcs + 6 * (M - 1), sizeof upd);
        }
    }
    memcpy(sfcs + 6 * (M - 1), upd, sizeof upd);
    return code;
}
