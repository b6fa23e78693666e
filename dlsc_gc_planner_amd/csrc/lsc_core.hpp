// lsc_core.hpp -- DLSC-GC half-spaces between two agents for one trajectory segment.
//
// Replaces, per (agent, neighbouring agent, segment m), the body of TrajPlanner::generateDLSCGC
// (/root/reference/src/traj_planner.cpp:603-666) and what it calls:
//   initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol  (:409-421, :304-314)  shift of the previous plans
//   Trajectory::planConstVelTraj                                (trajectory.cpp:78-91) first replanning step
//   downwashBetween / coordinateTransform                       (:1150-1161, :1183-1187, trajectory.cpp:206-219)
//   normalVectorBetweenPolys -> closestPointsBetweenPointAndConvexHull -> openGJK (:1102-1127, geometry.hpp:276-306)
//   closestPointsBetweenLineSegments / Lines / PointAndLineSegment                 (geometry.hpp:76-274)
// One work item = one (pair, segment): independent of every other item, so the kernel is a flat grid and each
// lane writes the 6 (3 for segment 0) rows of its item straight into the SoA row arrays the QP kernel reads.
// The reference computes in octomap::point3d (three float32, float32 operators; dot/norm return double) and
// calls openGJK in double; the same precisions are used here.  The point-to-hull problem is solved exactly by
// enumerating the hull's tetrahedra (containment), triangles, edges and vertices -- branch-light, no
// iteration count, identical to what GJK converges to (tests pin it against the reference's openGJK).
// Compiled with -ffp-contract=off.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/dlscqp.h"

#ifndef DQ_FN
#define DQ_FN inline
#endif
// DQ_WAVE_ALL(c): c holds for every lane of the wavefront (a wave-uniform value a real branch can test); host: c itself
#ifndef DQ_WAVE_ALL
#define DQ_WAVE_ALL(c) (c)
#endif
// DQ_UNIFORM_F64(x): x is computed from wave-uniform inputs only (the GPU build keeps it in scalar registers); host: x itself
#ifndef DQ_UNIFORM_F64
#define DQ_UNIFORM_F64(x) (x)
#endif
// DQ_FRESH_INT(i): (GPU build) the compiler forgets what it knows about i -- loads through addresses derived from it are issued again
// instead of their earlier results being kept in registers across the code in between; host: nothing
#ifndef DQ_FRESH_INT
#define DQ_FRESH_INT(i) ((void)0)
#endif

namespace dlscqp {

struct F3 {
    float x, y, z;
};
DQ_FN F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
DQ_FN F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DQ_FN F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DQ_FN F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DQ_FN double dot(F3 a, F3 b) { float s = a.x * b.x + a.y * b.y + a.z * b.z; return (double)s; }  // float32 inside
DQ_FN double norm(F3 a) { float s = a.x * a.x + a.y * a.y + a.z * a.z; return sqrt((double)s); }
DQ_FN double distance(F3 a, F3 b) {  // float32 differences, double squares (octomath::Vector3::distance)
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrt(dx * dx + dy * dy + dz * dz);
}
DQ_FN F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DQ_FN F3 normalized(F3 a) {
    double len = norm(a);
    if (len > 0) { float f = (float)len; a.x /= f; a.y /= f; a.z /= f; }
    return a;
}
DQ_FN bool same(F3 a, F3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// ---- closest point of conv{p_0..p_5} to the origin, exact ------------------------------------------------
struct D3 {
    double x, y, z;
};
DQ_FN D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
DQ_FN D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
DQ_FN double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DQ_FN double det3(D3 a, D3 b, D3 c) {
    return a.x * (b.y * c.z - b.z * c.y) - a.y * (b.x * c.z - b.z * c.x) + a.z * (b.x * c.y - b.y * c.x);
}

// A lane's six hull points (float32, as the trajectories are), component c of point a at base[(3 * a + c) * stride].  On the
// device the store is a slice of LDS (stride = workgroup size, so the lanes of a wavefront hit distinct banks): the loops
// below index the points with their counters, and register arrays indexed like that either go to scratch memory or force the
// compiler to unroll all 15 + 20 + 15 simplices into one 280-register body (measured: 1 wavefront per SIMD).
struct HullStore {
    float *base;
    int stride;
    DQ_FN void put(int a, F3 v) const {
        base[(3 * a) * stride] = v.x; base[(3 * a + 1) * stride] = v.y; base[(3 * a + 2) * stride] = v.z;
    }
    DQ_FN F3 getf(int a) const { return f3(base[(3 * a) * stride], base[(3 * a + 1) * stride], base[(3 * a + 2) * stride]); }
    DQ_FN D3 get(int a) const { const F3 v = getf(a); return d3(v.x, v.y, v.z); }
};

// x is THE closest point of a convex set to the origin iff p.x >= x.x for every point p of the set (for a hull: every
// vertex).  With the tolerance below a candidate that passes is within 1.5e-7 |x| of the exact projection, i.e. the same
// point to float32 resolution -- which is what the caller rounds it to.
constexpr double HULL_KKT_TOL = 1e-14;
DQ_FN bool hull6_is_projection(const HullStore &p, D3 x, double xx) {
    double lo = ddot(p.get(0), x);
#pragma unroll 1
    for (int a = 1; a < 6; a++) {
        const double s = ddot(p.get(a), x);
        lo = s < lo ? s : lo;
    }
    return lo >= xx * (1.0 - HULL_KKT_TOL);
}

DQ_FN D3 hull6_closest_to_origin(const HullStore &p) {
    D3 best = p.get(0);
    double bd = ddot(best, best);
    // The closest point of the hull's boundary: vertices, then edges, then faces -- each stage only if the best point so
    // far is not already the projection (a neighbour's relative hull is a thin sliver far from the origin: its closest
    // point is a vertex or lies on an edge, in 2-D always; the 20 faces are 60 % of the arithmetic).
#pragma unroll 1
    for (int a = 1; a < 6; a++) {
        const D3 pa = p.get(a);
        const double da = ddot(pa, pa);
        if (da < bd) { bd = da; best = pa; }
    }
    if (hull6_is_projection(p, best, bd)) return best;
#pragma unroll 1
    for (int a = 0; a < 6; a++) {
        const D3 pa = p.get(a);
#pragma unroll 1
        for (int b = a + 1; b < 6; b++) {
            const D3 ab = p.get(b) - pa;
            const double den = ddot(ab, ab), num = -ddot(pa, ab);
            if (den > 0) {
                const double t = num / den;
                if (t > 0 && t < 1) {
                    const D3 c = d3(pa.x + t * ab.x, pa.y + t * ab.y, pa.z + t * ab.z);
                    const double dc = ddot(c, c);
                    if (dc < bd) { bd = dc; best = c; }
                }
            }
        }
    }
    if (hull6_is_projection(p, best, bd)) return best;
#pragma unroll 1
    for (int a = 0; a < 6; a++) {
        const D3 pa = p.get(a);
#pragma unroll 1
        for (int b = a + 1; b < 6; b++) {
            const D3 ab = p.get(b) - pa;
            const double den = ddot(ab, ab), num = -ddot(pa, ab);
#pragma unroll 1
            for (int ci = b + 1; ci < 6; ci++) {
                const D3 ac = p.get(ci) - pa;
                const double g11 = den, g12 = ddot(ab, ac), g22 = ddot(ac, ac);
                const double r1 = num, r2 = -ddot(pa, ac);
                const double det = g11 * g22 - g12 * g12;
                if (det > 1e-14 * g11 * g22) {
                    const double u = (r1 * g22 - r2 * g12) / det, w = (g11 * r2 - g12 * r1) / det;
                    if (u > 0 && w > 0 && u + w < 1) {
                        const D3 c = d3(pa.x + u * ab.x + w * ac.x, pa.y + u * ab.y + w * ac.y, pa.z + u * ab.z + w * ac.z);
                        const double dc = ddot(c, c);
                        if (dc < bd) { bd = dc; best = c; }
                    }
                }
            }
        }
    }
    // Is the origin inside the hull?  If the plane through the origin with normal `best` has every point strictly on its
    // positive side, it separates the origin from the hull and the 15 tetrahedra need not be looked at (the usual case:
    // two vehicles whose relative hull contains the origin are in collision).
    double lo = ddot(p.get(0), best);
#pragma unroll 1
    for (int a = 1; a < 6; a++) {
        const double s = ddot(p.get(a), best);
        lo = s < lo ? s : lo;
    }
    if (lo > 0) return best;
    bool inside = false;
#pragma unroll 1
    for (int a = 0; a < 6; a++) {
        const D3 pa = p.get(a);
        const D3 oa = d3(-pa.x, -pa.y, -pa.z);
#pragma unroll 1
        for (int b = a + 1; b < 6; b++) {
            const D3 ab = p.get(b) - pa;
#pragma unroll 1
            for (int ci = b + 1; ci < 6; ci++) {
                const D3 ac = p.get(ci) - pa;
#pragma unroll 1
                for (int di = ci + 1; di < 6; di++) {  // origin inside this tetrahedron => inside the hull
                    const D3 ad = p.get(di) - pa;
                    const double vol = det3(ab, ac, ad);
                    if (fabs(vol) >= 1e-300) {
                        const double l1 = det3(oa, ac, ad) / vol, l2 = det3(ab, oa, ad) / vol, l3 = det3(ab, ac, oa) / vol;
                        if (l1 >= 0 && l2 >= 0 && l3 >= 0 && l1 + l2 + l3 <= 1) inside = true;
                    }
                }
            }
        }
    }
    return inside ? d3(0, 0, 0) : best;
}

// ---- geometry.hpp:76-111 ---------------------------------------------------------------------------------
struct Closest {
    double dist;
    F3 p1, p2;
};

DQ_FN Closest point_segment(F3 point, F3 ls, F3 le) {
    const F3 a = ls - point, b = le - point;
    F3 rel = a;
    double dist_min = norm(a);
    if (!same(a, b)) {
        double dist = norm(b);
        if (dist_min > dist) { dist_min = dist; rel = b; }
        const F3 n_line = normalized(b - a);
        const F3 c = a - n_line * (float)dot(a, n_line);
        dist = norm(c);
        if (dot(c - a, c - b) < 0 && dist_min > dist) { dist_min = dist; rel = c; }
    }
    Closest r;
    r.dist = dist_min;
    r.p1 = point;
    r.p2 = rel + point;
    return r;
}

// geometry.hpp:139-182 (Eigen::Matrix3f::inverse() = cofactors / determinant)
DQ_FN Closest lines(F3 s1, F3 e1, F3 s2, F3 e2) {
    Closest r;
    const F3 n1 = normalized(e1 - s1), n2 = normalized(e2 - s2);
    if (distance(n1, n2) < 1e-5 || distance(n1, f3(-n2.x, -n2.y, -n2.z)) < 1e-5) {
        F3 delta = s2 - s1;
        delta = delta - n1 * (float)dot(delta, n1);
        r.dist = norm(delta);
        r.p1 = s1;
        r.p2 = s1 + delta;
    } else {
        const F3 delta = s2 - s1;
        const F3 n3 = normalized(cross(n2, n1));
        // A = [n1 | -n2 | n3] (columns); alphas = A^{-1} delta
        const float a00 = n1.x, a01 = -n2.x, a02 = n3.x, a10 = n1.y, a11 = -n2.y, a12 = n3.y, a20 = n1.z, a21 = -n2.z, a22 = n3.z;
        const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
        const float c10 = a21 * a02 - a22 * a01, c11 = a22 * a00 - a20 * a02, c12 = a20 * a01 - a21 * a00;
        const float c20 = a01 * a12 - a02 * a11, c21 = a02 * a10 - a00 * a12, c22 = a00 * a11 - a01 * a10;
        const float det = a00 * c00 + a10 * c10 + a20 * c20;
        const float invdet = 1.0f / det;
        const float al0 = (c00 * invdet) * delta.x + (c10 * invdet) * delta.y + (c20 * invdet) * delta.z;
        const float al1 = (c01 * invdet) * delta.x + (c11 * invdet) * delta.y + (c21 * invdet) * delta.z;
        const float al2 = (c02 * invdet) * delta.x + (c12 * invdet) * delta.y + (c22 * invdet) * delta.z;
        r.dist = fabs((double)al2);
        r.p1 = s1 + n1 * al0;
        r.p2 = s2 + n2 * al1;
    }
    return r;
}

// geometry.hpp:184-274
DQ_FN Closest segments(F3 s1, F3 e1, F3 s2, F3 e2) {
    Closest r;
    if (distance(s1, e1) < 1e-5) {
        r = point_segment(s1, s2, e2);
    } else if (distance(s2, e2) < 1e-5) {
        r = point_segment(s2, s1, e1);
        const F3 t = r.p1; r.p1 = r.p2; r.p2 = t;
    } else {
        const F3 v1 = e1 - s1, v2 = e2 - s2;
        const double l1 = norm(v1), l2 = norm(v2);
        const F3 n1 = v1 * (float)(1 / l1), n2 = v2 * (float)(1 / l2);
        if (norm(cross(n1, n2)) < 1e-5) {  // parallel segments
            double bound_min = dot(s2 - s1, n1), bound_max = dot(e2 - s1, n1);
            F3 p2_min = s2, p2_max = e2;
            if (bound_max < bound_min) {
                const double t = bound_min; bound_min = bound_max; bound_max = t;
                const F3 tv = p2_min; p2_min = p2_max; p2_max = tv;
            }
            F3 delta = s2 - s1;
            delta = delta - n1 * (float)dot(delta, n1);
            if (l1 < bound_min) { r.p1 = e1; r.p2 = p2_min; }
            else if (bound_max < 0) { r.p1 = s1; r.p2 = p2_max; }
            else if (bound_min < 0) { r.p1 = s1; r.p2 = s1 + delta; }
            else { r.p1 = p2_min - delta; r.p2 = p2_min; }
            r.dist = distance(r.p1, r.p2);
        } else {
            r = lines(s1, e1, s2, e2);
            const double alpha1 = dot(r.p1 - s1, n1) / l1, alpha2 = dot(r.p2 - s2, n2) / l2;
            if (alpha1 < 0) r.p1 = s1; else if (alpha1 > 1) r.p1 = e1;
            if (alpha2 < 0) r.p2 = s2; else if (alpha2 > 1) r.p2 = e2;
            if (alpha1 < 0 || alpha1 > 1) {
                double dt = dot(n2, r.p1 - s2);
                dt = dt < 0 ? 0 : (dt > l2 ? l2 : dt);
                r.p2 = s2 + n2 * (float)dt;
            }
            if (alpha2 < 0 || alpha2 > 1) {
                double dt = dot(n1, r.p2 - s1);
                dt = dt < 0 ? 0 : (dt > l1 ? l1 : dt);
                r.p1 = s1 + n1 * (float)dt;
            }
            r.dist = distance(r.p1, r.p2);
        }
    }
    return r;
}

// ---- trajectories the half-spaces are built from ---------------------------------------------------------
struct LscGenArgs {  // device mirror of dlscqp_lsc_gen (include/dlscqp.h)
    int n_local, first_plan, M, n1, phi;
    double dt;
    const int32_t *local_ids;
    const float *plans, *position, *velocity, *goals;
    const double *radius, *downwash;
    const int32_t *nbr_offsets, *nbr_ids;
    const int32_t *nbr_counts;       // lists of fixed pitch: neighbours of local agent k (else nbr_offsets[k+1] - nbr_offsets[k])
    float *nx, *ny, *nz, *px, *py, *pz;
    double *d;
    uint32_t *key;
    int n_dyn, dyn_size_prediction;
    double dyn_uncertainty_horizon;
    const float *dyn_position, *dyn_velocity;
    const double *dyn_radius, *dyn_max_acc, *dyn_downwash;
    const float *observed_position;  // checkObstacleDisturbance, or null
    double reset_threshold;
    double comm_range;               // param.communication_range (waypoint trap)
    int use_sfc;
    // presolve (dlscqp_lsc_gen.prune_sfc_boxes): rows implied by the agent's own box rows are not written
    const float *prune_sfc;          // [n_local][M][6] corridor boxes of the local agents, or null
    int32_t *row_counts;             // [n_local] rows written per agent (with prune_sfc), or null
    int tail_from, tail_segments;    // neighbours at list positions >= tail_from contribute their last tail_segments segments only (0: off)
    double world_min[3], world_max[3];
    int dim;
};

// One item's rows before they are written: segment m, control points i0 .. 5 (one normal for all of them).  The arrays are
// indexed by the control point and every loop over them is unrolled, so that they live in registers, not in scratch memory.
struct LscRows {
    float nx, ny, nz;
    float px[6], py[6], pz[6];
    double d[6];
    uint32_t key[6];
    int i0;
};

constexpr double PRUNE_MARGIN = 1e-4;  // metres by which a row has to be implied before it is dropped

// Is row `i` of `r` implied by the rows of the model that confine control point (m, i) to a box -- the variable bounds
// (world box, traj_optimizer.cpp:251-265) and the corridor box of segment m (:385-410)?  min over the box of n.c is
// attained at a corner; if even that corner satisfies n.(c - p) - d >= PRUNE_MARGIN the row can never be active or
// violated at a point that satisfies the box rows, so the QP and the goal LP have the same solutions without it (what a
// presolve does; rows with a slack variable included: with the hard row implied, slack 0 is feasible and optimal).
// keep[i] for the (up to) six rows of an item, returns how many are kept.  The rows share normal and box, so the box part
// -- the smallest n.c over the box -- is computed once per item and only  d_i + n.p_i  per row.
DQ_FN int lsc_rows_kept(const LscGenArgs &g, const float *box, const LscRows &r, bool keep[6]) {
    const float nv[3] = {r.nx, r.ny, r.nz};
    // The last point of the last segment of an agent obstacle is also a row of the goal LP (goal_optimizer.cpp:176-195),
    // whose variable is not a control point: the LP has the faces of the last corridor box (:158-173) but no variable
    // bounds, so there only the corridor box may imply the row.
    const uint32_t key5 = r.key[5];  // all rows of an item: same segment, same kind of obstacle
    const bool last_agent_item = (int)(key5 & 0xFF) == g.M - 1 && !((key5 >> 12) & 1);
    double mn = 0.0, mn_lp = 0.0;
    for (int k = 0; k < 3; k++) {
        if (k >= g.dim) continue;
        double lo = g.world_min[k], hi = g.world_max[k];
        if (g.use_sfc) {
            lo = (double)box[k] > lo ? (double)box[k] : lo;
            hi = (double)box[3 + k] < hi ? (double)box[3 + k] : hi;
            mn_lp += (double)nv[k] * (nv[k] >= 0 ? (double)box[k] : (double)box[3 + k]);
        }
        mn += (double)nv[k] * (nv[k] >= 0 ? lo : hi);
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float pv[3] = {r.px[i], r.py[i], r.pz[i]};
        double rhs = r.d[i];
        for (int k = 0; k < 3; k++) {
            if (k >= g.dim) continue;
            rhs += (double)nv[k] * (double)pv[k];
        }
        const bool lp_row = last_agent_item && i == g.n1 - 1;
        const bool implied = lp_row ? (g.use_sfc && mn_lp - rhs >= PRUNE_MARGIN) : (mn - rhs >= PRUNE_MARGIN);
        keep[i] = i >= r.i0 && !implied;
        cnt += keep[i];
    }
    return cnt;
}

// control point i of segment m of agent `id`'s initial trajectory (= what the others predict for it)
DQ_FN F3 initial_point(const LscGenArgs &g, int id, int m, int i) {
    const int n = g.n1 - 1;
    if (g.first_plan) {  // planConstVelTraj: time advances by dt / n after every control point
        const float *p = g.position + 3 * (size_t)id, *v = g.velocity + 3 * (size_t)id;
        double time = 0;
        const int steps = m * g.n1 + i;
        const double tick = DQ_UNIFORM_F64(g.dt / n);
        for (int s = 0; s < steps; s++) time += tick;
        const float tf = (float)time;
        return f3(p[0] + v[0] * tf, p[1] + v[1] * tf, p[2] + v[2] * tf);
    }
    const float *pl = g.plans + (size_t)id * g.M * g.n1 * 3;
    const int mm = m < g.M - 1 ? m + 1 : g.M - 1, ii = m < g.M - 1 ? i : n;
    const float *c = pl + ((size_t)mm * g.n1 + ii) * 3;
    return f3(c[0], c[1], c[2]);
}

// What agent `id` is PREDICTED to do (obstaclePredictionWithPrevSol + checkObstacleDisturbance, traj_planner.cpp:290-336):
// its shifted previous plan, unless that starts farther than reset_threshold from where the agent is observed -- then it
// is taken to stand still at the observed position (planConstVelTraj(position, 0)).
DQ_FN F3 predicted_point(const LscGenArgs &g, int id, int m, int i) {
    if (g.observed_position) {  // uniform
        const float *o = g.observed_position + 3 * (size_t)id;
        const F3 obs = f3(o[0], o[1], o[2]);
        const F3 start = initial_point(g, id, 0, 0);
        if (norm(start - obs) > g.reset_threshold) return obs;
    }
    return initial_point(g, id, m, i);
}

DQ_FN void lsc_store_rows(const LscGenArgs &g, const LscRows &r, size_t at, const bool *keep) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        if (i < r.i0 || (keep && !keep[i])) continue;
        g.nx[at] = r.nx; g.ny[at] = r.ny; g.nz[at] = r.nz;
        g.px[at] = r.px[i]; g.py[at] = r.py[i]; g.pz[at] = r.pz[i];
        g.d[at] = r.d[i];
        g.key[at] = r.key[i];
        at++;
    }
}

// The last segment of a pair (m = M - 1): one half-space between the two straight paths "end of the plan -> sub-goal"
// (traj_planner.cpp:642-660), the same normal, point and distance for all n + 1 rows.
struct LscLast {
    float nx, ny, nz, px, py, pz;
    double dd;
};
DQ_FN LscLast lsc_last_geometry(const LscGenArgs &g, int self, int other) {
    const int n = g.n1 - 1, M = g.M;
    const double r_self = g.radius[self], r_oth = g.radius[other];
    const double collision_dist = r_oth + r_self;
    const double downwash = (g.downwash[self] * r_self + g.downwash[other] * r_oth) / (r_self + r_oth);
    const float dwf = (float)downwash;
    F3 l1s = predicted_point(g, other, M - 1, n), l2s = initial_point(g, self, M - 1, n);
    l1s.z = l1s.z / dwf;
    l2s.z = l2s.z / dwf;
    const float *go = g.goals + 3 * (size_t)other, *gs = g.goals + 3 * (size_t)self;
    // TrajPlanner::coordinateTransform of a point: double division, :1183-1187
    const F3 obs_goal = f3(go[0], go[1], (float)((double)go[2] / downwash));
    const F3 agent_goal = f3(gs[0], gs[1], (float)((double)gs[2] / downwash));
    const Closest c = segments(l1s, obs_goal, l2s, agent_goal);
    const F3 nt = normalized(c.p2 - c.p1);
    LscLast L;
    L.dd = 0.5 * (collision_dist + c.dist);
    L.nx = nt.x; L.ny = nt.y; L.nz = (float)((double)nt.z / downwash);
    L.px = c.p1.x; L.py = c.p1.y; L.pz = (float)((double)c.p1.z * downwash);
    return L;
}

// One (pair, segment) item: its rows into `out` (nothing is written to memory).  `last`: the pair's last-segment geometry
// if the caller has computed it beforehand (the GPU kernel does, for all pairs of an agent at once: among the items of a
// wavefront one in M is a last segment, and computed in place it sends the whole wavefront through the segment-to-segment
// code), else null.
// reload_pred (a constant at every call site): see below -- measured on one MI355X: the two-agents-per-wavefront kernel of the S64 closed
// loop gains (147 -> 114 registers, a fourth wavefront per SIMD: one loop's step 0.499 -> 0.487 ms), the many-items kernel of the crowd
// loses (0.0563 -> 0.0595 ms for the stress batch's rows, also with a fifth wavefront per SIMD), so only the former asks for it
DQ_FN void lsc_item_rows(const LscGenArgs &g, int self, int other, int m, LscRows &out, const HullStore &hull, const LscLast *last = nullptr,
                         const bool reload_pred = false) {
    const int M = g.M;
    const double r_self = g.radius[self], r_oth = g.radius[other];
    const double collision_dist = r_oth + r_self;                                                       // :605
    const double downwash = (g.downwash[self] * r_self + g.downwash[other] * r_oth) / (r_self + r_oth);  // :1150-1153
    const float dwf = (float)downwash;
    const float dw_inv = 1.0f / dwf;
    uint32_t dw_bits;
    memcpy(&dw_bits, &dwf, 4);
    const bool dw_pow2 = (dw_bits & 0x7FFFFFu) == 0 && dwf > 1e-30f && dwf < 1e30f;  // zero mantissa: +-2^e, reciprocal exact
    // rows of segment 0 start at i = phi (the others are never part of the model, traj_optimizer.cpp:417-419)
    out.i0 = m == 0 ? g.phi : 0;
    if (m < M - 1) {
        F3 pred[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            F3 a = initial_point(g, self, m, i);
            pred[i] = predicted_point(g, other, m, i);
            F3 b = pred[i];
            // Trajectory::coordinateTransform, trajectory.cpp:206-219.  The downwash factor is 1, 2 or another power of two in
            // every mission of the reference: dividing by it is then the same as multiplying by its reciprocal, bit for
            // bit, and a float32 division is a dozen instructions -- twelve of them per item.
            if (DQ_WAVE_ALL(dw_pow2)) {
                a.z = a.z * dw_inv;
                b.z = b.z * dw_inv;
            } else {
                a.z = a.z / dwf;
                b.z = b.z / dwf;
            }
            hull.put(i, a - b);
        }
        const D3 v = hull6_closest_to_origin(hull);
        F3 cp = f3((float)v.x, (float)v.y, (float)v.z);
        cp = f3(0, 0, 0) + cp;
        const F3 nt = normalized(cp);
        const F3 nrm = f3(nt.x, nt.y, (float)((double)nt.z / downwash));  // :634-636
        out.nx = nrm.x; out.ny = nrm.y; out.nz = nrm.z;
        if (reload_pred) {
            // the other agent's points are read again (L1 / L2) instead of held in eighteen registers across the closest-point iteration
            DQ_FRESH_INT(other);
#pragma unroll
            for (int i = 0; i < 6; i++) pred[i] = predicted_point(g, other, m, i);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {  // entries below i0 are computed too and never stored
            out.px[i] = pred[i].x; out.py[i] = pred[i].y; out.pz[i] = pred[i].z;
            out.d[i] = 0.5 * (collision_dist + dot(hull.getf(i), nt));  // :639-640
            out.key[i] = DLSCQP_KEY(m, i, -1, 0);
        }
    } else {
        LscLast L;
        if (last) L = *last;
        else L = lsc_last_geometry(g, self, other);
        out.nx = L.nx; out.ny = L.ny; out.nz = L.nz;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            out.px[i] = L.px; out.py[i] = L.py; out.pz[i] = L.pz;
            out.d[i] = L.dd;
            out.key[i] = DLSCQP_KEY(m, i, -1, 0);
        }
    }
}

// The item's rows at their fixed place: `row0` = index of the pair's first row in the SoA arrays.
DQ_FN void lsc_item(const LscGenArgs &g, int self, int other, int m, size_t row0, const HullStore &hull, const LscLast *last = nullptr) {
    LscRows r;
    lsc_item_rows(g, self, other, m, r, hull, last);
    lsc_store_rows(g, r, row0 + (m == 0 ? 0 : (size_t)(g.n1 - g.phi) + (size_t)(m - 1) * g.n1), nullptr);
}

// planConstVelTraj (trajectory.cpp:78-91): control point (m, i) of the constant-velocity prediction; time advances
// by dt / n after every control point, the last of a segment included
DQ_FN F3 const_vel_point(const float *p, const float *v, double dt, int n1, int m, int i) {
    double time = 0;
    const int steps = m * n1 + i;
    const double tick = DQ_UNIFORM_F64(dt / (n1 - 1));
    for (int s = 0; s < steps; s++) time += tick;
    const float tf = (float)time;
    return f3(p[0] + v[0] * tf, p[1] + v[1] * tf, p[2] + v[2] * tf);
}

// closestPointsBetweenLinePaths (geometry.hpp:113-137): the two points at the common parameter alpha where the
// straight paths line1 and line2 are closest
DQ_FN Closest line_paths(F3 s1, F3 e1, F3 s2, F3 e2) {
    const F3 rs = s2 - s1, re = e2 - e1;  // Line operator-
    const Closest rel = point_segment(f3(0, 0, 0), rs, re);
    const double len = distance(rs, re);  // Line::length
    const double alpha = len > 0 ? norm(rel.p2 - rs) / len : 0.0;
    Closest c;
    c.dist = rel.dist;
    c.p1 = s1 + (e1 - s1) * (float)alpha;
    c.p2 = s2 + (e2 - s2) * (float)alpha;
    return c;
}

// One (agent, dynamic obstacle, segment) item: traj_planner.cpp:616-627, rows into `out`.
DQ_FN void lsc_dyn_item_rows(const LscGenArgs &g, int self, int o, int m, LscRows &out) {
    const int n = g.n1 - 1, M = g.M;
    const double r_self = g.radius[self], r_obs = g.dyn_radius[o];
    const double downwash = (r_self + g.dyn_downwash[o] * r_obs) / (r_self + r_obs);  // :1155-1157
    const float *op = g.dyn_position + 3 * (size_t)o, *ov = g.dyn_velocity + 3 * (size_t)o;
    // normalVectorDynamicObs (:1129-1148) -> normalVectorBetweenLines (:1080-1100), untransformed coordinates
    const F3 os = const_vel_point(op, ov, g.dt, g.n1, m, 0), oe = const_vel_point(op, ov, g.dt, g.n1, m, n);
    const F3 as = initial_point(g, self, m, 0), ae = initial_point(g, self, m, n);
    const Closest c = line_paths(os, oe, as, ae);
    F3 nt = normalized(c.p2 - c.p1);
    if (norm(nt) == 0) {  // the reference's heuristic for coinciding paths
        const F3 a = as - os, b = ae - oe;
        nt = (norm(a) == 0 && norm(b) == 0) ? f3(1, 0, 0) : cross(b - a, f3(0, 0, 1));
    }
    const F3 nrm = f3(nt.x, nt.y, (float)((double)nt.z / downwash));  // :618-620
    // obstacleSizePredictionWithConstAcc (:338-368): radius + 0.5 a t^2 up to the uncertainty horizon, as Bernstein
    // control points (coef * B_inv with B_inv(k, i) = C(i,k) / C(n,k))
    int Mu = (int)((g.dyn_uncertainty_horizon + 1e-9) / g.dt);
    Mu = Mu < M ? Mu : M;
    const double amax = g.dyn_max_acc[o];
    out.i0 = m == 0 ? g.phi : 0;
    out.nx = nrm.x; out.ny = nrm.y; out.nz = nrm.z;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double size = r_obs;
        if (g.dyn_size_prediction) {
            if (m < Mu) {
                const double mt = m * g.dt;
                const double c0 = 0.5 * amax * (mt * mt), c1 = amax * m * g.dt * g.dt, c2 = 0.5 * amax * DQ_UNIFORM_F64(g.dt * g.dt);
                size = r_obs + (c0 + c1 * DQ_UNIFORM_F64((double)i / n) + c2 * DQ_UNIFORM_F64((double)(i * (i - 1)) / (double)(n * (n - 1))));
            } else {
                const double mt = DQ_UNIFORM_F64(Mu * g.dt);
                size = r_obs + 0.5 * amax * DQ_UNIFORM_F64(mt * mt);
            }
        }
        const F3 pt = const_vel_point(op, ov, g.dt, g.n1, m, i);
        out.px[i] = pt.x; out.py[i] = pt.y; out.pz[i] = pt.z;
        out.d[i] = size + r_self;  // :623
        out.key[i] = DLSCQP_KEY(m, i, o, 1);
    }
}

DQ_FN void lsc_dyn_item(const LscGenArgs &g, int self, int o, int m, size_t row0) {
    LscRows r;
    lsc_dyn_item_rows(g, self, o, m, r);
    lsc_store_rows(g, r, row0 + (m == 0 ? 0 : (size_t)(g.n1 - g.phi) + (size_t)(m - 1) * g.n1), nullptr);
}

// ---- TrajPlanner::checkWaypointTrap (traj_planner.cpp:708-735) ---------------------------------------------------
// LSC::isPointInLSC (collision_constraints.cpp:35-37): float32 differences and dot product, compared in double
DQ_FN bool point_in_lsc(float nx, float ny, float nz, float px, float py, float pz, double d, const float *pt) {
    const float e0 = pt[0] - px, e1 = pt[1] - py, e2 = pt[2] - pz;
    const float dotv = e0 * nx + e1 * ny + e2 * nz;
    return (double)dotv - d > -1e-9;
}
// Box::isPointInBox (collision_constraints.cpp:109-116)
DQ_FN bool point_in_box(const float *bmin, const float *bmax, const float *pt) {
    for (int k = 0; k < 3; k++)
        if (!((double)pt[k] > (double)bmin[k] - 1e-5 && (double)pt[k] < (double)bmax[k] + 1e-5)) return false;
    return true;
}
// Obstacle::isCollided (obstacle.hpp:26-36)
DQ_FN bool obstacle_reaches(const float *pos, const float *vel, double radius, double max_acc, const float *pt, double agent_radius,
                            double horizon, double uncertainty_horizon) {
    const double step = 0.1 * horizon < 0.1 ? 0.1 * horizon : 0.1;
    for (double t = 0; t <= horizon; t += step) {
        const float tf = (float)t;
        const F3 op = f3(pos[0] + vel[0] * tf, pos[1] + vel[1] * tf, pos[2] + vel[2] * tf);
        const double tm = t < uncertainty_horizon ? t : uncertainty_horizon;
        if (distance(op, f3(pt[0], pt[1], pt[2])) < agent_radius + radius + 0.5 * max_acc * tm * tm) return true;
    }
    return false;
}
// The part of isPointInFeasibleRegion(point, M-1, n) that does not depend on the rows: last corridor box and the
// communication-range box around the waypoint.  The reference sets that box in constructCommunicationRange, which only
// generateSFC's non-initialising branch calls (traj_planner.cpp:692-705, collision_constraints.cpp:538-546): it is still the
// default Box -- the origin -- on the first plan, without octomap corridors, and when the range is unlimited.
DQ_FN bool point_in_boxes(const LscGenArgs &g, const float *sfc_last, const float *wp, const float *pt) {
    if (g.use_sfc && sfc_last && !point_in_box(sfc_last, sfc_last + 3, pt)) return false;
    float cmin[3] = {0, 0, 0}, cmax[3] = {0, 0, 0};
    if (g.comm_range > 0 && g.use_sfc && !g.first_plan) {
        const float dl = (float)(0.5 * g.comm_range);
        for (int k = 0; k < 3; k++) { cmin[k] = wp[k] - dl; cmax[k] = wp[k] + dl; }
    }
    return point_in_box(cmin, cmax, pt);
}
// row r of a neighbour AGENT excludes the sub-goal or the waypoint from the feasible region of the last control point?
DQ_FN bool trap_row_excludes(const LscGenArgs &g, size_t r, const float *cg, const float *wp) {
    const uint32_t key = g.key[r];
    const int m = key & 0xFF, i = (key >> 8) & 0xF, dyn = (key >> 12) & 1;
    if (dyn || m != g.M - 1 || i != g.n1 - 1) return false;
    return !point_in_lsc(g.nx[r], g.ny[r], g.nz[r], g.px[r], g.py[r], g.pz[r], g.d[r], cg) ||
           !point_in_lsc(g.nx[r], g.ny[r], g.nz[r], g.px[r], g.py[r], g.pz[r], g.d[r], wp);
}
DQ_FN void trap_empty_row(const LscGenArgs &g, size_t r) {  // the default LSC: zero normal, zero point, d = 0 (:726-728)
    g.nx[r] = 0; g.ny[r] = 0; g.nz[r] = 0; g.px[r] = 0; g.py[r] = 0; g.pz[r] = 0;
    g.d[r] = 0;
}

}  // namespace dlscqp
