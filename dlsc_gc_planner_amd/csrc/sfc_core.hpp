// sfc_core.hpp -- safe-flight-corridor boxes, the code one wavefront runs for one agent.
//
// CollisionConstraints::initializeSFC / constructSFCFromInitialTraj / expandSFCFromConvexHull /
// expandSFCIncrementally / isObstacleInSFC / isSFCInBoundary of the reference
// (src/collision_constraints.cpp:435-452, 502-536, 781-901, 1023-1093) on a dense occupancy grid: the octomap +
// DynamicEDT3D query "closest occupied cell of a point" is a table, built once per world on the host
// (sfc_world.cpp), of the occupied cells nearest to each grid cell.  Control flow is wave-uniform (one box at a
// time, exactly the reference's sequence of trial expansions); the lanes share the lattice points of the slab
// that is being tested.  Float32 boxes / double parameters as in the reference.  Included by dlscqp.hip (GPU) and by
// tests/emul/emul.cpp (lanes run one after another on the host).
#pragma once
#include <cmath>
#include <cstdint>

#ifndef DQ_FN
#define DQ_FN inline
#endif

namespace dlscqp {

constexpr double SFC_EPS_F = 1e-5;  // SP_EPSILON_FLOAT, sp_const.hpp:4

DQ_FN int sfc_ctz64(uint64_t v) { return __builtin_ctzll(v); }

struct SfcWorldView {  // device (or host) pointers of a world built by build_sfc_world (sfc_world.cpp)
    double res;
    float wmin[3], wmax[3];
    int lo[3], n[3];            // cell index range per axis; cell i covers [i res, (i+1) res)
    // [cell][8] 32-bit words, one 32-byte record per cell so that a query is two 16-byte loads instead of 26 scattered ones:
    // word 0 = squared index distance to the nearest occupied cell (-1: none within 1 m), word 1 = how many cells at that
    // distance (up to 8), words 2..7 = their offsets, 8 x 3 signed bytes
    const uint32_t *near;
    // [cell] one word, eight nibbles: nibble (dx + 2 dy + 4 dz) = the distance, in whole cells, from corner (dx, dy, dz) of the
    // cell to the nearest of the record's occupied cells -- what sfc_point_blocked's arithmetic comes to for a lattice point
    // (sfc_corner_gaps) -- or 15 when the record has no occupied cell within 1 m.  The corridor routines only ever test lattice
    // points of the grid (box corners are multiples of the resolution), so a query is this one word.  nullptr: not built.
    const uint32_t *gaps;
    // [15] pointers, entry g = the summed-volume table for agents with sfc_gap_max = g, or nullptr (not built; the array itself
    // may be nullptr).  The table counts, over the lattice points I of the grid (n_k + 1 per axis), those that `gaps` calls blocked
    // in ANY of the eight (cell, corner) readings of I -- which reading a box test takes is float32 rounding of the box's corner
    // (see sfc_corner_gaps), so a count of zero over a box's lattice range means "free" whatever the rounding, in eight reads;
    // a positive count proves nothing and the points are tested one by one as before.  Layout: (n_0 + 2)(n_1 + 2)(n_2 + 2) int32,
    // entry [i][j][k] = number of such points with local lattice index < i, < j, < k (sfc_world.cpp: build_sfc_svt).
    const int32_t *const *svt;
};

DQ_FN int sfc_near_offset(const uint32_t *rec, int i) {  // signed byte i (0..23) of the offsets of a record
    const uint32_t wv = rec[2 + (i >> 2)];
    return (int)(int8_t)((wv >> (8 * (i & 3))) & 0xFFu);
}

struct SfcParams {
    double margin;           // agent radius
    double grid_resolution;  // param.grid_resolution
    double agent_max_vel;
    double dt;
    // what every box test needs of the above and of the world, computed once per agent (sfc_prepare): the divisions would
    // otherwise sit between any two trial expansions
    double inv_res;      // 1 / world resolution
    int gap_max;         // sfc_gap_max(margin, world resolution)
    const int32_t *svt;  // SfcWorldView::svt[gap_max] or nullptr
};

// getDistanceAndClosestObstacle + collision_constraints.cpp:878-886 for one search point.  Written without
// lane-dependent branches (clamped table index, fixed trip count, selects): the lanes of a wave test different points.
template <class W>
DQ_FN bool sfc_point_blocked(const W &wave, const SfcWorldView &w, const float sp[3], double margin) {
    int c[3];
    for (int k = 0; k < 3; k++) c[k] = (int)floor((1.0 / w.res) * (double)sp[k]);  // OcTreeBaseImpl::coordToKey
    const int i = c[0] - w.lo[0], j = c[1] - w.lo[1], kk = c[2] - w.lo[2];
    const bool inside = (i >= 0) & (j >= 0) & (kk >= 0) & (i < w.n[0]) & (j < w.n[1]) & (kk < w.n[2]);
    // clamped cell for the table reads of points outside the grid (masked by `inside` at the end); written with
    // min/max, not with ?: on the whole index -- that came out as an EXEC-masked branch
    const int ic = i < 0 ? 0 : (i >= w.n[0] ? w.n[0] - 1 : i);
    const int jc = j < 0 ? 0 : (j >= w.n[1] ? w.n[1] - 1 : j);
    const int kc = kk < 0 ? 0 : (kk >= w.n[2] ? w.n[2] - 1 : kk);
    const size_t q = ((size_t)ic * w.n[1] + jc) * w.n[2] + kc;
    uint32_t rec[8];
#pragma unroll
    for (int t = 0; t < 8; t++) rec[t] = w.near[q * 8 + t];
    const int d2 = (int)rec[0];
    const int nt = (int)rec[1];
    const float dist = (float)(sqrt((double)(d2 < 0 ? 0 : d2)) * w.res);
    const bool near_obstacle = inside & (d2 >= 0) & (dist < 1);
    // a record lists up to 8 nearest cells, most list one or two: the loop runs to the largest count among the points the
    // wavefront tests (wave-uniform bound; 0 when none of them has an occupied cell within 1 m)
    const int nt_max = wave.uniform_max8(near_obstacle ? nt : 0);
    double best = 1e30;
    for (int t = 0; t < nt_max; t++) {
        double linf = 0;
        for (int k = 0; k < 3; k++) {
            const int cell = c[k] + sfc_near_offset(rec, 3 * t + k);
            const float centre = (float)((cell + 0.5) * w.res);  // keyToCoord
            const float delta = (float)(0.5 * w.res);
            const float lo = centre - delta, hi = centre + delta;
            const float cp = sp[k] < lo ? lo : (sp[k] > hi ? hi : sp[k]);  // Box::closestPoint
            const double dd = fabsf(cp - sp[k]);                             // LInfinityDistance, util.hpp:131-140
            linf = linf < dd ? dd : linf;
        }
        best = ((t < nt) & (linf < best)) ? linf : best;
    }
    return near_obstacle & (best < margin + SFC_EPS_F);
}

// What sfc_point_blocked computes for a LATTICE point, as integers.  A lattice point is a corner of grid cells: sp_k = I_k res up
// to float32 rounding, and coordToKey's floor puts it into cell c_k = I_k or I_k - 1 -- whichever the rounding decides: that
// choice is the reference's and is kept (it picks the record that is read).  Given the cell, everything else is exact: the
// L-infinity distance from the corner I = c + delta to an occupied cell O of the record is max_k gap_k cells with
// gap_k = O_k - I_k (O_k >= I_k) or I_k - O_k - 1, and float32 noise (1e-6 m at +-20 m) cannot move gap x res across
// margin + 1e-5 unless the margin sits within 1e-5 of a multiple of the resolution from below.  Host, once per world:
// nibble of corner delta = min over the record's cells of max_k gap_k (15: none, or none within 1 m).
DQ_FN uint32_t sfc_corner_gaps(const uint32_t *rec, double res) {
    const int d2 = (int)rec[0], nt = (int)rec[1];
    const float dist = (float)(sqrt((double)(d2 < 0 ? 0 : d2)) * res);
    const bool near_obstacle = (d2 >= 0) & (dist < 1);
    uint32_t word = 0;
    for (int corner = 0; corner < 8; corner++) {
        int best = 15;
        for (int t = 0; near_obstacle && t < nt; t++) {
            int linf = 0;
            for (int k = 0; k < 3; k++) {
                const int off = sfc_near_offset(rec, 3 * t + k) - ((corner >> k) & 1);  // O_k - I_k
                const int gap = off >= 0 ? off : -off - 1;
                linf = gap > linf ? gap : linf;
            }
            best = linf < best ? linf : best;
        }
        word |= (uint32_t)(best > 15 ? 15 : best) << (4 * corner);
    }
    return word;
}

// sfc_point_blocked for a lattice point through the table of sfc_corner_gaps: the same cell as the reference's floor, the corner
// from the nearest integer of sp / res, one word read.
template <class W>
DQ_FN bool sfc_lattice_point_blocked(const W &wave, const SfcWorldView &w, const float sp[3], int gap_max) {
    int c[3], corner = 0;
    for (int k = 0; k < 3; k++) {
        const double sc = (1.0 / w.res) * (double)sp[k];
        c[k] = (int)floor(sc);  // OcTreeBaseImpl::coordToKey
        int d = (int)floor(sc + 0.5) - c[k];
        d = d < 0 ? 0 : (d > 1 ? 1 : d);
        corner |= d << k;
    }
    const int i = c[0] - w.lo[0], j = c[1] - w.lo[1], kk = c[2] - w.lo[2];
    const bool inside = (i >= 0) & (j >= 0) & (kk >= 0) & (i < w.n[0]) & (j < w.n[1]) & (kk < w.n[2]);
    const int ic = i < 0 ? 0 : (i >= w.n[0] ? w.n[0] - 1 : i);
    const int jc = j < 0 ? 0 : (j >= w.n[1] ? w.n[1] - 1 : j);
    const int kc = kk < 0 ? 0 : (kk >= w.n[2] ? w.n[2] - 1 : kk);
    return inside & wave.corner_blocked(w, ic, jc, kc, corner, gap_max);
}

// the largest whole number of cells g with g res < margin + 1e-5 (-1: not even 0), what `best < margin + SFC_EPS_F` admits
DQ_FN int sfc_gap_max(double margin, double res) {
    int g = (int)floor((margin + SFC_EPS_F) / res);
    if (!((double)g * res < margin + SFC_EPS_F)) g--;  // an exact multiple: strictly below
    return g > 14 ? 14 : g;
}

// One axis of sfc_lattice_point_blocked, for the lattice index `it` of a box that starts at b_k: which cell the reference's floor
// picks along the axis and which of the cell's two ends the point is.  Packed so that the three axes' entries ADD to the query:
// bits 0..25 the clamped cell index times the axis' stride in the table, bit 26 + k the corner bit of axis k, bits 29..30 count
// the axes along which the point lies outside the grid (a sum of at most three ones).
constexpr int SFC_AXIS_CAP = 64;               // lattice points per axis the entry tables hold
constexpr uint32_t SFC_GAPS_MAX_CELLS = 1u << 26;  // the offset field; larger worlds keep the 32-byte records (gaps == nullptr)
DQ_FN uint32_t sfc_axis_entry(const SfcWorldView &w, double inv_res, int k, float bk, int it) {
    const float sp = (float)(bk + it * w.res);
    const double sc = inv_res * (double)sp;  // inv_res = 1.0 / w.res, as sfc_lattice_point_blocked forms it
    const int c = (int)floor(sc);  // OcTreeBaseImpl::coordToKey
    int d = (int)floor(sc + 0.5) - c;
    d = d < 0 ? 0 : (d > 1 ? 1 : d);
    const int i = c - w.lo[k];
    const uint32_t outside = (uint32_t)((i < 0) | (i >= w.n[k]));
    const int ic = i < 0 ? 0 : (i >= w.n[k] ? w.n[k] - 1 : i);
    const uint32_t stride = k == 0 ? (uint32_t)(w.n[1] * w.n[2]) : (k == 1 ? (uint32_t)w.n[2] : 1u);
    return (uint32_t)ic * stride + ((uint32_t)d << (26 + k)) + (outside << 29);
}

// isObstacleInSFC (:859-892): the lattice points of the box, one per lane.  W::any_of(n, f): true if f(i) holds for some i in
// [0, n) (GPU: every thread of the workgroup takes its share of the points, one OR at the end; host: a loop).
// A point's coordinates are functions of one axis each, and the reference computes them in float32 / double from the box's own
// corner (the rounding decides which cell is read): with the lattice-point table the per-axis part -- everything but one add, one
// table word and one compare -- is done once per box and axis (W::axis_fill / W::axis_entry) instead of once per point.
DQ_FN void sfc_prepare(const SfcWorldView &g, SfcParams &p) {
    p.inv_res = 1.0 / g.res;
    p.gap_max = sfc_gap_max(p.margin, g.res);
    p.svt = nullptr;
    if (g.svt && g.gaps && p.gap_max >= 0) p.svt = g.svt[p.gap_max];  // uniform
}

// The lattice index (nearest whole multiple of the resolution) of point `it` along one axis of a box that starts at bk: the cell
// and corner bit of sfc_axis_entry, added.  Non-decreasing in `it`.
DQ_FN int sfc_axis_lattice(const SfcWorldView &w, double inv_res, float bk, int it) {
    const float sp = (float)(bk + it * w.res);
    const double sc = inv_res * (double)sp;
    const int c = (int)floor(sc);
    int d = (int)floor(sc + 0.5) - c;
    d = d < 0 ? 0 : (d > 1 ? 1 : d);
    return c + d;
}

// How many lattice points of the box are blocked in some reading (SfcWorldView::svt; p.svt != nullptr): 0 = the box test's
// answer is "free" (also for a box without points, which the reference's loops do not enter).  No branches, every table index
// clamped into the table: the lanes of a wavefront may each ask about a box of their own (sfc_expand).
DQ_FN int sfc_box_count(const SfcWorldView &g, const SfcParams &p, const float *b) {
    int a[3], e[3];  // local lattice range per axis, [a, e)
    bool some = true;
    for (int k = 0; k < 3; k++) {
        const int mi = (int)floor((b[3 + k] - b[k] + SFC_EPS_F) * p.inv_res) + 1;
        some &= mi > 0;
        const int first = sfc_axis_lattice(g, p.inv_res, b[k], 0) - g.lo[k];
        const int last = sfc_axis_lattice(g, p.inv_res, b[k], (mi > 0 ? mi : 1) - 1) - g.lo[k];
        const int top = g.n[k] + 1;
        a[k] = first < 0 ? 0 : (first > top ? top : first);
        e[k] = last + 1 < 0 ? 0 : (last + 1 > top ? top : last + 1);
        some &= e[k] > a[k];  // else all of the box lies outside the grid along this axis: no point is inside, none is blocked
    }
    const int s1 = g.n[2] + 2, s0 = (g.n[1] + 2) * s1;
    const int32_t *t = p.svt;
    const int count = t[e[0] * s0 + e[1] * s1 + e[2]] - t[a[0] * s0 + e[1] * s1 + e[2]] - t[e[0] * s0 + a[1] * s1 + e[2]] -
                      t[e[0] * s0 + e[1] * s1 + a[2]] + t[a[0] * s0 + a[1] * s1 + e[2]] + t[a[0] * s0 + e[1] * s1 + a[2]] +
                      t[e[0] * s0 + a[1] * s1 + a[2]] - t[a[0] * s0 + a[1] * s1 + a[2]];
    return some ? count : 0;
}

template <class W>
DQ_FN bool sfc_obstacle_in_box(const W &w, const SfcWorldView &g, const float *b, const SfcParams &p, bool use_svt = true) {
    const double margin = p.margin;
    int mi[3];
    // (the reference divides by the resolution; the quotient is a whole number of cells + 1e-4, float32 noise of the box edges 1e-5 of
    // a cell: multiplying by the reciprocal -- a third of the instructions, and this runs between any two trial expansions -- lands
    // on the same side of the floor)
    const double inv_res = p.inv_res;
    for (int k = 0; k < 3; k++) mi[k] = (int)floor((b[3 + k] - b[k] + SFC_EPS_F) * inv_res) + 1;
    // (a box that is inside out along an axis has no points: the reference's three nested loops do not run)
    const bool has_points = w.uniform((int)((mi[0] > 0) & (mi[1] > 0) & (mi[2] > 0))) != 0;
    if (!has_points) return false;  // uniform
    if (use_svt && p.svt && w.uniform(sfc_box_count(g, p, b)) == 0) return false;  // uniform
    const int total = mi[0] * mi[1] * mi[2];
    const int gap_max = p.gap_max;
    const bool small = (mi[0] <= SFC_AXIS_CAP) & (mi[1] <= SFC_AXIS_CAP) & (mi[2] <= SFC_AXIS_CAP);
    if (g.gaps && small) {  // uniform
        w.axis_fill(g, inv_res, b);
        // idx -> (a1, b2, c3) without integer division: idx + 0.5 is exact in float32 (idx < 2^18), the product with the rounded
        // reciprocal is off by at most (idx / m) 2^-23 < 1 / (32 m), and (idx + 0.5) / m is never closer than 1 / (2 m) to a whole
        // number -- the truncation is floor(idx / m) (and stays it with a reciprocal that is an ulp off: W::rcp)
        const float r2 = w.rcp((float)mi[2]), r1 = w.rcp((float)mi[1]);
        return w.any_of(total, [&](int idx) {
            const int ab = (int)(((float)idx + 0.5f) * r2), c3 = idx - ab * mi[2];
            const int a1 = (int)(((float)ab + 0.5f) * r1), b2 = ab - a1 * mi[1];
            const uint32_t s = w.axis_entry(0, a1) + w.axis_entry(1, b2) + w.axis_entry(2, c3);
            const uint32_t word = g.gaps[s & (SFC_GAPS_MAX_CELLS - 1u)];
            return ((s >> 29) == 0u) & ((int)((word >> (4u * ((s >> 26) & 7u))) & 15u) <= gap_max);
        });
    }
    return w.any_of(total, [&](int idx) {
        const int c3 = idx % mi[2], ab = idx / mi[2], b2 = ab % mi[1], a1 = ab / mi[1];
        const int it[3] = {a1, b2, c3};
        float sp[3];
        for (int k = 0; k < 3; k++) sp[k] = (float)(b[k] + it[k] * g.res);
        if (g.gaps) return sfc_lattice_point_blocked(w, g, sp, gap_max);  // uniform
        return sfc_point_blocked(w, g, sp, margin);
    });
}

DQ_FN bool sfc_in_boundary(const SfcWorldView &g, const float *b, double margin) {  // isSFCInBoundary, :894-901
    for (int k = 0; k < 3; k++)
        if (!(b[k] > g.wmin[k] + margin - SFC_EPS_F && b[3 + k] < g.wmax[k] - margin + SFC_EPS_F)) return false;
    return true;
}

// expandSFCIncrementally, :1023-1093, one trial expansion after the other as the reference writes it
template <class W>
DQ_FN bool sfc_expand_in_sequence(const W &w, const SfcWorldView &g, const SfcParams &p, const float *init, int max_iter, float *out) {
    if (sfc_obstacle_in_box(w, g, init, p)) return false;
    float sfc[6], cand[6], upd[6];
    int axis_cand[6] = {0, 1, 2, 3, 4, 5}, n_cand = 6;  // -x -y -z +x +y +z
    int axis_iter[6] = {0, 0, 0, 0, 0, 0};
    int i = -1;
    for (int k = 0; k < 6; k++) sfc[k] = init[k];
    while (n_cand > 0) {
        for (int k = 0; k < 6; k++) { cand[k] = sfc[k]; upd[k] = sfc[k]; }
        while (sfc_in_boundary(g, upd, 0) && !sfc_obstacle_in_box(w, g, upd, p)) {
            i++;
            if (i >= n_cand) i = 0;
            // axis_cand[i] without dynamic indexing of a local array
            int axis = 0;
            for (int t = 0; t < 6; t++) axis = t == i ? axis_cand[t] : axis;
            for (int k = 0; k < 6; k++) { sfc[k] = cand[k]; upd[k] = cand[k]; }
            for (int k = 0; k < 3; k++) {
                if (axis == k) {  // -k: the new slab is [min - res, min]
                    upd[3 + k] = cand[k];
                    cand[k] = (float)(cand[k] - g.res);
                    upd[k] = cand[k];
                }
                if (axis == 3 + k) {  // +k
                    upd[k] = cand[3 + k];
                    cand[3 + k] = (float)(cand[3 + k] + g.res);
                    upd[3 + k] = cand[3 + k];
                }
            }
            bool over = false;
            for (int t = 0; t < 6; t++)
                if (t == axis) { axis_iter[t]++; over = axis_iter[t] > max_iter; }
            if (over) break;
        }
        for (int t = 0; t < 5; t++)  // axis_cand.erase(begin + i)
            if (t >= i && t + 1 < n_cand) axis_cand[t] = axis_cand[t + 1];
        n_cand--;
        if (i > 0) i--; else i = n_cand - 1;
    }
    const double delta = p.margin - ((int)(p.margin / g.res) * g.res);  // SFC margin compensation
    for (int k = 0; k < 3; k++) {
        if (sfc[k] > g.wmin[k] + SFC_EPS_F) sfc[k] = (float)(sfc[k] - delta);
        if (sfc[3 + k] < g.wmax[k] - SFC_EPS_F) sfc[3 + k] = (float)(sfc[3 + k] + delta);
    }
    for (int k = 0; k < 6; k++) out[k] = sfc[k];
    return true;
}

// ---- the same routine, many trial expansions at a time ------------------------------------------------------------------------
// The reference's loop is a chain: whether trial t + 1 is even made depends on trial t's box test.  But almost every trial is
// accepted, and IF trials 1 .. t - 1 are accepted, trial t is known in advance: the directions take turns (axis_cand, i), each
// accepted trial moves one face by one cell, and a face after c moves is c float32 subtractions from where it started
// (W::chain, filled once).  So lane t of the wavefront evaluates trial t under that assumption -- limit of moves, boundary, and
// the box test where the summed-volume table settles it -- and the first lane that does not pass tells how many trials really
// were accepted; a trial the table cannot settle is tested point by point and the lanes behind it stay valid if it is free.
// ~65 dependent rounds become one round per direction that closes.  Same trials, same boxes, same floats as sfc_expand_in_sequence.
constexpr int SFC_CHAIN_CAP = 64;  // moves per direction the chain holds (max_iter + 2 entries are used)

struct SfcTurns {   // uniform state of the reference's loop
    int moves[6];   // accepted trials per direction (-x -y -z +x +y +z) = axis_iter of the directions still in play
    int cand[6];    // axis_cand
    int n_cand, i, max_iter;
};

// Trial tau >= 1 counted from the state (tau = 0: the test of the whole box that opens a round): the box `upd` that is tested;
// returns bit 0: the round ends here without a box test (limit of moves, or upd outside the boundary).  Branch-free in tau.
template <class W>
DQ_FN int sfc_turn(const W &w, const SfcWorldView &g, const SfcTurns &st, int tau, float *upd) {
    const bool is_trial = tau >= 1;
    const int q = is_trial ? tau - 1 : 0;
    const int r = (int)(((float)q + 0.5f) * w.rcp((float)st.n_cand)), sl = q - r * st.n_cand;  // q / n_cand, q % n_cand (exact: q < 128)
    int axis = 0, at = 0;   // the direction whose turn it is, and how often it has moved by then
    float face[6];
    for (int d = 0; d < 6; d++) {
        int pos = -1;
        for (int t = 0; t < 6; t++) pos = ((t < st.n_cand) & (st.cand[t] == d)) ? t : pos;  // uniform
        // turns go i + 1, i + 2, ... (mod n_cand): direction d is the o-th of a round
        int o = pos - st.i - 1;
        o = o < 0 ? o + st.n_cand : o;
        o = o < 0 ? o + st.n_cand : o;  // (i may be n_cand - 1 and pos 0: -n_cand <= pos - i - 1)
        const bool live = pos >= 0;
        const int done = st.moves[d] + ((is_trial & live) ? r + (o < sl ? 1 : 0) : 0);
        const bool mine = is_trial & live & (o == sl);
        axis = mine ? d : axis;
        at = mine ? done : at;
        face[d] = w.chain(d, done);
    }
    const float moved = w.chain(axis, at + 1);
    for (int d = 0; d < 6; d++) upd[d] = face[d];
    for (int k = 0; k < 3; k++) {
        const bool lo = is_trial & (axis == k), hi = is_trial & (axis == 3 + k);
        upd[3 + k] = lo ? face[k] : (hi ? moved : upd[3 + k]);  // -k: the new slab is [min - res, min]; +k: [max, max + res]
        upd[k] = lo ? moved : (hi ? face[3 + k] : upd[k]);
    }
    bool inb = true;
    for (int k = 0; k < 3; k++) inb &= (upd[k] > g.wmin[k] + 0.0 - SFC_EPS_F) & (upd[3 + k] < g.wmax[k] - 0.0 + SFC_EPS_F);  // sfc_in_boundary, margin 0
    const bool over = is_trial & (at + 1 > st.max_iter);
    return (int)(over | !inb);
}

template <class W>
DQ_FN bool sfc_expand(const W &w, const SfcWorldView &g, const SfcParams &p, const float *init, float *out) {
    const double mx = 2 * p.grid_resolution > p.agent_max_vel * p.dt ? 2 * p.grid_resolution : p.agent_max_vel * p.dt;
    const int max_iter = (int)round(mx / g.res) + 1;
    if (!p.svt || max_iter + 2 > SFC_CHAIN_CAP) return sfc_expand_in_sequence(w, g, p, init, max_iter, out);  // uniform
    if (sfc_obstacle_in_box(w, g, init, p)) return false;
    w.chain_fill(init, g.res, max_iter + 2);
    SfcTurns st;
    for (int d = 0; d < 6; d++) { st.moves[d] = 0; st.cand[d] = d; }
    st.n_cand = 6; st.i = -1; st.max_iter = max_iter;
    bool whole = true;  // the next lane 0 is the test of the whole box that opens a round
    while (st.n_cand > 0) {
        const int tau0 = whole ? 0 : 1;
        uint64_t ends = 0, open = 0;  // per lane: the round ends there for sure / the table does not settle the box
        w.lanes([&](int lane) {
            float upd[6];
            const int hard = sfc_turn(w, g, st, tau0 + lane, upd);
            const int count = sfc_box_count(g, p, upd);
            return hard | ((count != 0) << 1);
        }, ends, open);
        open &= ~ends;
        // the first lane that is not simply accepted
        int f = 0, ended = 0;
        for (;;) {
            const uint64_t rest = f < 64 ? ((ends | open) >> f) << f : 0;
            if (rest == 0) { f = 64; break; }
            f = w.uniform(sfc_ctz64(rest));
            if ((ends >> f) & 1) { ended = 1; break; }
            float upd[6];
            (void)sfc_turn(w, g, st, tau0 + f, upd);  // uniform: every lane forms lane f's box
            for (int k = 0; k < 6; k++) upd[k] = w.uniform_f(upd[k]);
            if (w.uniform((int)sfc_obstacle_in_box(w, g, upd, p, false)) != 0) { ended = 1; break; }
            f++;
        }
        // lanes [0, f) passed: f - (whole ? 1 : 0) trials accepted (lane 0 of an opening pass is not a trial)
        const int taken = f - (whole ? 1 : 0) < 0 ? 0 : f - (whole ? 1 : 0);
        const int tau_end = tau0 + f;  // the trial (or the opening test, 0) at which the round ended, if it did
        {
            const int r = taken / st.n_cand, sl = taken % st.n_cand;
            for (int d = 0; d < 6; d++) {
                int pos = -1;
                for (int t = 0; t < 6; t++) pos = ((t < st.n_cand) & (st.cand[t] == d)) ? t : pos;
                int o = pos - st.i - 1;
                o = o < 0 ? o + st.n_cand : o;
                o = o < 0 ? o + st.n_cand : o;
                st.moves[d] += pos >= 0 ? r + (o < sl ? 1 : 0) : 0;
            }
            const int turns = ended ? tau_end : taken;  // i++ happened for the trial that ended the round, too
            if (turns > 0) st.i = (st.i + turns) % st.n_cand;
        }
        if (ended) {
            for (int t = 0; t < 5; t++)  // axis_cand.erase(begin + i)
                if (t >= st.i && t + 1 < st.n_cand) st.cand[t] = st.cand[t + 1];
            st.n_cand--;
            if (st.i > 0) st.i--; else st.i = st.n_cand - 1;
            whole = true;
        } else {
            whole = false;
        }
    }
    float sfc[6];
    for (int d = 0; d < 6; d++) sfc[d] = w.uniform_f(w.chain(d, st.moves[d]));
    const double delta = p.margin - ((int)(p.margin / g.res) * g.res);  // SFC margin compensation
    for (int k = 0; k < 3; k++) {
        if (sfc[k] > g.wmin[k] + SFC_EPS_F) sfc[k] = (float)(sfc[k] - delta);
        if (sfc[3 + k] < g.wmax[k] - SFC_EPS_F) sfc[3 + k] = (float)(sfc[3 + k] + delta);
    }
    for (int k = 0; k < 6; k++) out[k] = sfc[k];
    return true;
}

DQ_FN bool sfc_point_in_box(const float *b, const float *pt) {  // Box::isPointInBox, :109-116
    for (int k = 0; k < 3; k++)
        if (!(pt[k] > b[k] - SFC_EPS_F && pt[k] < b[3 + k] + SFC_EPS_F)) return false;
    return true;
}

DQ_FN bool sfc_superset_of_hull(const float *b, const float *pts, int n) {  // Box::isSuperSetOfConvexHull, :170-185
    for (int k = 0; k < 3; k++) {
        float mn = pts[k], mxv = pts[k];
        for (int q = 1; q < n; q++) {
            mn = pts[3 * q + k] < mn ? pts[3 * q + k] : mn;
            mxv = pts[3 * q + k] > mxv ? pts[3 * q + k] : mxv;
        }
        if (mn < b[k] - SFC_EPS_F || mxv > b[3 + k] + SFC_EPS_F) return false;
    }
    return true;
}

// expandSFCFromConvexHull, both overloads (:781-857); prev == nullptr: the first
template <class W>
DQ_FN bool sfc_from_hull(const W &w, const SfcWorldView &g, const SfcParams &p, const float *pts, int n_pts, const float *prev,
                         float *out) {
    float b[6];
    for (int k = 0; k < 3; k++) b[k] = b[3 + k] = pts[k];
    for (int q = 0; q < n_pts; q++)
        for (int k = 0; k < 3; k++) {
            b[k] = pts[3 * q + k] < b[k] ? pts[3 * q + k] : b[k];
            b[3 + k] = pts[3 * q + k] > b[3 + k] ? pts[3 * q + k] : b[3 + k];
        }
    if (!prev) {
        for (int k = 0; k < 3; k++) {
            b[k] = (float)(round(b[k] / g.res) * g.res);
            b[3 + k] = (float)(round(b[3 + k] / g.res) * g.res);
        }
        bool ok = sfc_expand(w, g, p, b, out);
        if (ok && !sfc_superset_of_hull(out, pts, n_pts)) ok = false;
        return ok;
    }
    for (int k = 0; k < 3; k++) {
        b[k] = (float)(floor(b[k] / g.res) * g.res);
        b[3 + k] = (float)(ceil(b[3 + k] / g.res) * g.res);
    }
    if (!(sfc_point_in_box(prev, b) && sfc_point_in_box(prev, b + 3))) {  // not sfc_prev.include(sfc_initial)
        for (int k = 0; k < 3; k++) {
            const float lo = prev[k] > b[k] ? prev[k] : b[k], hi = prev[3 + k] < b[3 + k] ? prev[3 + k] : b[3 + k];
            b[k] = (float)(ceil((lo - SFC_EPS_F) / g.res) * g.res);
            b[3 + k] = (float)(floor((hi + SFC_EPS_F) / g.res) * g.res);
        }
    }
    return sfc_expand(w, g, p, b, out);
}

// initializeSFC (:435-452): M copies of the box grown from the cell of the agent's position.  sfcs: [M][6] in
// global memory, written by W::store6 (one lane).  Returns false where the reference throws.
template <class W>
DQ_FN bool sfc_initialize(const W &w, const SfcWorldView &g, const SfcParams &p, const float *pos, int M, float *sfcs) {
    float b[6], e[6];
    for (int k = 0; k < 3; k++) {
        b[k] = (float)(floor(pos[k] / g.res) * g.res);
        b[3 + k] = (float)(ceil(pos[k] / g.res) * g.res);
    }
    if (!sfc_expand(w, g, p, b, e)) return false;
    for (int m = 0; m < M; m++) w.store6(sfcs + 6 * m, e);
    return true;
}

// Box::isSuperSetOfConvexHull without branches (the lanes of a wavefront ask about different boxes)
DQ_FN bool sfc_superset_of_hull_lane(const float *b, const float *pts, int n) {
    bool ok = true;
    for (int k = 0; k < 3; k++) {
        float mn = pts[k], mxv = pts[k];
        for (int q = 1; q < n; q++) {
            mn = pts[3 * q + k] < mn ? pts[3 * q + k] : mn;
            mxv = pts[3 * q + k] > mxv ? pts[3 * q + k] : mxv;
        }
        ok &= !((mn < b[k] - SFC_EPS_F) | (mxv > b[3 + k] + SFC_EPS_F));
    }
    return ok;
}

constexpr int SFC_BOXES_CAP = 32;  // segments whose boxes W keeps at hand (W::boxes_load); more: the boxes are walked in memory

// constructSFCFromInitialTraj (:502-536).  initial_traj [M][n1][3]; sfcs [M][6] in/out.
// Returns 0: greedy box (hull incl. waypoint), 1: box inside the previous one, 2: previous box reused.
template <class W>
DQ_FN int sfc_construct(const W &w, const SfcWorldView &g, const SfcParams &p, const float *initial_traj, int M, int n1,
                        const float *goal, const float *waypoint, float *sfcs) {
    float cur[6], nxt[6];
    const bool at_hand = M <= SFC_BOXES_CAP && M >= 2;
    if (at_hand) {  // uniform
        // The reference shifts the boxes by one segment and then lets box m take over box m + 1 where that one holds segment m of
        // the initial trajectory -- two loops over M of load, test, store.  No iteration reads what another wrote (iteration m
        // writes box m and reads box m + 1 of the SHIFTED list), so with the old boxes b at hand:
        //     box m = holds(b[m + 2], segment m) ? b[m + 2] : b[m + 1]   (m <= M - 3),   box M - 2 = b[M - 1],
        // every test on a lane of its own, one trip to memory in and one out.
        w.boxes_load(sfcs, M);
        uint64_t takes_next = 0, unused = 0;
        w.lanes([&](int lane) {
            const int m = lane < M - 2 ? lane : (M > 2 ? M - 3 : 0);
            float b[6];
            w.box(m + 2 < M ? m + 2 : M - 1, b);
            return (int)(sfc_superset_of_hull_lane(b, initial_traj + (size_t)m * n1 * 3, n1) & (lane < M - 2));
        }, takes_next, unused);
        w.box(M - 1, cur);
        for (int k = 0; k < 6; k++) cur[k] = w.uniform_f(cur[k]);
        for (int m = 0; m < M - 1; m++) {
            const bool next = m < M - 2 && ((takes_next >> m) & 1);  // uniform
            w.box(next ? m + 2 : m + 1, nxt);
            w.box_keep(m, nxt);  // (box m is not read again: m + 1 and m + 2 are what the later iterations read)
        }
    } else {
        // W::fence() orders the global accesses of all threads that work on this agent (GPU: several wavefronts): every thread has
        // read box m + 1 before any thread overwrites it in the next iteration
        for (int m = 0; m < M - 1; m++) {  // shift, then the minor refinement reads the shifted boxes
            w.load6(sfcs + 6 * (m + 1), nxt);
            w.fence();
            w.store6(sfcs + 6 * m, nxt);
        }
        w.fence();
        for (int m = 0; m < M - 2; m++) {
            w.load6(sfcs + 6 * (m + 1), nxt);
            w.fence();
            if (sfc_superset_of_hull(nxt, initial_traj + (size_t)m * n1 * 3, n1)) {
                w.store6(sfcs + 6 * m, nxt);
                w.fence();
            }
        }
        w.load6(sfcs + 6 * (M - 1), cur);
        w.fence();
    }
    float hull[9];
    for (int k = 0; k < 3; k++) {
        hull[k] = initial_traj[((size_t)(M - 1) * n1 + (n1 - 1)) * 3 + k];  // lastPoint
        hull[3 + k] = goal[k];
        hull[6 + k] = waypoint[k];
    }
    float upd[6];
    int code = 0;
    if (!sfc_from_hull(w, g, p, hull, 3, (const float *)nullptr, upd)) {
        code = 1;
        if (!sfc_from_hull(w, g, p, hull, 2, cur, upd)) {
            code = 2;
            for (int k = 0; k < 6; k++) upd[k] = cur[k];
        }
    }
    if (at_hand) {  // uniform
        w.box_keep(M - 1, upd);
        w.boxes_store(sfcs, M);
    } else {
        w.store6(sfcs + 6 * (M - 1), upd);
    }
    return code;
}

}  // namespace dlscqp
