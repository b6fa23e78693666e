// gi_core.hpp -- per-agent goal LP + trajectory QP, one agent per wavefront.
//
// Replaces GoalOptimizer::solve (goal_optimizer.cpp:7-198) and TrajOptimizer::solve
// (traj_optimizer.cpp:18-165, model at :225-527) of the reference.  The equalities are eliminated
// and the Hessian whitened on the host (qp_tables.cpp), which leaves, per agent, a projection
// problem  min 1/2 |w|^2  s.t. sparse half-spaces in control-point space (x = x0 + N_T w), solved here
// with a Goldfarb-Idnani dual active-set method in RANGE-SPACE form -- the kernel never sees w or N_T:
//   * every constraint row the wave will ever look at is decoded ONCE into registers: lane l owns
//     LSC rows l, l+64, ... (coalesced SoA loads) and structured slots l, l+64, ...; a scan is then
//     pure LDS reads of control points + FMAs, followed by one arg-max butterfly;
//   * a row touches at most three control-point coordinates (and one slack variable), so everything the
//     method needs from the whitened space is an entry of the Gram table G_T = N_T N_T' (P x P per axis,
//     one table per number of terminal segments, built on the host, copied to LDS when an agent first
//     iterates): a_i' G a_j is at most nine table entries, the step direction dx = G (a - A' r) is one
//     table row per touched control point.  There is no orthogonal factor: work per iteration is
//     O(active rows), not O(r^2);
//   * lane i keeps active row i (its sparse entries, its dual) in registers; S = R^-1 (packed upper
//     triangle; R'R = A G A' is the Gram matrix of the active rows) lives in LDS.  Keeping the INVERSE
//     turns both triangular solves into lane-parallel products, appending a row into writing one column
//     (-r / alpha, 1 / alpha), and deleting a row into a column-rotation sweep whose angles come from one
//     prefix sum of row l of S.  ~60 % of the agents never get that far: their unconstrained optimum is feasible.
//
// Compiled with -ffp-contract=off: the float32 expressions mirror octomap::point3d arithmetic of
// the reference and must not be fused; the hot FP64 loops call fma() explicitly.
//
// The code is SPMD over a "wave" interface W: Reg<T> is one value per lane, w.each(f) runs f for
// every lane, w.bcast / w.sum2 / w.max / w.min / w.any / w.first_lane / w.shift_down are the
// cross-lane operations.  On the GPU Reg<T> is a VGPR and each() is the lane itself; the very same
// source also compiles as a lane-serial host build (tests/emul: Reg<T> = T[64]) that the CPU-only
// unit tests use to check the algorithm.  That build is test infrastructure and is never loaded by
// the product library.  Rule that keeps both builds equivalent: inside each() only Regs, LDS
// elements owned by the lane and lambda-local temporaries are written, and nothing written inside
// an each() is read from another lane before it ends.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/dlscqp.h"
#include "qp_types.hpp"

#ifndef DQ_FN
#define DQ_FN inline
#endif
// DQ_KEEP(v): the value must exist here, unconditionally.  On the GPU it stops the compiler from sinking an
// LDS load into the branch of a later select (which would serialise every load behind its own s_waitcnt).
#ifndef DQ_KEEP
#define DQ_KEEP(v) ((void)0)
#endif
// DQ_SCORE_MAX(a, b): the larger of two tagged pivot scores (never NaN) -- on the GPU one v_max_f64, without the
// canonicalisation of its operands that DQ_SCORE_MAX() asks for
#ifndef DQ_SCORE_MAX
#define DQ_SCORE_MAX(a, b) ((a) > (b) ? (a) : (b))
#endif
// DQ_RSQRT(x): 1 / sqrt(x) for a pivot scaling (ranks rows, never enters a result): v_rsq_f64 and one Newton step on the GPU
#ifndef DQ_RSQRT
#define DQ_RSQRT(x) (1.0 / sqrt(x))
#endif
// DQ_WAVE_ALL(c): c holds for every lane of the wavefront (wave-uniform, a real branch can test it); lane-serial host build: c itself --
// only where both sides of the branch compute the same thing for a lane for which c holds
#ifndef DQ_WAVE_ALL
#define DQ_WAVE_ALL(c) (c)
#endif
// DQ_SETPRIO(n): the wavefront's issue priority, 0 .. 3 (s_setprio; nothing on the host)
#ifndef DQ_SETPRIO
#define DQ_SETPRIO(n) ((void)0)
#endif
#ifndef DQ_PRIO_STEPS  // 1: an iterating agent's wavefront raises its priority to 1, 2, 3 at its DQ_PRIO_AT1 / 2 / 3-th pricing pass (0: never)
#define DQ_PRIO_STEPS 1
#endif
#ifndef DQ_PRIO_AT1
#define DQ_PRIO_AT1 3
#define DQ_PRIO_AT2 8
#define DQ_PRIO_AT3 16
#endif
// DQ_MARK(text): a comment in the generated assembly (tests/divergence_check.py recognises the exit stores by it)
#ifndef DQ_MARK
#define DQ_MARK(text) ((void)0)
#endif

namespace dlscqp {

// KU: LSC rows cached in registers per lane (64 KU rows; the rest is read from HBM on every scan)
// SU: structured slots cached per lane (64 SU slots; the rest from HBM).  Template parameters of
// plan_agent: the register budget of the wave (256 VGPRs at 2 waves/SIMD) is shared with the unrolled loops.

struct MemLayout {
    int r_max, nx_max, ldg, nflag;
    // COLUMN RECORDS of the active rows, one per active row k: [ column k of S = R^-1, r_max entries, zeros below the diagonal |
    // column k of Z = G A' S, nx_max entries ] -- cs doubles apart (odd: lane k reading its own column and lane p reading one
    // column are both conflict-free).  The first qz records live in the working set, records qz .. r_max - 1 in the agent's SPILL
    // record in HBM (plan_agent's `spill` argument: 32 dump doubles, then the records); qz == r_max: no spill needed.
    int oC, cs, qz;
    int oX, oDx, oBlo, oBhi;  // in doubles (oDx < 0: the instantiation keeps the step direction in registers, DXREG)
    int oU;                   // nx_max doubles: a - A' r in the variables' own space (plan_agent: whitened_length)
    int oKiw;                 // LEAN instantiation: 3 nxc doubles, pivot scalings of the bound / velocity / acceleration rows by coordinate (else -1)
    int oMail;                // 8 doubles: what the planning wavefront and its helper wavefronts tell each other (dlscqp.hip)
    int oDump;                // 32 dump slots (lane & 31) for the lanes that have nothing to store; then two doubles that stay 0.0
    int oFlag;                // byte offset of the skip flags: two bytes per structured slot beyond the slot cache, then (oFlagRows)
    int oFlagRows;            // ONE BIT per LSC row that is priced from HBM (beyond the register cache and the LDS rows below)
    int kmax;                 // most LSC rows of one agent the layout was sized for (dlscqp_device_batch.max_rows_per_agent)
    int cold_cap, oCold;      // rows [64 KU, 64 KU + cold_cap) of an agent are kept in the working set, decoded, 32 bytes each
                              // (plan_agent: "rows in LDS"); oCold in doubles
    int spill_doubles;        // doubles of one agent's spill record (0: none needed)
    int bytes;
};

// The Gram tables stay in HBM / L2 for every shape (round 5): an iteration reads one to three ROWS of G_T (coalesced, the row of
// every control point the new row touches), everything else it needs of the whitened space is in the column records.
// lds_budget: bytes of LDS the launch lets one agent have (0: the working set proper, no rows beyond the register cache in it).
// qz_want: column records in the working set (0 or >= r_max: all of them; else rounded down to a multiple of 8, at least 8: the
// loops over the records go in blocks of 8 and of 4 that must not straddle the boundary); the rest is spilled.
DQ_FN MemLayout make_layout(const KParams &kp, int max_slack_obs, int kmax, int KU, int SU, bool by_kind, int lds_budget = 0, int qz_want = 0) {
    MemLayout L;
    const int nslk = kp.M * max_slack_obs;
    L.r_max = kp.dim * kp.f + nslk;  // the active rows are independent in the whitened space: never more than its dimension
    L.nx_max = kp.dim * kp.P + nslk;
    L.ldg = kp.P | 1;  // odd leading dimension of the Gram table (as the host builds it)
    const int cold_rows = kmax > 64 * KU ? kmax - 64 * KU : 0;
    const int gen_slots = by_kind ? kp.n_slots - kp.n_struct : kp.n_slots;  // the slots the generic code scans
    const int cold_slots = gen_slots > 64 * SU ? gen_slots - 64 * SU : 0;
    const bool dxreg = by_kind && nslk == 0 && kp.P <= 64;  // plan_agent: DXREG
    L.kmax = kmax;
    L.cs = (L.r_max + L.nx_max) | 1;
    L.qz = (qz_want <= 0 || qz_want >= L.r_max) ? L.r_max : ((qz_want & ~7) < 8 ? 8 : (qz_want & ~7));
    if (L.qz >= L.r_max) L.qz = L.r_max;
    L.spill_doubles = L.qz < L.r_max ? 32 + (L.r_max - L.qz) * L.cs : 0;
    int o = 0;
    L.oDump = o; o += 32 + 2;
    L.oX = o; o += L.nx_max;
    L.oDx = dxreg ? -1 : o; o += dxreg ? 0 : L.nx_max;
    L.oC = o; o += L.qz * L.cs;
    L.oBlo = o; o += kp.nb;
    L.oBhi = o; o += kp.nb;
    L.oMail = o; o += 8;
    L.oU = o; o += L.nx_max;
    // (plan_agent: LEAN -- the instantiation whose rows and slots are all in registers keeps the coordinates' pivot scalings here)
    const bool lean = by_kind && nslk == 0 && KU == 2 && SU == 0;
    L.oKiw = lean ? o : -1; o += lean ? 3 * kp.dim * kp.P : 0;
    L.oCold = o;
    L.cold_cap = 0;
    if (lds_budget > 0 && cold_rows > 0) {
        int cap = (lds_budget - (o * 8 + 2 * cold_slots + cold_rows / 8 + 48)) / 32;  // a row in the working set: 32 bytes (and no flag bit)
        cap = cap > cold_rows ? cold_rows : cap;
        cap &= ~1;
        L.cold_cap = cap >= 64 ? cap : 0;
        o += 4 * L.cold_cap;
    }
    L.nflag = 2 * cold_slots + (cold_rows - L.cold_cap + 7) / 8;
    L.oFlag = o * 8;
    L.oFlagRows = L.oFlag + 2 * cold_slots;
    L.bytes = (L.oFlag + L.nflag + 1 + 15) & ~15;  // + one flag byte as the dump of the flag array
    return L;
}

// Instantiations of plan_agent: id, RC, DIMS, FS, MS, KU, SU, SLK.  Static shapes for the reference's launch
// files (3-D M=5: f=13; 2-D / 3-D M=10: f=28) with register caches sized to the row counts, with and without
// slack (dynamic) obstacles; runtime-shape instantiations for everything else.  One list for the HIP launcher
// and for the lane-serial test build.
#ifndef DQ_C3_KU  // register cache of the 2-D, M = 10 instantiation: it runs one wavefront per SIMD (dlscqp.hip: DQ_KERNEL_ATTR) and has the
#define DQ_C3_KU 12  // whole register file -- 768 rows of a crowd's ~1 035 in (accumulation) registers (3 / 8 / 12: flown crowd 0.311 / 0.306 / 0.292 ms)
#endif
#ifndef DQ_G2_KU  // register caches of the two-register runtime-shape instantiation (also the one with HBM working-set variants)
#define DQ_G2_KU 2
#define DQ_G2_SU 6
#endif
#define DQ_INSTANCES(X)          \
    X(0, 1, 3, 13, 5, 2, 0, 0)   \
    X(1, 1, 3, 13, 5, 4, 4, 0)   \
    X(2, 1, 3, 13, 5, 4, 4, -1)  \
    X(3, 1, 2, 28, 10, DQ_C3_KU, 3, 0)  \
    X(4, 2, 3, 28, 10, 2, 4, 0)  \
    X(5, 2, 3, 28, 10, 2, 4, -1) \
    X(6, 1, 0, 0, 0, 2, 6, -1)   \
    X(7, 2, 0, 0, 0, DQ_G2_KU, DQ_G2_SU, -1)
constexpr int DQ_N_INSTANCES = 8;

struct InstanceChoice {
    int id, KU, SU;  // id < 0: unsupported (reduced dimension above 128)
    bool by_kind;    // static shapes: bound / velocity / acceleration rows by coordinate, not from the slot table
};

// Which instantiation plans a batch (r_max = dim f + M max_slack_obs, kmax = most LSC rows of one agent).
// register-cache sizes of an instantiation, from the table above (the one place they are written down)
inline InstanceChoice instance_choice(int id) {
    switch (id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID: return {ID, KU, SU, (DIMS) > 0};
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    default: return {-1, 0, 0, false};
    }
}

inline InstanceChoice select_instance(const KParams &kp, int max_slack_obs, int kmax, bool force_generic) {
    const int r_max = kp.dim * kp.f + kp.M * max_slack_obs;
    const bool ns = max_slack_obs == 0;
    if (!force_generic) {
        if (kp.dim == 3 && kp.f == 13 && kp.M == 5 && r_max <= 64) {
            const InstanceChoice all_cached = instance_choice(0);  // compiled without the loops over rows / slots beyond its caches
            if (ns && kmax <= 64 * all_cached.KU && kp.n_slots - kp.n_struct <= 64 * all_cached.SU) return all_cached;
            return instance_choice(ns ? 1 : 2);
        }
        if (kp.dim == 2 && kp.f == 28 && kp.M == 10 && ns) return instance_choice(3);
        if (kp.dim == 3 && kp.f == 28 && kp.M == 10 && r_max <= 128) return instance_choice(ns ? 4 : 5);
    }
    if (r_max <= 64) return instance_choice(6);
    if (r_max <= 128) return instance_choice(7);
    return {-1, 0, 0, false};
}

// octomath::Vector3 semantics: float32 products and sum, sqrt in double
DQ_FN double f32_norm3(float a, float b, float c) {
    float sq = a * a + b * b + c * c;
    return sqrt((double)sq);
}

// f32_norm3(a, b, c) < SP_EPSILON_FLOAT without the square root: the float32 sum of squares is below the smallest float32
// whose double square root reaches 1e-5 (0x2EDBE6FF = 1.0000000e-10; the square root is monotone and correctly rounded, so
// the two tests agree on every input -- tests/test_norm_threshold.py walks the floats around the threshold).  A double
// square root is ~40 instructions and the zero-normal test runs for every cached row of every agent, twice.
DQ_FN bool f32_norm3_below_eps(float a, float b, float c) {
    const float sq = a * a + b * b + c * c;
    uint32_t lim_bits = 0x2EDBE6FFu;
    float lim;
    memcpy(&lim, &lim_bits, 4);
    return sq < lim;
}

constexpr uint32_t KEY_SKIP = 0xFFFFFFFFu;  // cached LSC row that is not part of the model

// cond ? a : b on indices with mask arithmetic: the index of a store that a lane may have to send to the dump slot
// (written as ?: the compiler turned some of these into EXEC-masked branches around the store)
DQ_FN int blend_idx(bool cond, int a, int b) {
    const int m = -(int)cond;
    return b ^ ((a ^ b) & m);
}

// cond ? v : 0 by masking the bits (the same reason: a select of a loaded value against zero came out as an EXEC-masked move)
DQ_FN float keep_f32(bool cond, float v) {
    uint32_t b;
    memcpy(&b, &v, 4);
    b &= (uint32_t)(-(int32_t)cond);
    memcpy(&v, &b, 4);
    return v;
}
DQ_FN double keep_f64(bool cond, double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    b &= (uint64_t)(-(int64_t)cond);
    memcpy(&v, &b, 8);
    return v;
}

DQ_FN int ctz64(uint64_t m) { return __builtin_ctzll(m); }
DQ_FN int popc64(uint64_t m) { return __builtin_popcountll(m); }

// Pivot scores carry the code of their row in the TAG_BITS low bits of the mantissa (a relative perturbation of 2^-32):
// the arg-max over all rows of an agent is then a plain maximum -- one v_max_f64 per candidate row, one wave reduction,
// and the winner's code is read off the result.  code = unit * 8 + c: `unit` numbers the things a lane owns (a coordinate
// of the control polygon, an LSC row, a structured slot, a slack variable), c the candidate row of that unit.
constexpr int TAG_BITS = 20;
DQ_FN double tag_score(double sc, uint32_t code) {
    uint64_t b;
    memcpy(&b, &sc, 8);
    b = (b & ~(uint64_t)((1u << TAG_BITS) - 1u)) | (uint64_t)code;
    memcpy(&sc, &b, 8);
    return sc;
}
DQ_FN uint32_t score_code(double sc) {
    uint64_t b;
    memcpy(&b, &sc, 8);
    return (uint32_t)(b & (uint64_t)((1u << TAG_BITS) - 1u));
}
// a tagged score is "violated" when it is a positive NORMAL number: a score of exactly +0 becomes a denormal by its tag
constexpr double SCORE_MIN = 2.2250738585072014e-308;
constexpr double NEVER = 1e200;  // limit of a row that is not in the model (or is active): never violated (times a pivot scaling it stays finite:
                                 // an infinite score would turn into a NaN under its tag)

// the element `bytes` behind a (uniform) base: a 32-bit byte offset, so that a vector load can take the base from scalar registers
template <class T>
DQ_FN T at_offset(const T *base, uint32_t bytes) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)bytes);
}

// v[k] for a per-lane k without dynamic indexing of a register array (which would go through scratch memory)
template <class T>
DQ_FN T pick3(const T *v, int k) {
    const T v0 = v[0], v1 = v[1], v2 = v[2];  // flat selects: nested ?: came out as EXEC-masked branches
    T r = v2;
    r = k == 1 ? v1 : r;
    r = k == 0 ? v0 : r;
    return r;
}

// element e of a per-lane vector spread over RC registers
#define DQ_BC(arr, e) (RC == 1 ? w.bcast((arr)[0], (e)) : ((e) < 64 ? w.bcast((arr)[0], (e)) : w.bcast((arr)[RC - 1], (e) - 64)))
#define DQ_BCI(arr, e) DQ_BC(arr, e)
// element e of column t of a [RC][3] array of per-lane vectors
#define DQ_BC2(arr, t, e) (RC == 1 ? w.bcast((arr)[0][t], (e)) : ((e) < 64 ? w.bcast((arr)[0][t], (e)) : w.bcast((arr)[RC - 1][t], (e) - 64)))

// The rows of an agent that do not fit the register cache of the wavefront that plans it ("cold" rows): priced straight from
// HBM, CB blocks of 64 rows per trip -- first every row's record, then the loads that depend on it (control points, scaling,
// slack variable), then the arithmetic.  One lane's share of rows [jb, je) of the agent's K rows; returns the lane's best
// tagged pivot score (tag_score); `least` = min(least, the smallest scaled slack -score the lane saw).  Shared by the planning wavefront and the HELPER wavefronts of a workgroup that only
// price cold rows (dlscqp.hip: WPA > 1).
struct ColdRows {
    const float *nx, *ny, *nz, *px, *py, *pz;  // the agent's rows (RowsView + row_begin)
    const double *d;
    const uint32_t *key;
    const unsigned char *flag;  // skip flags in the working set, one BIT per row from Kc on (Kc here: the first row that is priced from HBM)
    const double *x;            // control points and slack variables
    const double *tNN, *sigma;
    int K, Kc, P, M, dim, nxc, nso, phi, unit0;
};
template <int CB>
DQ_FN double cold_rows_lane_best(const ColdRows &c, int l, int jb, int je, double bs, double &least) {
    constexpr int n1 = 6;
    for (int j0 = jb; j0 < je; j0 += 64 * CB) {  // uniform trip count
        uint32_t key[CB];
        float g[CB][6];
        double gd[CB];
        int cf0[CB];
#pragma unroll
        for (int u = 0; u < CB; u++) {
            const int j = j0 + 64 * u + l;
            const int jc = j < je ? j : je - 1;
            const uint32_t o4 = (uint32_t)jc * 4u, o8 = (uint32_t)jc * 8u;  // (32-bit byte offsets: at_offset)
            key[u] = at_offset(c.key, o4);
            g[u][0] = at_offset(c.nx, o4); g[u][1] = at_offset(c.ny, o4); g[u][2] = at_offset(c.nz, o4);
            g[u][3] = at_offset(c.px, o4); g[u][4] = at_offset(c.py, o4); g[u][5] = at_offset(c.pz, o4);
            gd[u] = at_offset(c.d, o8);
            cf0[u] = (c.flag[(jc - c.Kc) >> 3] >> ((jc - c.Kc) & 7)) & 1;
        }
#pragma unroll
        for (int u = 0; u < CB; u++) {
            DQ_KEEP(key[u]); DQ_KEEP(gd[u]); DQ_KEEP(cf0[u]);
#pragma unroll
            for (int t = 0; t < 6; t++) DQ_KEEP(g[u][t]);
        }
        double xv[CB][4], tn[CB], sg[CB];
        bool inm[CB], hsl[CB];
#pragma unroll
        for (int u = 0; u < CB; u++) {
            const int j = j0 + 64 * u + l;
            const int m = key[u] & 0xFF, i = (key[u] >> 8) & 0xF;
            inm[u] = (j < je) & (cf0[u] == 0) & !((m == 0) & (i < c.phi)) & !f32_norm3_below_eps(g[u][0], g[u][1], g[u][2]);
            const int p = inm[u] ? m * n1 + i : 0;
            const uint32_t sl = key[u] >> 16;
            hsl[u] = inm[u] & (sl != 0xFFFFu) & ((int)sl < c.nso);
            xv[u][0] = c.x[p]; xv[u][1] = c.x[c.P + p]; xv[u][2] = c.x[(c.dim == 3 ? 2 : 0) * c.P + p];
            xv[u][3] = c.x[c.nxc + (hsl[u] ? (int)sl * c.M + m : 0)];
            tn[u] = c.tNN[p];
            sg[u] = c.sigma[inm[u] ? m : 0];
        }
#pragma unroll
        for (int u = 0; u < CB; u++) { DQ_KEEP(xv[u][0]); DQ_KEEP(xv[u][1]); DQ_KEEP(xv[u][2]); DQ_KEEP(xv[u][3]); DQ_KEEP(tn[u]); DQ_KEEP(sg[u]); }
#pragma unroll
        for (int u = 0; u < CB; u++) {
            const int j = j0 + 64 * u + l;
            const float nx = g[u][0], ny = g[u][1], nz = g[u][2];
            double val = (double)nx * xv[u][0] + (double)ny * xv[u][1];
            double rhs = gd[u] + (double)nx * (double)g[u][3] + (double)ny * (double)g[u][4];
            const double val3 = val + (double)nz * xv[u][2], rhs3 = rhs + (double)nz * (double)g[u][5];
            val = c.dim == 3 ? val3 : val;
            rhs = c.dim == 3 ? rhs3 : rhs;
            val = hsl[u] ? val - xv[u][3] : val;
            const double v = (rhs - VIOL_TOL) - val;
            double wsq = ((double)nx * (double)nx + (double)ny * (double)ny + (c.dim == 3 ? (double)nz * (double)nz : 0.0)) * tn[u];
            wsq = hsl[u] ? wsq + sg[u] * sg[u] : wsq;
            double qv = v * DQ_RSQRT(wsq > 0 ? wsq : 1.0);
            DQ_KEEP(qv);
            const double sc = (inm[u] & (wsq > 0)) ? qv : -1.0;
            least = -sc < least ? -sc : least;
            bs = DQ_SCORE_MAX(bs, tag_score(sc, (uint32_t)(c.unit0 + (j < je ? j : je - 1)) << 3));
        }
    }
    return bs;
}
// rows [jb, je) of the cold rows [Kc, K) that wavefront `wave` of `nw` prices: contiguous shares of whole blocks of 64
DQ_FN void cold_rows_share(int Kc, int K, int wave, int nw, int &jb, int &je) {
    const int blocks = (K - Kc + 63) / 64, per = (blocks + nw - 1) / nw;
    jb = Kc + 64 * per * wave;
    je = Kc + 64 * per * (wave + 1);
    jb = jb < K ? jb : K;
    je = je < K ? je : K;
}

// The rows right behind the register cache, decoded once into the working set ("rows in LDS": right-hand side, pivot scaling, normal,
// key -- 32 bytes a row): one lane's share of rows [jb, je) of them (global row numbers; the first row in LDS is row Kc), two blocks
// of 64 a trip.  Shared by the planning wavefront and the helper wavefronts, like cold_rows_lane_best.
struct LdsRows {
    const double *rhs, *iw;
    const float *nx, *ny, *nz;
    const uint32_t *key;
    const double *x;  // control points and slack variables
    int Kc, P, M, dim, nxc, unit0;
};
template <bool HAS_SLACK>
DQ_FN double lds_rows_lane_best(const LdsRows &r, int l, int jb, int je, double bs, double &least) {
    constexpr int n1 = 6;
    for (int j0 = jb; j0 < je; j0 += 128) {  // uniform trip count
        double xv[2][4], rh[2], iwv[2];
        float nv[2][3];
        uint32_t kq[2];
        int jr[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = j0 + 64 * u + l;
            jr[u] = (j < je ? j : je - 1) - r.Kc;
            kq[u] = r.key[jr[u]];
            rh[u] = r.rhs[jr[u]]; iwv[u] = r.iw[jr[u]];
            nv[u][0] = r.nx[jr[u]]; nv[u][1] = r.ny[jr[u]]; nv[u][2] = r.nz[jr[u]];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            DQ_KEEP(kq[u]);
            const int m = kq[u] & 0xFF, p = m * n1 + (int)((kq[u] >> 8) & 0xF);
            const uint32_t sl = kq[u] >> 16;
            xv[u][0] = r.x[p]; xv[u][1] = r.x[r.P + p]; xv[u][2] = r.dim == 3 ? r.x[2 * r.P + p] : 0.0;
            if (HAS_SLACK) {  // (read by every lane through a valid index, then selected: no lane-guarded load)
                const bool has_slack = sl != 0xFFFFu;
                double xs = r.x[r.nxc + (has_slack ? (int)sl * r.M + m : 0)];
                DQ_KEEP(xs);
                xv[u][3] = has_slack ? xs : 0.0;
            } else {
                xv[u][3] = 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) { DQ_KEEP(xv[u][0]); DQ_KEEP(xv[u][1]); DQ_KEEP(xv[u][2]); DQ_KEEP(xv[u][3]); DQ_KEEP(rh[u]); DQ_KEEP(iwv[u]); }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            double val = (double)nv[u][0] * xv[u][0];
            val = fma((double)nv[u][1], xv[u][1], val);
            if (r.dim == 3) val = fma((double)nv[u][2], xv[u][2], val);
            if (HAS_SLACK) val -= xv[u][3];
            const double v = (rh[u] - VIOL_TOL) - val;
            // (a lane past the last row has priced the last row again: the same score under the same code)
            const double sc = v * iwv[u];
            least = -sc < least ? -sc : least;
            bs = DQ_SCORE_MAX(bs, tag_score(sc, (uint32_t)(r.unit0 + r.Kc + jr[u]) << 3));
        }
    }
    return bs;
}

// RC: registers per per-lane vector over the active rows (at most 64 RC of them; the whitened dimension r bounds their number).
// DIMS/FS/MS: compile-time world dimension, free degrees of freedom per axis and segment count (0 = read them from KParams).
// With static shapes the loops over control points and bound-table entries have constant trip counts and are fully unrolled.
//
// NO DIVERGENCE.  Every lane executes every instruction of plan_agent: there is no branch and no loop bound that
// depends on the lane.  Lanes that have no element to work on use a clamped (valid) index for their loads,
// mask the result with a select, and send their stores to a dump slot (MemLayout::oDump).
// This is a correctness rule, not a style: the solver keeps ~100 VGPRs of per-lane state and wave-uniform
// values (LDS addresses, cached rows) alive across the whole iteration, and when the register allocator
// splits or spills such a live range INSIDE a lane-divergent region, the copy runs under the region's EXEC
// mask and the inactive lanes come back as garbage after the region (seen on gfx950 / ROCm 7.2: lanes
// >= r - 64 of the two-register generic instantiation, depending on unrelated source changes).  Branches on
// wave-uniform conditions do not touch EXEC and are fine.
// profiling builds only (-DDQ_PHASE_PROFILE, tests/tools/phase_cost.py): DLSCQP_DEBUG_EXIT_PHASE=k in the environment ends every
// agent's solve after phase k, so that the difference of two launches' times is what a phase costs in throughput (not in
// one wavefront's latency)
#ifdef DQ_PHASE_PROFILE
#define DQ_PHASE_EXIT(k)                                          \
    if (kp.dbg_exit_phase == (k)) {                               \
        exit_stores(DLSCQP_OK, true, 0, 0, 0, 0, false, 0.0);     \
        return;                                                   \
    }
#else
#define DQ_PHASE_EXIT(k)
#endif
#ifdef DLSCQP_TRACE
#define DQ_TRACE(tag, a, b, c, d) w.trace(tag, a, b, c, d)  // debug builds only (tests/trace_gpu.py)
#else
#define DQ_TRACE(tag, a, b, c, d)
#endif

// SLK: compile-time number of slack obstacles of the batch (0), or -1 = read it from the arguments.  With
// DIMS > 0 and SLK = 0 the LDS layout up to the column records (dump, x, stride and place of the records) is made of
// constants: LDS addresses become "lane part + immediate", no address arithmetic in the loops.
// spill: the agent's spill record in HBM (MemLayout::spill_doubles doubles), or null where the layout keeps every column
// record in the working set (lay.qz == lay.r_max).
template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK, class W>
DQ_FN void plan_agent(const W &w, const KParams &kp, const MemLayout &lay, unsigned char *mem, double *spill,
                      const dlscqp_agent *agp, const RowsView &rows, int row_begin, int K_rows,
                      const float *sfc, int max_slack_obs, int do_goal, int do_traj,
                      float *out_goal, int *out_goal_status, float *out_ctrl, double *out_x,
                      double *out_cost, double *out_slack_cost, int *out_status, int *out_iters, int *iters_done = nullptr,
                      double *out_conflict = nullptr) {
    // out_conflict (optional, device memory: dlscqp_device_batch.out_conflict, 4 + 2 (r_max + 1) doubles): the certificate of an
    // INFEASIBLE verdict -- [0] members n (0: none), [1] how far the blocked row is violated (metres), then n pairs (row id,
    // multiplier): the active rows with -rv, the blocked row with 1 (conflict_id() below; include/dlscqp.h).  While it is asked for,
    // the early proof by weak duality is off: the active-set method runs until no step is possible, which is what the certificate says.
    // iters_done (optional, the caller's own variable -- not device memory): the number of iterations of this solve
    static_assert(RC == 1 || RC == 2, "r <= 128");
    typedef typename W::template Reg<double> RegD;
    typedef typename W::template Reg<float> RegF;
    typedef typename W::template Reg<uint32_t> RegU;
    typedef typename W::template Reg<int> RegI;

    constexpr int CB = 4;  // rows beyond the register cache are read in trips of CB blocks of 64
    // select_instance() hands the KU = 2 instantiation only batches whose rows and slots all fit the register caches
    // (kmax <= 128, n_slots <= 256): it is compiled without the loops over rows and slots beyond them
    constexpr bool ALL_CACHED = KU == 2 && SU == 0;
    // ... and, knowing all of an agent's rows, it asks first whether any of them is violated at all (the screen, below)
    constexpr bool SCREEN = ALL_CACHED && DIMS > 0 && SLK == 0;
    // LEAN (the same instantiation: S64): compiled for THREE wavefronts per SIMD -- 168 vector registers.  What the pricing pass alone
    // reads does not sit in registers across the iteration: the coordinates' pivot scalings live in LDS (lay.oKiw), the velocity and
    // acceleration limits are one uniform number each plus a bit per coordinate ("this row is out of the pass": lim_off), row codes
    // are computed from the lane number, and the float32 normals go through DQ_KEEP at the head of a pass, so that their
    // conversions to double are made there instead of once in front of the loop (twelve registers).  Same arithmetic, same bits.
    constexpr bool LEAN = SCREEN;
    constexpr int SUA = SU > 0 ? SU : 1;  // register arrays of the slot cache (never empty)
    // BYK (static shapes): the rows of one KIND that every agent has -- variable bounds / box faces, velocity and
    // acceleration limits (traj_optimizer.cpp:237-270, 385-410, 453-487) -- are not decoded from the slot table: lane l owns
    // COORDINATES l, l + 64 of the control polygon (axis k, control point p: index k P + p) with their six rows
    //   lo <= x_j <= hi,   |x_{j+1} - x_j| <= hv,   |x_{j+2} - 2 x_{j+1} + x_j| <= ha,
    // the limits in registers (NEVER where the model has no such row, or while the row is active).  Only the rows of the
    // communication range (kinds COMM, WAYPOINT), where there are any, go through the generic slot code.
    constexpr bool BYK = DIMS > 0;
    const int dim = DIMS > 0 ? DIMS : kp.dim, f = FS > 0 ? FS : kp.f;
    const int M = MS > 0 ? MS : kp.M;
    constexpr int n1 = 6;  // n = 5 is the only degree the reference supports (traj_optimizer.cpp:209)
    const int P = M * n1;
    const int n = n1 - 1;
    const int nxc = dim * P;  // control-point variables ("coordinates")
    const int slot0 = BYK ? kp.n_struct : 0;  // the slots the generic code handles
    const int n_slots_all = kp.n_slots - slot0;
    const int K = ALL_CACHED && K_rows > 64 * KU ? 64 * KU : K_rows;
    // The caller's max_rows_per_agent picked this instantiation and sized the flag array: an agent with more rows than
    // declared (or, in the all-cached instantiation, more rows or slots than the register caches hold) is not planned
    // with some of its rows dropped -- it fails with DLSCQP_NUMERICAL.
    const bool bad_shape = K_rows < 0 || K_rows > lay.kmax || (ALL_CACHED && (K_rows > 64 * KU || n_slots_all > 64 * SU)) ||
                           (nxc + K_rows + n_slots_all + M * max_slack_obs) >= (1 << (TAG_BITS - 3));
#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(3);
#endif
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
    // the corridor box of the last segment (the goal LP's rows, goal_optimizer.cpp:158-173): read with the record, all six
    // numbers at once (read where they are used, every one of them was a round trip of its own)
    float lastbox[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (kp.use_sfc && sfc && do_goal) {
        const float *lb = sfc + ((MS > 0 ? MS : kp.M) - 1) * 6;
#pragma unroll
        for (int t = 0; t < 6; t++) lastbox[t] = lb[t];
    }
    int nso_rt = agp->n_slack_obs;
    if (nso_rt > max_slack_obs) nso_rt = max_slack_obs;
    if (nso_rt < 0) nso_rt = 0;
    const int nso = SLK == 0 ? 0 : nso_rt;
    const int nslk = M * nso;

    const float *Rnx = rows.nx + row_begin, *Rny = rows.ny + row_begin, *Rnz = rows.nz + row_begin;
    const float *Rpx = rows.px + row_begin, *Rpy = rows.py + row_begin, *Rpz = rows.pz + row_begin;
    const double *Rd = rows.d + row_begin;
    const uint32_t *Rkey = rows.key + row_begin;

#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(4);
#endif
    // ---- LSC rows into registers: lane l owns rows l + 64u (coalesced loads, once) --------------
    // Every load of every block of 64 is issued here, unconditionally (clamped index: lanes beyond the rows drop the value
    // below), and NOTHING of it is used before the table reads further down have been issued as well: the rows' trip to HBM
    // and the tables' trip to L2 overlap.  (With a branch and the selects per block of 64, as this was written before, the
    // second block's loads waited for the first block's data.)
    RegF cnx[KU], cny[KU], cnz[KU], cpx[KU], cpy[KU], cpz[KU];
    RegD cd[KU];
    RegU ckey[KU];
    const int Kc = ALL_CACHED ? K : (K < 64 * KU ? K : 64 * KU);  // cached rows
    w.each([&](int l) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            if (Kc > 0) {  // uniform (an agent without rows may sit at the very end of the arrays: nothing is read for it)
                const int j = l + 64 * u;
                const int jc = j < Kc ? j : Kc - 1;
                // (byte offsets of 32 bits: one shift for the seven 4-byte arrays, one for the 8-byte one, added to the arrays' uniform
                // bases by the load itself -- instead of a 64-bit address per array and row)
                const uint32_t o4 = (uint32_t)jc * 4u, o8 = (uint32_t)jc * 8u;
                cnx[u].at(l) = at_offset(Rnx, o4); cny[u].at(l) = at_offset(Rny, o4); cnz[u].at(l) = at_offset(Rnz, o4);
                cpx[u].at(l) = at_offset(Rpx, o4); cpy[u].at(l) = at_offset(Rpy, o4); cpz[u].at(l) = at_offset(Rpz, o4);
                cd[u].at(l) = at_offset(Rd, o8);
                ckey[u].at(l) = at_offset(Rkey, o4);
            } else {
                cnx[u].at(l) = 0; cny[u].at(l) = 0; cnz[u].at(l) = 0;
                cpx[u].at(l) = 0; cpy[u].at(l) = 0; cpz[u].at(l) = 0;
                cd[u].at(l) = 0;
                ckey[u].at(l) = KEY_SKIP;
            }
        }
    });
    auto finish_row_loads = [&]() {  // the first use of the rows: lanes beyond the agent's rows hold no row
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const bool in = l + 64 * u < Kc;
                cnx[u].at(l) = keep_f32(in, cnx[u].at(l)); cny[u].at(l) = keep_f32(in, cny[u].at(l)); cnz[u].at(l) = keep_f32(in, cnz[u].at(l));
                cpx[u].at(l) = keep_f32(in, cpx[u].at(l)); cpy[u].at(l) = keep_f32(in, cpy[u].at(l)); cpz[u].at(l) = keep_f32(in, cpz[u].at(l));
                cd[u].at(l) = keep_f64(in, cd[u].at(l));
                ckey[u].at(l) = ckey[u].at(l) | ~(uint32_t)(-(int32_t)in);  // KEY_SKIP = all ones
            }
        });
    };

    // ---- structured slots into registers (issued here so that all prologue loads are in flight together)
    const Slot *slots = kp.slots + slot0;
    const int n_slots = ALL_CACHED && n_slots_all > 64 * SU ? 64 * SU : n_slots_all;
    const int Sc = ALL_CACHED ? n_slots : (n_slots < 64 * SU ? n_slots : 64 * SU);  // cached slots
    RegU sidx[SUA], scb[SUA];  // 3 x 10-bit x-space indices (0x3FF = unused) | 3 x int8 coefficients + bsel
    RegU sraw[SUA][3];         // the slot records as they are read (16 bytes; the fourth word, the row scaling, is not cached)
    static_assert(sizeof(Slot) == 16, "finish_slot_loads() unpacks the record from its words");
    w.each([&](int l) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            if (Sc > 0) {  // uniform
                const int s = l + 64 * u;
                const uint32_t *sw = reinterpret_cast<const uint32_t *>(slots + (s < Sc ? s : Sc - 1));
                sraw[u][0].at(l) = sw[0]; sraw[u][1].at(l) = sw[1]; sraw[u][2].at(l) = sw[2];
            } else {
                sraw[u][0].at(l) = 0; sraw[u][1].at(l) = 0; sraw[u][2].at(l) = 0;
            }
        }
    });
    auto finish_slot_loads = [&]() {  // the first use of the slot records (little-endian words of struct Slot)
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const uint32_t d0 = sraw[u][0].at(l), d1 = sraw[u][1].at(l), d2 = sraw[u][2].at(l);
                // idx[0] idx[1] | idx[2] coef[0] coef[1] | coef[2] kind bsel
                const uint32_t pi = (d0 & 0x3FFu) | (((d0 >> 16) & 0x3FFu) << 10) | ((d1 & 0x3FFu) << 20);
                const uint32_t pc = ((d1 >> 16) & 0xFFu) | (((d1 >> 24) & 0xFFu) << 8) | ((d2 & 0xFFu) << 16) | ((d2 >> 16) << 24);
                const bool in = l + 64 * u < Sc;
                const uint32_t m = (uint32_t)(-(int32_t)in);
                sidx[u].at(l) = (pi & m) | (0xBFFFFFFFu & ~m);  // bit 31: no slot in this register
                scb[u].at(l) = pc & m;
            }
        });
    };

    // ---- static shapes: the tables of the agent's number of terminal segments T, read AHEAD of the goal LP ------------------
    // T depends on the sub-goal, which is what the LP is about to compute -- but the LP's answer is the waypoint itself whenever
    // the waypoint can be reached (t = 0: three of four agent-steps of the reference's logged run), so the tables of
    // T(waypoint) are requested now, their round trip to L2 runs under the LP's arithmetic, and the few agents whose LP stops
    // short in a way that changes T read again (uniform branch below).  Every table read of the phase -- x0 factors, pivot
    // scalings of the rows by kind, of the slots, of the cached LSC rows, the quadratic form of the unconstrained objective -- is
    // issued before the first use: one exposed round trip instead of one per table, and none when the guess holds.
    constexpr int NXP = DIMS > 0 ? (DIMS * MS * 6 + 63) / 64 : 1;
    RegD ciw[KU], siw[SUA], nnr[KU];
    RegD kiw[NXP][3];  // BYK: pivot scalings of the bound / velocity / acceleration rows of the lane's coordinates
    RegD xa0[NXP], xa1[NXP], xa2[NXP], xag[NXP];  // x0 = xa0 pos + xa1 vel + xa2 acc + xag goal, per coordinate of the lane
    double phq[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // tabPhi of T (uniform)
    auto terminal_segments = [&](const float *gl) -> int {  // traj_optimizer.cpp:543-551 (float32 point difference)
        const double ift = f32_norm3(gl[0] - pos[0], gl[1] - pos[1], gl[2] - pos[2]) / nominal_velocity;
        int Tt = (int)((M * kp.dt - ift + SP_EPSILON) / kp.dt);
        Tt = Tt < 1 ? 1 : Tt;
        return Tt > M ? M : Tt;
    };
    auto load_tables = [&](int Tt) {
        const double *tXs = kp.tabX0s + (size_t)(Tt - 1) * P * 3, *tXg = kp.tabX0g + (size_t)(Tt - 1) * P;
        const double *tSIW = kp.tabSIW + (size_t)(Tt - 1) * kp.n_slots + slot0;
        const double *tKIW = kp.tabKIW + (size_t)(Tt - 1) * 3 * P;
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const int j = 64 * u + l;
                const int jc = j < nxc ? j : nxc - 1;
                const int p = jc - (jc / P) * P;
                xa0[u].at(l) = tXs[p * 3 + 0]; xa1[u].at(l) = tXs[p * 3 + 1]; xa2[u].at(l) = tXs[p * 3 + 2]; xag[u].at(l) = tXg[p];
                kiw[u][0].at(l) = tKIW[p]; kiw[u][1].at(l) = tKIW[P + p]; kiw[u][2].at(l) = tKIW[2 * P + p];
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int s = l + 64 * u;
                const double v = tSIW[s < Sc ? s : (Sc > 0 ? Sc - 1 : 0)];
                siw[u].at(l) = s < Sc ? v : 0.0;
            }
        });
        if (kp.phi_ok) {
            const double *ph = kp.tabPhi + (size_t)(Tt - 1) * 10;
#pragma unroll
            for (int j = 0; j < 10; j++) phq[j] = ph[j];
        }
    };
    auto load_row_scalings = [&](int Tt) {  // |N_T[p,:]|^2 of the cached rows' control points: needs the rows' keys
        const double *tNN = kp.tabNN + (size_t)(Tt - 1) * P;
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const uint32_t key = ckey[u].at(l);
                const bool have = key != KEY_SKIP;
                int mi = (int)(key & 0xFF) * n1 + (int)((key >> 8) & 0xF);
                DQ_KEEP(mi);  // computed for every lane, then selected: as one ?: expression this came out as an EXEC-masked region
                mi = have ? mi : 0;
                nnr[u].at(l) = tNN[mi < P ? mi : 0];
            }
        });
    };
    int T_ahead = 0;
#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(5);
#endif
    if (DIMS > 0 && do_traj && !bad_shape) {
        T_ahead = terminal_segments(do_goal ? wp : cg);
        load_tables(T_ahead);
    }
    finish_row_loads();
    finish_slot_loads();
    if (DIMS > 0 && do_traj && !bad_shape) load_row_scalings(T_ahead);

    // ================================ goal LP ===================================================
    // min t in [0, 1+1e-5] s.t. a_r t + b_r >= 0 (goal_optimizer.cpp:138-198), closed form.
    float goal[3] = {cg[0], cg[1], cg[2]};
    int goal_status = DLSCQP_OK;
#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(6);
#endif
    if (do_goal && !bad_shape) {
        float dx = cg[0] - wp[0], dy = cg[1] - wp[1], dz = cg[2] - wp[2];
        if (f32_norm3_below_eps(dx, dy, dz)) {  // goal_optimizer.cpp:12-14 (the same test without the square root)
            goal[0] = wp[0]; goal[1] = wp[1]; goal[2] = wp[2];
        } else {
            const float diff[3] = {dx, dy, dz};
            double t_lo = 0.0, t_hi = 1.0 + SP_EPSILON_FLOAT;
            int infeasible = 0;
            if (kp.use_sfc && sfc) {  // goal_optimizer.cpp:158-173
                const float *bmin = lastbox, *bmax = lastbox + 3;
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
            // goal_optimizer.cpp:176-195: last control point of the last segment, static obstacles.
            // One pass over the rows: per-lane bounds, then max / min / any across the wave.
            RegD llo, lhi;
            RegI lbad, lviol;
            // a row's coefficients (select form, no lane-dependent branch), then its bound t >= -b / a or t <= -b / a
            auto lp_coeff = [&](uint32_t key, float nx, float ny, float nz, float qx, float qy, float qz, double dd,
                                bool &in_lp, double &a, double &b, int &viol) {
                const int m = key & 0xFF, i = (key >> 8) & 0xF, dyn = (key >> 12) & 1;
                in_lp = (key != KEY_SKIP) & (dyn == 0) & (m == M - 1) & (i == n) & !f32_norm3_below_eps(nx, ny, nz);
                const float ox = wp[0] - qx, oy = wp[1] - qy, oz = wp[2] - qz;
                a = (double)nx * (double)diff[0] + (double)ny * (double)diff[1];
                b = -dd + (double)nx * (double)ox + (double)ny * (double)oy;
                const double a3 = a + (double)nz * (double)diff[2], b3 = b + (double)nz * (double)oz;
                a = dim == 3 ? a3 : a;
                b = dim == 3 ? b3 : b;
                // exception branch of goal_optimizer.cpp:63-77: does current_goal_point violate the row?
                const float e0 = cg[0] - qx, e1 = cg[1] - qy, e2 = cg[2] - qz;
                const float dot = nx * e0 + ny * e1 + nz * e2;  // float32 dot of float32 differences
                viol = (in_lp & ((double)dot - dd < -SP_EPSILON_FLOAT)) ? 1 : viol;
            };
            auto lp_bound = [&](bool in_lp, double a, double b, double &lo, double &hi, int &bad) {
                // rows that are not part of the LP leave lo / hi / bad as they are
                double tq = -b / a;  // not finite when a == 0: never selected then
                DQ_KEEP(tq);
                lo = (in_lp & (a > 0) & (tq > lo)) ? tq : lo;
                hi = (in_lp & (a < 0) & (tq < hi)) ? tq : hi;
                bad = (in_lp & !(a > 0) & !(a < 0) & (b < -CPLEX_FEAS_TOL)) ? 1 : bad;
            };
            // The rows of the LP are those of the last control point: a few of an agent's rows, and hardly ever two of one lane's
            // cached rows.  Where no lane has two (one ballot), each lane divides once, for the row it has (a double-precision
            // division is ~35 instructions); else row by row.
            RegD c_a[KU], c_b[KU];
            RegI c_in_r[KU], lviol0, lmulti;
            w.each([&](int l) {
                int viol = 0, cnt = 0;
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    double a, b;
                    bool in_u;
                    lp_coeff(ckey[u].at(l), cnx[u].at(l), cny[u].at(l), cnz[u].at(l), cpx[u].at(l), cpy[u].at(l), cpz[u].at(l), cd[u].at(l),
                             in_u, a, b, viol);
                    c_a[u].at(l) = a; c_b[u].at(l) = b; c_in_r[u].at(l) = in_u ? 1 : 0;
                    cnt += in_u ? 1 : 0;
                }
                lviol0.at(l) = viol;
                lmulti.at(l) = cnt > 1 ? 1 : 0;
            });
            const bool lp_multi = w.any(lmulti);
            w.each([&](int l) {
                double lo = -1e300, hi = 1e300;
                int bad = 0, viol = lviol0.at(l);
                if (!lp_multi) {  // uniform
                    bool in = false;
                    double a = 1.0, b = 0.0;
#pragma unroll
                    for (int u = 0; u < KU; u++) {
                        const bool iu = c_in_r[u].at(l) != 0;
                        a = iu ? c_a[u].at(l) : a; b = iu ? c_b[u].at(l) : b;
                        in |= iu;
                    }
                    lp_bound(in, a, b, lo, hi, bad);
                } else {
#pragma unroll
                    for (int u = 0; u < KU; u++) lp_bound(c_in_r[u].at(l) != 0, c_a[u].at(l), c_b[u].at(l), lo, hi, bad);
                }
                // rows beyond the register cache (uniform trip count): CB blocks of 64 rows per trip, all their loads
                // issued before the first use (one HBM round trip per CB blocks)
                if (!ALL_CACHED)
                for (int j0 = 64 * KU; j0 < K; j0 += 64 * CB) {
                    uint32_t kk[CB];
                    float g[CB][6];
                    double gd[CB];
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        const int j = j0 + 64 * u + l;
                        const int jc = j < K ? j : K - 1;
                        const uint32_t o4 = (uint32_t)jc * 4u, o8 = (uint32_t)jc * 8u;  // (32-bit byte offsets: at_offset)
                        kk[u] = at_offset(Rkey, o4);
                        g[u][0] = at_offset(Rnx, o4); g[u][1] = at_offset(Rny, o4); g[u][2] = at_offset(Rnz, o4);
                        g[u][3] = at_offset(Rpx, o4); g[u][4] = at_offset(Rpy, o4); g[u][5] = at_offset(Rpz, o4);
                        gd[u] = at_offset(Rd, o8);
                    }
#pragma unroll
                    for (int u = 0; u < CB; u++) {  // loaded by every lane: no branch around the loads of rows past the end
                        DQ_KEEP(kk[u]); DQ_KEEP(gd[u]);
#pragma unroll
                        for (int t = 0; t < 6; t++) DQ_KEEP(g[u][t]);
                    }
                    // (one division per lane and trip where no lane has two rows of the LP among its CB rows: the rows of the last
                    // control point are one in 27 of a neighbour's rows)
                    bool tin[CB];
                    double ta[CB], tb[CB];
                    int tcnt = 0;
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        const uint32_t kq = blend_idx(j0 + 64 * u + l < K, (int)kk[u], (int)KEY_SKIP);
                        lp_coeff(kq, g[u][0], g[u][1], g[u][2], g[u][3], g[u][4], g[u][5], gd[u], tin[u], ta[u], tb[u], viol);
                        tcnt += tin[u] ? 1 : 0;
                    }
                    if (DQ_WAVE_ALL(tcnt <= 1)) {  // uniform
                        bool in = false;
                        double a = 1.0, b = 0.0;
#pragma unroll
                        for (int u = 0; u < CB; u++) {
                            a = tin[u] ? ta[u] : a; b = tin[u] ? tb[u] : b;
                            in |= tin[u];
                        }
                        lp_bound(in, a, b, lo, hi, bad);
                    } else {
#pragma unroll
                        for (int u = 0; u < CB; u++) lp_bound(tin[u], ta[u], tb[u], lo, hi, bad);
                    }
                }
                llo.at(l) = lo; lhi.at(l) = hi; lbad.at(l) = bad; lviol.at(l) = viol;
            });
            const double lo2 = w.max(llo), hi2 = w.min(lhi);
            if (lo2 > t_lo) t_lo = lo2;
            if (hi2 < t_hi) t_hi = hi2;
            if (w.any(lbad) || t_lo > t_hi + CPLEX_FEAS_TOL) infeasible = 1;
            if (!infeasible) {
                float tf = (float)t_lo;  // Vector3::operator*(float), goal_optimizer.cpp:51
                for (int k = 0; k < 3; k++) goal[k] = diff[k] * tf + wp[k];
            } else {
                // exception branch, goal_optimizer.cpp:54-81
                int numerical_error = 1;
                if (kp.use_sfc && sfc) {
                    const float *bmin = lastbox, *bmax = lastbox + 3;
                    for (int k = 0; k < 3; k++)
                        if (!((double)cg[k] > (double)bmin[k] - SP_EPSILON_FLOAT &&
                              (double)cg[k] < (double)bmax[k] + SP_EPSILON_FLOAT))
                            numerical_error = 0;
                }
                if (w.any(lviol)) numerical_error = 0;
                goal_status = numerical_error ? DLSCQP_GOAL_FALLBACK : DLSCQP_INFEASIBLE;
            }
        }
    }
    // Scalar results (sub-goal, verdicts, counters, cost) are stored by lane 0 (lanes 0-2 for the sub-goal) in
    // exit_stores(), which every exit path calls as its LAST action: a lane-guarded store is a divergent region,
    // allowed only where nothing is live after it (see "NO DIVERGENCE").
    auto exit_stores = [&](int st_traj, bool have_traj, int it0, int it1, int it2, int it3, bool have_cost, double fval) {
        DQ_MARK("DQ_EXIT_STORES");
        if (iters_done) *iters_done = it0;
        w.each([&](int l) {
            if (do_goal) {
                if (l < 3 && out_goal && goal_status != DLSCQP_INFEASIBLE && goal_status != DLSCQP_NUMERICAL) out_goal[l] = pick3(goal, l);
                if (l == 0 && out_goal_status) *out_goal_status = goal_status;
            }
            if (l == 0 && have_traj) {
                if (out_status) *out_status = st_traj;
                if (out_iters) { out_iters[0] = it0; out_iters[1] = it1; out_iters[2] = it2; out_iters[3] = it3; }
                if (have_cost && out_cost) *out_cost = fval;
            }
        });
    };
    if (bad_shape) {  // uniform
        goal_status = DLSCQP_NUMERICAL;
        exit_stores(DLSCQP_NUMERICAL, do_traj != 0, 0, 0, 0, 0, false, 0.0);
        return;
    }
    if (do_goal && goal_status == DLSCQP_INFEASIBLE) {
        // the reference throws before trajOptimization is reached (traj_planner.cpp:491-496)
        exit_stores(DLSCQP_INFEASIBLE, do_traj != 0, 0, 0, 0, 0, false, 0.0);
        return;
    }
#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(7);
#endif
    w.stamp(0);  // agent record, row cache, goal LP
    DQ_PHASE_EXIT(0);
    if (!do_traj) {
        exit_stores(0, false, 0, 0, 0, 0, false, 0.0);
        return;
    }

    // ================================ trajectory QP =============================================
    double *base = reinterpret_cast<double *>(mem);
    // layout (make_layout): compile-time for static shapes without slack obstacles, else from the arguments
    constexpr bool ST = DIMS > 0 && SLK == 0;
    constexpr int LDG_ST = (MS * 6) | 1, R_ST = DIMS * FS, NX_ST = DIMS * MS * 6;
    constexpr int ODUMP_ST = 0, OX_ST = 32 + 2, OC_ST = OX_ST + NX_ST, CS_ST = (R_ST + NX_ST) | 1;  // (make_layout; ST shapes keep no dx in LDS)
    const int r_max = ST ? R_ST : lay.r_max;
    const int ldg = DIMS > 0 ? LDG_ST : lay.ldg;
    const int odump = ST ? ODUMP_ST : lay.oDump;  // 32 dump slots, base[odump + (lane & 31)]
    const int oX = ST ? OX_ST : lay.oX;
    const int oDx = lay.oDx;  // (DXREG: none)
    const int oBlo = lay.oBlo, oBhi = lay.oBhi;
    // column records of the active rows (MemLayout): record k = [ S(0..r_max-1, k) | Z(0..nx-1, k) ]
    const int oC = ST ? OC_ST : lay.oC;
    const int cs = ST ? CS_ST : lay.cs;
    const int zo = r_max;   // where the Z part of a record starts
    const int qz = lay.qz;  // records in the working set; records qz .. live in the spill record
    double *colw = base + oC;
    double *colsp = spill ? spill + 32 : nullptr;  // (32 dump doubles in front: colsp[-32 + (lane & 31)])
    double *x = base + oX, *dxl = base + (oDx < 0 ? 0 : oDx);
    double *blo = base + oBlo, *bhi = base + oBhi;
    unsigned char *cold_flag = mem + lay.oFlag, *row_flag = mem + lay.oFlagRows;
    const int rc = dim * f;      // whitened control-point dimension
    const int r = rc + nslk;     // whitened dimension: the active rows are independent, there are never more than r
    // rows beyond the register cache: [Kc, Kl) decoded once into the working set ("rows in LDS": right-hand side, pivot scaling,
    // normal, key -- 32 bytes a row, priced from there in every pass), [Kl, K) priced from HBM in every pass
    const int Kl = ALL_CACHED ? K : (K < Kc + lay.cold_cap ? K : Kc + lay.cold_cap);
    double *lrhs = base + lay.oCold, *liw = lrhs + lay.cold_cap;
    float *lnx = reinterpret_cast<float *>(liw + lay.cold_cap), *lny = lnx + lay.cold_cap, *lnz = lny + lay.cold_cap;
    uint32_t *lkey = reinterpret_cast<uint32_t *>(lnz + lay.cold_cap);

    // terminal segments, traj_optimizer.cpp:543-551 (the value computed ahead of the LP stands whenever the LP answered with the
    // very point it was computed from -- the waypoint: three agent-steps of four --: a norm, a square root and two divisions less)
    const float *gl_ahead = do_goal ? wp : cg;
    const bool same_goal = T_ahead > 0 && goal[0] == gl_ahead[0] && goal[1] == gl_ahead[1] && goal[2] == gl_ahead[2];
    const int T = same_goal ? T_ahead : terminal_segments(goal);
    int status = DLSCQP_OK;
    if (!kp.t_valid[T - 1]) status = DLSCQP_NUMERICAL;
    if (nslk > 0 && !(kp.w_s > 0)) status = DLSCQP_NUMERICAL;
    if (r > 64 * RC || r > r_max || P > 255 || dim != kp.dim || f != kp.f || M != kp.M || n1 != kp.n1) status = DLSCQP_NUMERICAL;  // the launcher picks the instantiation; defensive
    if (qz < r_max && !spill) status = DLSCQP_NUMERICAL;  // (the launcher sizes the layout for the spill record it has; defensive)
    if (ST && (lay.oX != OX_ST || lay.oC != OC_ST || lay.cs != CS_ST || lay.oDump != ODUMP_ST)) status = DLSCQP_NUMERICAL;  // make_layout and the constants above agree
    if (status != DLSCQP_OK) {
        exit_stores(status, true, 0, 0, 0, 0, false, 0.0);
        return;
    }

    const double *G = kp.tabG + (size_t)(T - 1) * P * ldg;  // the Gram table of T, in HBM / L2: an iteration reads rows of it
    const double *tXs = kp.tabX0s + (size_t)(T - 1) * P * 3;
    const double *tXg = kp.tabX0g + (size_t)(T - 1) * P;
    const double *tNN = kp.tabNN + (size_t)(T - 1) * P;
    const double *tSIW = kp.tabSIW + (size_t)(T - 1) * kp.n_slots + slot0;

#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(8);
#endif
    // ---- unconstrained minimiser and bound table into LDS -------------------------------------------
    // (uniform trip counts; a lane past the end computes a clamped element and stores it to the dump slot)
    double fval0 = 0.0;  // objective of the unconstrained minimiser: what an agent that never iterates reports
    if (DIMS > 0) {
        if (T != T_ahead) {  // uniform: the LP stopped short of the waypoint far enough to change T
            load_tables(T);
            load_row_scalings(T);
        }
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const int j = 64 * u + l;
                const bool in = j < nxc;
                const int k = (in ? j : nxc - 1) / P;
                const double v = xa0[u].at(l) * (double)pick3(pos, k) + xa1[u].at(l) * (double)pick3(vel, k) + xa2[u].at(l) * (double)pick3(acc, k) +
                                 xag[u].at(l) * (double)pick3(goal, k);
                base[blend_idx(in, oX + j, odump + (l & 31))] = v;
                if (LEAN) {  // the coordinate's pivot scalings: read back by every pricing pass (kind-major: the lane's address + a constant)
#pragma unroll
                    for (int t = 0; t < 3; t++) base[blend_idx(in, lay.oKiw + t * nxc + j, odump + (l & 31))] = kiw[u][t].at(l);
                }
            }
        });
        // its objective is a 4 x 4 quadratic form per axis (qp_tables.cpp: tabPhi), evaluated here, where its table has just
        // arrived with the others, not at the exit, where the read would be a round trip of its own
        if (out_cost && kp.phi_ok) {
            for (int k = 0; k < dim; k++) {
                const double s0 = (double)pos[k], s1 = (double)vel[k], s2 = (double)acc[k], s3 = (double)goal[k];
                fval0 += s0 * (phq[0] * s0 + 2.0 * (phq[1] * s1 + phq[2] * s2 + phq[3] * s3)) + s1 * (phq[4] * s1 + 2.0 * (phq[5] * s2 + phq[6] * s3)) +
                         s2 * (phq[7] * s2 + 2.0 * phq[8] * s3) + phq[9] * s3 * s3;
            }
        }
    }
    const double hv_lim = max_vel * kp.vel_scale, ha_lim = max_acc * kp.acc_scale;  // dynamic limits in metres (T3g)
    w.each([&](int l) {
        if (DIMS == 0) {
        for (int j0 = 0; j0 < nxc; j0 += 64) {
            const int j = j0 + l;
            const bool in = j < nxc;
            const int jc = in ? j : nxc - 1;
            const int k = jc / P, p = jc - k * P;
            const double v = tXs[p * 3 + 0] * (double)pick3(pos, k) + tXs[p * 3 + 1] * (double)pick3(vel, k) +
                             tXs[p * 3 + 2] * (double)pick3(acc, k) + tXg[p] * (double)pick3(goal, k);
            base[blend_idx(in, oX + j, odump + (l & 31))] = v;
        }
        }
        for (int s0 = 0; s0 < nslk; s0 += 64) {
            const int s2 = s0 + l;
            base[blend_idx(s2 < nslk, oX + nxc + s2, odump + (l & 31))] = 0.0;
        }
        base[odump + 32 + (l & 1)] = 0.0;  // every lane, the same two zeros
        for (int j0 = 0; j0 < lay.nflag; j0 += 64) {
            const int j = j0 + l;
            cold_flag[j < lay.nflag ? j : lay.nflag] = 0;  // [nflag] is the dump byte
        }
        // per-agent bound table: boxes (world and SFC), velocity, acceleration, comm range, waypoint
        for (int b0 = 0; b0 < kp.nb; b0 += 64) {
            const int b = b0 + l;
            const bool in = b < kp.nb;
            const int bc = in ? b : 0;
            // box entry (axis k, segment m)
            const bool is_box = bc < dim * M;
            const int bb = is_box ? bc : 0;
            const int k = bb / M, m = bb - k * M;
            double blo_box = pick3(kp.world_min, k), bhi_box = pick3(kp.world_max, k);
            const bool rsfc = kp.rsfc_mode && k == 2 && m == 0;  // traj_optimizer.cpp:254-258
            blo_box = rsfc ? -100.0 : blo_box;
            bhi_box = rsfc ? 100.0 : bhi_box;
            if (kp.use_sfc && sfc) {  // uniform.  Box::convertToLSCs, collision_constraints.cpp:66-87
                const double smin = (double)sfc[m * 6 + k], smax = (double)sfc[m * 6 + 3 + k];
                blo_box = smin > blo_box ? smin : blo_box;
                bhi_box = smax < bhi_box ? smax : bhi_box;
            }
            const double hc = 0.5 * kp.comm_range - radius;  // traj_optimizer.cpp:495-500
            const int kw = bc - (dim * M + 3);
            const double cw = 0.5 * kp.comm_range - SP_EPSILON_FLOAT;  // traj_optimizer.cpp:507-510
            const double wpk = (double)pick3(wp, kw < 0 ? 0 : kw);
            double lo = wpk - cw, hi = wpk + cw;
            lo = bc == dim * M + 2 ? -hc : lo; hi = bc == dim * M + 2 ? hc : hi;
            lo = bc == dim * M + 1 ? -ha_lim : lo; hi = bc == dim * M + 1 ? ha_lim : hi;
            lo = bc == dim * M ? -hv_lim : lo;     hi = bc == dim * M ? hv_lim : hi;
            lo = is_box ? blo_box : lo;        hi = is_box ? bhi_box : hi;
            base[blend_idx(in, oBlo + b, odump + (l & 31))] = lo;
            base[blend_idx(in, oBhi + b, odump + (l & 31))] = hi;
        }
    });
    if (DIMS == 0) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            w.each([&](int l) { siw[u].at(l) = 0.0; });
            if (64 * u < Sc) {  // uniform
                w.each([&](int l) {
                    const int s = l + 64 * u;
                    const double v = tSIW[s < Sc ? s : Sc - 1];
                    siw[u].at(l) = s < Sc ? v : 0.0;
                });
            }
        }
    }
#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(9);
#endif
    // finish the LSC cache: rhs = d + n.p (traj_optimizer.cpp:427-439); a row that is not part of the model keeps a zero
    // normal and the right-hand side -NEVER: it is never violated, and the pricing pass needs no test for it
    RegI cpo[KU];   // control point of the row
    RegU ltag[KU];  // the row's code (tag of its pivot score)
#pragma unroll
    for (int u = 0; u < KU; u++) {
        w.each([&](int l) {
            const uint32_t key = ckey[u].at(l);
            const bool have = key != KEY_SKIP;
            const int m = have ? (int)(key & 0xFF) : 0, i = have ? (int)((key >> 8) & 0xF) : 0;
            const float nx = cnx[u].at(l), ny = cny[u].at(l), nz = cnz[u].at(l);
            // rows the reference leaves out of the model (:417-424)
            const bool in_model = have & !(((m == 0) & (i < kp.phi)) | f32_norm3_below_eps(nx, ny, nz));
            double rhs = cd[u].at(l) + (double)nx * (double)cpx[u].at(l) + (double)ny * (double)cpy[u].at(l);
            double nsq = (double)nx * (double)nx + (double)ny * (double)ny;
            const double rhs3 = rhs + (double)nz * (double)cpz[u].at(l), nsq3 = nsq + (double)nz * (double)nz;
            rhs = dim == 3 ? rhs3 : rhs;
            nsq = dim == 3 ? nsq3 : nsq;
            uint32_t sl = key >> 16;
            const bool no_such_slack = sl != 0xFFFFu && (int)sl >= nso;
            const uint32_t key2 = no_such_slack ? (key | 0xFFFF0000u) : key;
            sl = no_such_slack ? 0xFFFFu : sl;
            // pivot scaling 1 / |N' a|  (whitened length of the row normal)
            const int mi = m * n1 + i;
            double tnn = DIMS > 0 ? nnr[u].at(l) : tNN[mi < P ? mi : 0];
            DQ_KEEP(tnn);
            double wsq = nsq * tnn;
            if (SLK != 0) {  // (a build without slack variables has no use for the weight: two dependent trips to the table less)
                double sg = kp.sigma[m < M ? m : 0];
                DQ_KEEP(sg);
                wsq = sl != 0xFFFFu ? wsq + sg * sg : wsq;
            }
            DQ_KEEP(rhs);
            const bool live = in_model & (wsq > 0);
            cd[u].at(l) = live ? rhs : -NEVER;
            cnx[u].at(l) = live ? nx : 0.0f; cny[u].at(l) = live ? ny : 0.0f; cnz[u].at(l) = live ? nz : 0.0f;
            ckey[u].at(l) = live ? key2 : KEY_SKIP;
            cpo[u].at(l) = live && mi < P ? mi : 0;
            ltag[u].at(l) = (uint32_t)((BYK ? nxc : 0) + l + 64 * u) << 3;
            if (SCREEN) {
                ciw[u].at(l) = live ? wsq : 1.0;  // (its reciprocal square root behind the screen, below: most agents never need it)
            } else {
                double iw = DQ_RSQRT(live ? wsq : 1.0);
                DQ_KEEP(iw);
                ciw[u].at(l) = iw;
            }
        });
    }
    // ---- rows in LDS: the rows right behind the register cache, decoded like the cached ones, written once --------------------
    // An agent of a crowd has a thousand rows and a few dozen pricing passes: priced from HBM a pass is three or four round
    // trips of a couple of microseconds for ONE wavefront (the flown 2 048-agent crowd: 7 us a pass, up to 47 passes: the step
    // was its slowest agent); from here a pass over the same rows is LDS reads.  Trips of CB blocks of 64 rows, every load of a
    // trip issued before its first use; lanes past the last row recompute it (the same values to the same place).
    if (!ALL_CACHED && Kl > Kc) {  // uniform
        w.each([&](int l) {
            for (int j0 = Kc; j0 < Kl; j0 += 64 * CB) {
                uint32_t kk[CB];
                float g[CB][6];
                double gd[CB];
#pragma unroll
                for (int u = 0; u < CB; u++) {
                    const int j = j0 + 64 * u + l;
                    const int jc = j < Kl ? j : Kl - 1;
                    const uint32_t o4 = (uint32_t)jc * 4u, o8 = (uint32_t)jc * 8u;  // (32-bit byte offsets: at_offset)
                    kk[u] = at_offset(Rkey, o4);
                    g[u][0] = at_offset(Rnx, o4); g[u][1] = at_offset(Rny, o4); g[u][2] = at_offset(Rnz, o4);
                    g[u][3] = at_offset(Rpx, o4); g[u][4] = at_offset(Rpy, o4); g[u][5] = at_offset(Rpz, o4);
                    gd[u] = at_offset(Rd, o8);
                }
#pragma unroll
                for (int u = 0; u < CB; u++) {
                    DQ_KEEP(kk[u]); DQ_KEEP(gd[u]);
#pragma unroll
                    for (int t = 0; t < 6; t++) DQ_KEEP(g[u][t]);
                }
#pragma unroll
                for (int u = 0; u < CB; u++) {
                    const int j = j0 + 64 * u + l;
                    const int jc = (j < Kl ? j : Kl - 1) - Kc;
                    const uint32_t key = kk[u];
                    const int m = key & 0xFF, i = (key >> 8) & 0xF;
                    const float nx = g[u][0], ny = g[u][1], nz = g[u][2];
                    const bool in_model = !(((m == 0) & (i < kp.phi)) | f32_norm3_below_eps(nx, ny, nz)) & (m < M) & (i < n1);
                    double rhs = gd[u] + (double)nx * (double)g[u][3] + (double)ny * (double)g[u][4];
                    double nsq = (double)nx * (double)nx + (double)ny * (double)ny;
                    const double rhs3 = rhs + (double)nz * (double)g[u][5], nsq3 = nsq + (double)nz * (double)nz;
                    rhs = dim == 3 ? rhs3 : rhs;
                    nsq = dim == 3 ? nsq3 : nsq;
                    uint32_t sl = key >> 16;
                    const bool no_such_slack = sl != 0xFFFFu && (int)sl >= nso;
                    const uint32_t key2 = no_such_slack ? (key | 0xFFFF0000u) : key;
                    sl = no_such_slack ? 0xFFFFu : sl;
                    const int mi = in_model ? m * n1 + i : 0;
                    double tnn = tNN[mi];
                    DQ_KEEP(tnn);
                    double wsq = nsq * tnn;
                    double sg = kp.sigma[in_model ? m : 0];
                    DQ_KEEP(sg);
                    wsq = sl != 0xFFFFu ? wsq + sg * sg : wsq;
                    const bool live = in_model & (wsq > 0);
                    double iw = DQ_RSQRT(live ? wsq : 1.0);
                    DQ_KEEP(iw);
                    lrhs[jc] = live ? rhs : -NEVER;
                    liw[jc] = iw;
                    lnx[jc] = live ? nx : 0.0f; lny[jc] = live ? ny : 0.0f; lnz[jc] = live ? nz : 0.0f;
                    lkey[jc] = live ? key2 : (key2 & 0xFFFF0000u);  // (a row that is not in the model: control point 0, zero normal)
                }
            }
        });
    }
    w.sync();
    // BYK: the limits of the lane's coordinates (after the sync: the box faces come from the bound table in LDS)
    RegD blo_r[NXP], bhi_r[NXP], hv_r[NXP], ha_r[NXP];
    RegU jtag[NXP];
    RegU lim_off;  // LEAN: bit 2 u = the velocity rows of the lane's coordinate u are out of the pass (not in the model, or one is active), 2 u + 1: acceleration
    const double hv_c = hv_lim + VIOL_TOL, ha_c = ha_lim + VIOL_TOL;  // (limit + tolerance: what a pass compares with)
    if (BYK) {
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const int j = 64 * u + l;
                const bool in = j < nxc;
                const int jc = in ? j : nxc - 1;
                const int k = jc / P, p = jc - k * P, m = p / n1, i = p - m * n1;
                double lo = blo[k * M + m], hi = bhi[k * M + m];
                DQ_KEEP(lo); DQ_KEEP(hi);
                const bool has_b = in & !((m == 0) & (i < 3));              // traj_optimizer.cpp:260-262, 390-392
                const bool has_v = in & (i < n) & !((m == 0) & (i < 2));    // :457-459
                const bool has_a = in & (i < n - 1) & !((m == 0) & (i == 0));  // :471-473
                // (the registers hold limit -/+ tolerance, what the pricing pass compares with: NEVER stays NEVER)
                blo_r[u].at(l) = has_b ? lo - VIOL_TOL : -NEVER;
                bhi_r[u].at(l) = has_b ? hi + VIOL_TOL : NEVER;
                hv_r[u].at(l) = has_v ? hv_lim + VIOL_TOL : NEVER;
                ha_r[u].at(l) = has_a ? ha_lim + VIOL_TOL : NEVER;
                jtag[u].at(l) = (uint32_t)jc << 3;
                const uint32_t ob = ((has_v ? 0u : 1u) | (has_a ? 0u : 2u)) << (2 * u);
                lim_off.at(l) = u == 0 ? ob : (lim_off.at(l) | ob);
            }
        });
    }
    // the velocity / acceleration limit of the lane's coordinate u as a pass (or the screen) sees it
    auto lim_v = [&](uint32_t off, int u) -> double { return LEAN ? (((off >> (2 * u)) & 1u) ? NEVER : hv_c) : 0.0; };
    auto lim_a = [&](uint32_t off, int u) -> double { return LEAN ? (((off >> (2 * u + 1)) & 1u) ? NEVER : ha_c) : 0.0; };
#ifdef DQ_PROLOGUE_STAMPS
    w.stamp(10);
#endif
    // ---- SCREEN (the instantiation whose rows are all in registers): is any row violated at the unconstrained minimiser? --------
    // Two agents of three of the synthetic swarms sit at their unconstrained optimum: for them the first pricing pass only has
    // to say "none", and this says it without pivot scalings, row codes, the arg-max and everything the iterations set up before
    // their first pass.  Every row's violation is the pricing pass's own expression, (limit - tolerance) - value, so a row that
    // pass would pick (violation x scaling >= SCORE_MIN, scaling >= 0) is violated here as well: the screen lets through a
    // superset, and an agent it lets through by a hair finds no row in its first pass and leaves with zero iterations like the others.
    bool screened = false;
    if (SCREEN) {
        RegI sv;
        w.each([&](int l) {
            double xr[KU][3], c0[NXP], c1[NXP], c2[NXP];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int p = cpo[u].at(l);
                xr[u][0] = x[p]; xr[u][1] = x[P + p]; xr[u][2] = dim == 3 ? x[2 * P + p] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const int j = 64 * u + l;
                const int j0 = j < nxc ? j : nxc - 1, j1 = j + 1 < nxc ? j + 1 : nxc - 1, j2 = j + 2 < nxc ? j + 2 : nxc - 1;
                c0[u] = x[j0]; c1[u] = x[j1]; c2[u] = x[j2];
            }
#pragma unroll
            for (int u = 0; u < KU; u++) { DQ_KEEP(xr[u][0]); DQ_KEEP(xr[u][1]); DQ_KEEP(xr[u][2]); }
#pragma unroll
            for (int u = 0; u < NXP; u++) { DQ_KEEP(c0[u]); DQ_KEEP(c1[u]); DQ_KEEP(c2[u]); }
            bool any = false;
#pragma unroll
            for (int u = 0; u < KU; u++) {
                double val = (double)cnx[u].at(l) * xr[u][0];
                val = fma((double)cny[u].at(l), xr[u][1], val);
                if (dim == 3) val = fma((double)cnz[u].at(l), xr[u][2], val);
                any |= (cd[u].at(l) - VIOL_TOL) - val > 0.0;
            }
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const double dv = c1[u] - c0[u], da = (c2[u] - c1[u]) - dv;
                const double hv = LEAN ? lim_v(lim_off.at(l), u) : hv_r[u].at(l), ha = LEAN ? lim_a(lim_off.at(l), u) : ha_r[u].at(l);
                any |= (blo_r[u].at(l) - c0[u] > 0.0) | (c0[u] - bhi_r[u].at(l) > 0.0) | (-hv - dv > 0.0) | (dv - hv > 0.0) |
                       (-ha - da > 0.0) | (da - ha > 0.0);
            }
            sv.at(l) = any ? 1 : 0;
        });
        screened = !w.any(sv);
#ifdef DQ_CUT_AFTER_SCREEN  // developer experiment (tests/tools/zero_ab.py): no iteration code in the kernel at all -- what the prologue, the screen and
        screened = true;    // the read-out cost at other register budgets / occupancies (wrong plans for agents with a violated row)
#endif
        if (!screened) {  // uniform: the pivot scalings of the cached rows, now that they are needed
            w.each([&](int l) {
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    double iw = DQ_RSQRT(ciw[u].at(l));
                    DQ_KEEP(iw);
                    ciw[u].at(l) = iw;
                }
            });
        }
    }
    w.stamp(1);  // slot cache, x0, bounds
    DQ_PHASE_EXIT(1);

    // per-lane state of the method: lane i (+ 64 c) keeps active row i -- its dual, its row code and its sparse entries
    RegU fl_slo, fl_shi, fl_slk;  // "skip" bits of the cached slots and slack bounds: active or accepted-infeasible
    RegD uu[RC], rv[RC];
    RegI act[RC];
    RegU a_pk[RC];     // three control-point indices of 8 bits | axis << 24 | LINE << 26 (see the decode step)
    RegF a_cf[RC][3];  // POINT row: coefficient per axis, LINE row: per point (float32 normals and small integers: exact)
    RegI a_sl[RC];     // slack variable of the row (coefficient -1), or -1
    RegD a_rhs[RC];    // right-hand side of an active cached LSC row (its register in the row cache holds -NEVER meanwhile)
    constexpr uint32_t PK_NONE = 0;  // a POINT row at control point 0 with zero coefficients
    w.each([&](int l) {
        fl_slo.at(l) = 0; fl_shi.at(l) = 0; fl_slk.at(l) = 0;
#pragma unroll
        for (int c = 0; c < RC; c++) {
            uu[c].at(l) = 0; act[c].at(l) = -1; rv[c].at(l) = 0; a_rhs[c].at(l) = 0;
            a_pk[c].at(l) = PK_NONE; a_sl[c].at(l) = -1;
            a_cf[c][0].at(l) = 0; a_cf[c][1].at(l) = 0; a_cf[c][2].at(l) = 0;
        }
    });

    constexpr int QB = 8;  // block of the product rv = S d1 over the active rows' records
    constexpr int JB = 4;  // block of the step direction's loop over them
    constexpr bool DXREG = DIMS > 0 && SLK == 0 && MS * 6 <= 64;  // the step direction stays in registers (see there)
    constexpr bool P64 = DIMS > 0;  // P = 6 M <= 64 is known at compile time: lane p is control point p, one trip
    static_assert((64 % QB) == 0 && (64 % JB) == 0 && (QB % JB) == 0, "a block never runs past the last lane, nor across lay.qz");
    int q = 0, iters = 0, n_add = 0, n_drop = 0;
    const int max_iter = kp.max_iter > 0 ? kp.max_iter : 20 * (r + 8) + 200;
    // Row codes, code = unit * 8 + c (tag_score):
    //   units [0, UL0)   BYK: coordinate j of the control polygon; c = 2 kind + side, kind 0 bound / 1 velocity / 2 acceleration
    //   units [UL0, UX0) LSC row (c = 0)
    //   units [UX0, US0) structured slot of the generic code (c = side)
    //   units [US0, ...) slack variable: its bound eps <= 0 (c = 0)
    const int UL0 = BYK ? nxc : 0, UX0 = UL0 + K, US0 = UX0 + n_slots;

    // objective value of the current iterate x (cplex.getObjValue(), traj_optimizer.cpp:109, objective at :286-331)
    auto objective = [&]() -> double {
        RegD cj, cs2;
        w.each([&](int l) {
            double a1 = 0, a2 = 0;
            for (int g0 = 0; g0 < dim * M; g0 += 64) {  // one Bernstein segment (axis k, segment m) per lane
                const int sgm = g0 + l;
                const bool in = sgm < dim * M;
                const int sc = in ? sgm : 0;
                const int k = sc / M, m = sc - k * M;
                const double *seg = x + k * P + m * n1;
                double c6[n1];
#pragma unroll
                for (int c = 0; c < n1; c++) c6[c] = seg[c];
                double e1 = 0;
#pragma unroll
                for (int i = 0; i < n1; i++) {
                    double t = 0;
#pragma unroll
                    for (int c = 0; c < n1; c++) t = fma(kp.Qb[i * n1 + c], c6[c], t);
                    e1 = fma(c6[i], t, e1);
                }
                const double e = c6[n] - (double)pick3(goal, k);
                const double e1t = fma(kp.w_t * e, e, e1);
                e1 = m >= M - T ? e1t : e1;
                a1 += in ? e1 : 0.0;
            }
            for (int s0 = 0; s0 < nslk; s0 += 64) {
                const int s = s0 + l;
                const bool in = s < nslk;
                const int sc = in ? s : 0;
                const int m = sc % M;
                const double e = x[nxc + sc];
                double wv = kp.w_s * ((double)(M - m) / M) * e * e;
                DQ_KEEP(wv);
                a2 += in ? wv : 0.0;
            }
            cj.at(l) = a1;
            cs2.at(l) = a2;
        });
        double t1c, t2c;
        w.sum2(cj, cs2, t1c, t2c);
        return t1c + t2c;
    };

    // Upper bound of the objective over all points that satisfy the box rows (world / SFC, T3a, T3e) and the
    // acceleration rows (T3g); only meaningful without slack variables.
    //   jerk term: the acceleration of a segment is a cubic Bernstein polynomial whose control points are bounded by
    //   a = max(max_acc, |initial acceleration|), so |jerk| <= 3/dt * 2a and  c'Qb c = w_c * integral jerk^2 <= w_c 36 a^2 / dt
    //   per axis and segment;  terminal term: w_t * (largest distance to the goal of a point of that segment's box that the
    //   velocity rows allow the segment's end point to reach)^2.
    auto objective_upper_bound = [&]() -> double {
        double amax = max_acc;
        for (int k = 0; k < 3; k++) amax = fabs((double)acc[k]) > amax ? fabs((double)acc[k]) : amax;
        const double jerk = kp.w_c * 36.0 * amax * amax / kp.dt * (double)(dim * M);
        RegD tj, zero;
        w.each([&](int l) {
            double a1 = 0;
            for (int g0 = 0; g0 < dim * M; g0 += 64) {
                const int sgm = g0 + l;
                const bool in = sgm < dim * M;
                const int sc = in ? sgm : 0;
                const int k = sc / M, m = sc - k * M;
                const double g = (double)pick3(goal, k);
                const double dl = fabs(blo[sc] - g), dh = fabs(bhi[sc] - g);
                const double dbox = dl > dh ? dl : dh;
                // the end point of segment m is p((m+1) dt) = pos + integral of the velocity, whose control points are
                // bounded by max_vel except the two that the initial state fixes (v0 = vel, v1 = vel + acc dt / (n-1))
                const double v0 = fabs((double)pick3(vel, k));
                const double v1 = fabs((double)pick3(vel, k) + (double)pick3(acc, k) * kp.dt / (double)(n - 1));
                double vb = max_vel > v0 ? max_vel : v0;
                vb = vb > v1 ? vb : v1;
                const double dreach = fabs((double)pick3(pos, k) - g) + vb * (double)(m + 1) * kp.dt;
                const double dm2 = dbox < dreach ? dbox : dreach;
                double tv = kp.w_t * dm2 * dm2;
                DQ_KEEP(tv);
                a1 += (in & (m >= M - T)) ? tv : 0.0;
            }
            tj.at(l) = a1;
            zero.at(l) = 0.0;
        });
        double term, unused;
        w.sum2(tj, zero, term, unused);
        return jerk + term;
    };
    int n_price = 0;
    double f_max = -1.0;  // computed when first needed
#ifdef DLSCQP_TRACE
    double trace_min_ratio = 1.0;
#endif
    // the largest |N'a|^2 / |z|^2 a row was added with so far (1 / the smallest pivot ratio): what the rounding noise of every later
    // a.dx scales with.  After a pivot of 1e-12 |N'a|^2 -- a row that is independent of the active ones in exact arithmetic and by both
    // tests -- the active rows' records are numerically singular and a.dx of a DEPENDENT row came out at 0.12 |N'a|^2
    // (tests/tools/margins_long.py, seed 80048: an infeasible agent with slack variables on its way to the iteration limit); from
    // such a pivot on every row is measured in the whitened space (DEP_PIVOT below)
    // (kept as the threshold itself, raised where a row is added: the iteration compares with one number as before.  KParams: the
    // narrow constant only where it was calibrated)
    double suspect_now = DXREG ? kp.dep_suspect : DEP_SUSPECT_SLACK;

    // Takes a row out of the pricing passes (on: it became active, or was accepted within CPLEX's tolerance) or puts it back
    // (off: it left the active set; `rhs` = its right-hand side if it is a cached LSC row).  code and on are uniform, the
    // owner lane is picked by a select.
    auto set_skip = [&](int code, bool on, double rhs) {
        auto upd = [&](RegU &fr, int owner, int u) {
            w.each([&](int l) {
                const uint32_t cur = fr.at(l), bit = 1u << u;
                const uint32_t nv = on ? (cur | bit) : (cur & ~bit);
                fr.at(l) = l == owner ? nv : cur;
            });
        };
        const int unit = code >> 3, cc = code & 7;
        if (unit < UL0) {  // BYK: the limit register of the coordinate's row
            const int own = unit & 63, uo = unit >> 6, kind = cc >> 1, side = cc & 1;
            const int k = unit / P, m = (unit - k * P) / n1;
            double lim;  // what the register goes back to
            if (kind == 0) lim = side ? bhi[k * M + m] + VIOL_TOL : blo[k * M + m] - VIOL_TOL;  // (limit -/+ tolerance, as the registers hold it)
            else lim = (kind == 1 ? hv_lim : ha_lim) + VIOL_TOL;
            lim = w.uni(lim);
            const double off = (kind == 0 && side == 0) ? -NEVER : NEVER;
            const double nv = on ? off : lim;
            if (LEAN && kind != 0) {  // uniform: the coordinate's bit
                const uint32_t bit = 1u << (2 * uo + (kind - 1));
                w.each([&](int l) {
                    const uint32_t cur = lim_off.at(l);
                    lim_off.at(l) = l == own ? (on ? (cur | bit) : (cur & ~bit)) : cur;
                });
            } else {
                w.each([&](int l) {
#pragma unroll
                    for (int u = 0; u < NXP; u++) {
                        const bool me = (l == own) & (u == uo);
                        blo_r[u].at(l) = (me & (kind == 0) & (side == 0)) ? nv : blo_r[u].at(l);
                        bhi_r[u].at(l) = (me & (kind == 0) & (side == 1)) ? nv : bhi_r[u].at(l);
                        if (!LEAN) {
                            hv_r[u].at(l) = (me & (kind == 1)) ? nv : hv_r[u].at(l);
                            ha_r[u].at(l) = (me & (kind == 2)) ? nv : ha_r[u].at(l);
                        }
                    }
                });
            }
        } else if (unit < UL0 + Kc) {
            const int row = unit - UL0, own = row & 63, uo = row >> 6;
            const double nv = on ? -NEVER : rhs;
            w.each([&](int l) {
#pragma unroll
                for (int u = 0; u < KU; u++) cd[u].at(l) = ((l == own) & (u == uo)) ? nv : cd[u].at(l);
            });
        } else if (unit < UL0 + Kl) {  // a row in LDS: its right-hand side, like a cached row's register
            const double nv = on ? -NEVER : rhs;
            w.each([&](int) { lrhs[unit - UL0 - Kc] = nv; });  // every lane, same word, same value
        } else if (unit < UX0) {
            // (one bit per row: every lane reads the same byte and writes the same value back)
            const int fi = unit - UL0 - Kl;
            w.each([&](int) {
                const unsigned char cur = row_flag[fi >> 3], bit = (unsigned char)(1u << (fi & 7));
                row_flag[fi >> 3] = on ? (unsigned char)(cur | bit) : (unsigned char)(cur & ~bit);
            });
        } else if (unit < US0) {
            const int s = unit - UX0, side = cc & 1;
            if (ALL_CACHED || s < Sc) {
                if (side) upd(fl_shi, s & 63, s >> 6);
                else upd(fl_slo, s & 63, s >> 6);
            } else {
                w.each([&](int) { cold_flag[2 * (s - Sc) + side] = on ? 1 : 0; });
            }
        } else {
            const int s = unit - US0;
            upd(fl_slk, s & 63, s >> 6);
        }
        w.sync();
    };

    ColdRows cold;
    cold.nx = Rnx; cold.ny = Rny; cold.nz = Rnz; cold.px = Rpx; cold.py = Rpy; cold.pz = Rpz; cold.d = Rd; cold.key = Rkey;
    cold.flag = row_flag; cold.x = x; cold.tNN = tNN; cold.sigma = kp.sigma;
    cold.K = K; cold.Kc = Kl; cold.P = P; cold.M = M; cold.dim = dim; cold.nxc = nxc; cold.nso = nso; cold.phi = kp.phi; cold.unit0 = UL0;
    // this wavefront's share of the rows beyond the register cache [Kc, K) (all of them unless helper wavefronts price the rest):
    // its part in LDS [lds_jb, lds_je), its part in HBM [cold_jb, cold_je)
    int share_jb, share_je;
    cold_rows_share(Kc, K, 0, w.nw, share_jb, share_je);
    const int lds_jb = share_jb < Kl ? share_jb : Kl, lds_je = share_je < Kl ? share_je : Kl;
    const int cold_jb = share_jb > Kl ? share_jb : Kl, cold_je = share_je > Kl ? share_je : Kl;
    LdsRows ldsr;
    ldsr.rhs = lrhs; ldsr.iw = liw; ldsr.nx = lnx; ldsr.ny = lny; ldsr.nz = lnz; ldsr.key = lkey; ldsr.x = x;
    ldsr.Kc = Kc; ldsr.P = P; ldsr.M = M; ldsr.dim = dim; ldsr.nxc = nxc; ldsr.unit0 = UL0;

    // the path x has gone, in the metric of the Hessian and rounded up; per group of rows beyond the register cache the length at which
    // the group has to be priced again (see the pricing pass)
    double path = 0.0;
    RegD expiry;
    if (!ALL_CACHED) w.each([&](int l) { expiry.at(l) = -1e300; });
    auto expiry_of = [&](double at, double least) { return least > 0 ? at + least * (1.0 - 1e-6) - 1e-7 : -1e300; };
    for (; !(SCREEN && screened);) {  // (a screened agent has nothing to do here: straight to the read-out)
        // ---- most violated inactive row ---------------------------------------------------------
        // Every candidate row: violation v = limit - value (or value - limit), pivot score (v - VIOL_TOL) / |N'a| tagged with
        // the row's code, running maximum.  Rows that are not in the model, or are active, have a limit that is never
        // violated; all LDS reads of a part come first and unconditionally (one exposed round trip per part).
        RegD best;
        if (!ALL_CACHED && w.nw > 1) w.pass_begin(T);  // the helper wavefronts start on their shares of the cold rows
        w.each([&](int l) {
            double xr[KU][4];
            if (LEAN) {  // (the conversions of the normals to double belong to this pass: not in front of the loop, in twelve registers)
#pragma unroll
                for (int u = 0; u < KU; u++) { DQ_KEEP(cnx[u].at(l)); DQ_KEEP(cny[u].at(l)); DQ_KEEP(cnz[u].at(l)); }
            }
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int p = cpo[u].at(l);
                xr[u][0] = x[p];
                xr[u][1] = x[P + p];
                xr[u][2] = dim == 3 ? x[2 * P + p] : 0.0;
                if (SLK != 0) {
                    const uint32_t key = ckey[u].at(l);
                    const uint32_t sl = key >> 16;
                    const bool has_slack = key != KEY_SKIP && sl != 0xFFFFu;
                    const double xs = x[nxc + (has_slack ? (int)sl * M + (int)(key & 0xFF) : 0)];
                    xr[u][3] = has_slack ? xs : 0.0;
                } else {
                    xr[u][3] = 0.0;
                }
            }
            // BYK: x_j, x_{j+1}, x_{j+2} of the lane's coordinates (the neighbours that do not exist meet a limit that is never
            // violated)
            double c0[NXP], c1[NXP], c2[NXP], kw[NXP][3];
            if (BYK) {
#pragma unroll
                for (int u = 0; u < NXP; u++) {
                    const int j = 64 * u + l;
                    const int j0 = j < nxc ? j : nxc - 1, j1 = j + 1 < nxc ? j + 1 : nxc - 1, j2 = j + 2 < nxc ? j + 2 : nxc - 1;
                    c0[u] = x[j0]; c1[u] = x[j1]; c2[u] = x[j2];
                    if (LEAN) {
#pragma unroll
                        for (int t = 0; t < 3; t++) kw[u][t] = base[lay.oKiw + t * nxc + j0];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < KU; u++) { DQ_KEEP(xr[u][0]); DQ_KEEP(xr[u][1]); DQ_KEEP(xr[u][2]); DQ_KEEP(xr[u][3]); }
            if (BYK) {
#pragma unroll
                for (int u = 0; u < NXP; u++) { DQ_KEEP(c0[u]); DQ_KEEP(c1[u]); DQ_KEEP(c2[u]); }
                if (LEAN) {
#pragma unroll
                    for (int u = 0; u < NXP; u++) { DQ_KEEP(kw[u][0]); DQ_KEEP(kw[u][1]); DQ_KEEP(kw[u][2]); }
                }
            }
            double bs = 0.0;
#pragma unroll
            for (int u = 0; u < KU; u++) {  // cached LSC rows
                double val = (double)cnx[u].at(l) * xr[u][0];
                val = fma((double)cny[u].at(l), xr[u][1], val);
                if (dim == 3) val = fma((double)cnz[u].at(l), xr[u][2], val);
                if (SLK != 0) val -= xr[u][3];
                const double v = (cd[u].at(l) - VIOL_TOL) - val;
                bs = DQ_SCORE_MAX(bs, tag_score(v * ciw[u].at(l), LEAN ? (uint32_t)((BYK ? nxc : 0) + l + 64 * u) << 3 : ltag[u].at(l)));
            }
            if (BYK) {
                // the six rows of each of the lane's coordinates
#pragma unroll
                for (int u = 0; u < NXP; u++) {
                    const double dv = c1[u] - c0[u], da = (c2[u] - c1[u]) - dv;
                    const double wb = LEAN ? kw[u][0] : kiw[u][0].at(l), wv = LEAN ? kw[u][1] : kiw[u][1].at(l), wa = LEAN ? kw[u][2] : kiw[u][2].at(l);
                    const double hv = LEAN ? lim_v(lim_off.at(l), u) : hv_r[u].at(l), ha = LEAN ? lim_a(lim_off.at(l), u) : ha_r[u].at(l);  // (limit + tolerance)
                    double sb = DQ_SCORE_MAX(tag_score((blo_r[u].at(l) - c0[u]) * wb, 0u),
                                     tag_score((c0[u] - bhi_r[u].at(l)) * wb, 1u));
                    sb = DQ_SCORE_MAX(sb, tag_score((-hv - dv) * wv, 2u));
                    sb = DQ_SCORE_MAX(sb, tag_score((dv - hv) * wv, 3u));
                    sb = DQ_SCORE_MAX(sb, tag_score((-ha - da) * wa, 4u));
                    sb = DQ_SCORE_MAX(sb, tag_score((da - ha) * wa, 5u));
                    // the coordinate's number above the row's (tag_score keeps the three low bits of the lane's winner)
                    uint64_t b;
                    memcpy(&b, &sb, 8);
                    b |= (uint64_t)(LEAN ? (uint32_t)(64 * u + l < nxc ? 64 * u + l : nxc - 1) << 3 : jtag[u].at(l));
                    memcpy(&sb, &b, 8);
                    bs = DQ_SCORE_MAX(bs, sb);
#ifdef DLSCQP_DEBUG_SCAN
                    if (64 * u + l == 76 || 64 * u + l == 77) printf("coord %d c0 %.6f lo %.6g hi %.6g wb %.6g hv %.4g ha %.4g sb %.6g code %u\n", 64 * u + l, c0[u], blo_r[u].at(l), bhi_r[u].at(l), wb, hv_r[u].at(l), ha_r[u].at(l), sb, score_code(sb)), printf("   dv %g da %g wv %g wa %g c1 %g c2 %g\n", dv, da, wv, wa, c1[u], c2[u]);
#endif
                }
            }
            // structured slots of the generic code (BYK: the rows of the communication range only)
            {
                double sx[SUA][5];
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const uint32_t pi = sidx[u].at(l), pc = scb[u].at(l);
                    int i0 = pi & 0x3FF, i1 = (pi >> 10) & 0x3FF, i2 = (pi >> 20) & 0x3FF;
                    i0 = i0 == 0x3FF ? 0 : i0;  // unused entries carry coefficient 0
                    i1 = i1 == 0x3FF ? 0 : i1;
                    i2 = i2 == 0x3FF ? 0 : i2;
                    const int bsel = pc >> 24;
                    sx[u][0] = x[i0]; sx[u][1] = x[i1]; sx[u][2] = x[i2];
                    sx[u][3] = blo[bsel]; sx[u][4] = bhi[bsel];
                }
#pragma unroll
                for (int u = 0; u < SU; u++) { DQ_KEEP(sx[u][0]); DQ_KEEP(sx[u][1]); DQ_KEEP(sx[u][2]); DQ_KEEP(sx[u][3]); DQ_KEEP(sx[u][4]); }
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const uint32_t pi = sidx[u].at(l), pc = scb[u].at(l);
                    const bool valid = !(pi & 0x80000000u);
                    double val = (double)(int8_t)(pc & 0xFF) * sx[u][0];
                    val = fma((double)(int8_t)((pc >> 8) & 0xFF), sx[u][1], val);
                    val = fma((double)(int8_t)((pc >> 16) & 0xFF), sx[u][2], val);
                    const double vlo = (valid && !((fl_slo.at(l) >> u) & 1)) ? (sx[u][3] - VIOL_TOL) - val : -1.0;
                    const double vhi = (valid && !((fl_shi.at(l) >> u) & 1)) ? val - (sx[u][4] + VIOL_TOL) : -1.0;
                    const uint32_t code = (uint32_t)(UX0 + l + 64 * u) << 3;
                    bs = DQ_SCORE_MAX(bs, tag_score(vlo * siw[u].at(l), code));
                    bs = DQ_SCORE_MAX(bs, tag_score(vhi * siw[u].at(l), code | 1u));
                }
            }
            for (int s0 = 0; s0 < nslk; s0 += 64) {  // eps <= 0 (traj_optimizer.cpp:276); uniform trip count
                const int s = s0 + l;
                const bool in = s < nslk;
                const int sc2 = in ? s : 0;
                double v = x[nxc + sc2];
                double sg = kp.sigma[sc2 % M];
                DQ_KEEP(v); DQ_KEEP(sg);
                const bool cand = in & (sg > 0) & !((fl_slk.at(l) >> (s0 >> 6)) & 1);
                double qv = (v - VIOL_TOL) / (sg > 0 ? sg : 1.0);
                DQ_KEEP(qv);
                bs = DQ_SCORE_MAX(bs, tag_score(cand ? qv : -1.0, (uint32_t)(US0 + sc2) << 3));
            }
            if (!ALL_CACHED)
            for (int s0 = Sc; s0 < n_slots; s0 += 64) {
                const int s = s0 + l;
                const bool in = s < n_slots;
                const int sc2 = in ? s : n_slots - 1;
                const Slot sl = slots[sc2];
                double val = 0;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const bool used = sl.idx[t] >= 0;
                    double xv = x[used ? sl.idx[t] : 0];
                    DQ_KEEP(xv);
                    val = used ? val + (double)sl.coef[t] * xv : val;
                }
                const double vlo = (blo[sl.bsel] - VIOL_TOL) - val, vhi = val - (bhi[sl.bsel] + VIOL_TOL);
                const double iw = tSIW[sc2];
                const int fo = 2 * (sc2 - Sc);
                int f_lo = cold_flag[fo], f_hi = cold_flag[fo + 1];
                DQ_KEEP(f_lo); DQ_KEEP(f_hi);
                const uint32_t code = (uint32_t)(UX0 + sc2) << 3;
                bs = DQ_SCORE_MAX(bs, tag_score((in & (f_lo == 0)) ? vlo * iw : -1.0, code));
                bs = DQ_SCORE_MAX(bs, tag_score((in & (f_hi == 0)) ? vhi * iw : -1.0, code | 1u));
            }
            best.at(l) = bs;
        });
        if (!ALL_CACHED) {
            // Rows beyond the register cache: this wavefront's share of them (all of them unless helper wavefronts price the rest:
            // W::nw > 1), in groups of one trip each -- 128 rows in LDS, 64 CB rows in HBM.  A group is LEFT OUT while none of its
            // rows can be violated: the scaled score (limit - a.x) / |N'a| of a row changes by at most |dx|_H = t sqrt(a.dx) in a
            // step x += t dx (Cauchy-Schwarz in the metric of the Hessian; a.dx = dx' H dx for the projected direction), so a group
            // whose smallest scaled slack was s when the path of x was p long has no violated row before the path reaches p + s.
            // Lane g of `expiry` keeps that length for group g (groups from 64 on are priced in every pass); a row that comes
            // back into the model (a drop) resets all of them.  Same winner as the full pass: a group that is left out holds no
            // candidate.  (In a crowd the heavy agents price 36 % of their groups: tests/tools/crowd_skip_study.py.)
            const int n_lt = (lds_je - lds_jb + 127) / 128;
            for (int g = 0; g < n_lt; g++) {  // rows in LDS
                if (g < 64 && path < w.bcast(expiry, g)) continue;  // uniform
                const int j0 = lds_jb + 128 * g, j1 = j0 + 128 < lds_je ? j0 + 128 : lds_je;
                RegD ls;
                w.each([&](int l) {
                    double least = 1e300;
                    best.at(l) = lds_rows_lane_best<SLK != 0>(ldsr, l, j0, j1, best.at(l), least);
                    ls.at(l) = least;
                });
                const double e = expiry_of(path, w.min(ls));
                w.each([&](int l) { expiry.at(l) = l == g ? e : expiry.at(l); });
            }
            for (int g = n_lt, j0 = cold_jb; j0 < cold_je; g++, j0 += 64 * CB) {  // rows in HBM
                if (g < 64 && path < w.bcast(expiry, g)) continue;  // uniform
                const int j1 = j0 + 64 * CB < cold_je ? j0 + 64 * CB : cold_je;
                RegD ls;
                w.each([&](int l) {
                    double least = 1e300;
                    best.at(l) = cold_rows_lane_best<CB>(cold, l, j0, j1, best.at(l), least);
                    ls.at(l) = least;
                });
                const double e = expiry_of(path, w.min(ls));
                w.each([&](int l) { expiry.at(l) = l == g ? e : expiry.at(l); });
            }
        }
        double top = w.max_pos(best);  // best pivot score over all violated rows, with the row's code
        if (!ALL_CACHED && w.nw > 1) top = w.pass_end(top);  // ... the helpers' included
        w.stamp(2);                       // row scan + arg-max
        DQ_PHASE_EXIT(2);
        if (!(top >= SCORE_MIN)) break;   // no row violated by more than VIOL_TOL: optimal
        // ---- infeasibility by weak duality ------------------------------------------------------
        // x minimises the objective over the active rows with non-negative multipliers, so f(x) is a lower bound
        // of the optimum.  Every feasible point obeys the box and acceleration rows, which bound the objective
        // from above by f_max (objective_upper_bound).  f(x) > f_max therefore proves the QP infeasible, usually
        // after 5-20 rows instead of the ~r rows the active set needs to run out of dimensions (the launch is as
        // slow as its slowest agent, and those were the infeasible ones).  Checked every second pricing pass from the
        // eighth on.
        n_price++;
        // A launch is as long as its slowest agent, and with three wavefronts on a SIMD every one of them is slower than with two: an agent
        // that keeps iterating raises its wavefront's issue priority (the SIMD's arbiter serves the highest ready priority first) -- the
        // few agents on the launch's critical path run at the pace they had alone, the many that never iterate give way for a moment.
        if (DQ_PRIO_STEPS) {
            if (n_price == DQ_PRIO_AT1) DQ_SETPRIO(1);
            else if (n_price == DQ_PRIO_AT2) DQ_SETPRIO(2);
            else if (n_price == DQ_PRIO_AT3) DQ_SETPRIO(3);
        }
        if (kp.wd_cut && !out_conflict && nslk == 0 && n_price >= 8 && (n_price & 1) == 0) {
            if (n_price == 8) f_max = objective_upper_bound();  // first check of this agent
            const double f_now = objective();
            DQ_TRACE(4, f_now, f_max, (double)n_price, (double)q);
            if (f_now > f_max + 1.0) {  // margin 1 (about 1 % of f_max with the reference's weights); f(x) grows by orders of magnitude
                status = DLSCQP_INFEASIBLE;
                break;
            }
        }
        const int ip = (int)score_code(top);
#ifdef DLSCQP_TRACE
        DQ_TRACE(3, objective(), 0.0, 0.0, 0.0);
#endif

        // ---- decode the row into sparse x-space entries and evaluate its violation (uniform) ----------------------
        // Two shapes of rows.  POINT rows touch one control point on up to three axes (LSC rows, box faces, waypoint range, and
        // the rows of a slack variable alone): point p, one coefficient per axis.  LINE rows touch up to three control points of
        // one axis (velocity, acceleration, communication range): axis k, one coefficient per point.  Packed (also the form lane i
        // keeps active row i in): three point indices of 8 bits | axis << 24 | LINE << 26.
        int sidx_row = -1;
        double row_scale = 1.0, row_rhs = 0.0;
        bool new_line = false;
        int new_axis = 0, np0 = 0, np1 = 0, np2 = 0;
        double ncf[3] = {0.0, 0.0, 0.0};  // POINT: coefficient of axis k; LINE: coefficient of point t
        double viol;                      // limit - value of the row (> VIOL_TOL)
        const int unit = ip >> 3, icand = ip & 7;
        if (unit < UL0) {
            // BYK: row `icand` of coordinate `unit`
            const int kind = icand >> 1, side = icand & 1;
            const int k = unit / P, p = unit - k * P, m = p / n1;
            const int j1 = unit + 1 < nxc ? unit + 1 : nxc - 1, j2 = unit + 2 < nxc ? unit + 2 : nxc - 1;
            const double c0 = x[unit], c1 = x[j1], c2 = x[j2], lo = blo[k * M + m], hi = bhi[k * M + m];
            const double sg = side == 0 ? 1.0 : -1.0;  // val >= lo   |   -val >= -hi
            np0 = p;
            double val, lim_lo, lim_hi;
            if (kind == 0) {
                np1 = p; np2 = p;
                ncf[0] = k == 0 ? sg : 0.0; ncf[1] = k == 1 ? sg : 0.0; ncf[2] = k == 2 ? sg : 0.0;
                val = c0; lim_lo = lo; lim_hi = hi;
            } else if (kind == 1) {
                new_line = true; new_axis = k; np1 = p + 1; np2 = p;
                ncf[0] = -sg; ncf[1] = sg;
                val = c1 - c0; lim_lo = -hv_lim; lim_hi = hv_lim;
                row_scale = 1.0 / kp.vel_scale;  // back to the reference's row scaling: n/dt
            } else {
                new_line = true; new_axis = k; np1 = p + 1; np2 = p + 2;
                ncf[0] = sg; ncf[1] = -2.0 * sg; ncf[2] = sg;
                val = (c2 - c1) - (c1 - c0); lim_lo = -ha_lim; lim_hi = ha_lim;
                row_scale = 1.0 / kp.acc_scale;  // n(n-1)/dt^2
            }
            viol = w.uni(side == 0 ? lim_lo - val : val - lim_hi);
        } else if (unit < UX0) {
            const int row = unit - UL0;
            uint32_t key = 0;
            float nn[3] = {0, 0, 0};
            if (ALL_CACHED || row < Kc) {
                const int own = row & 63, u = row >> 6;
                double rh = 0.0;
#pragma unroll
                for (int uq = 0; uq < KU; uq++) {
                    if (uq == u) {  // uniform: the register set of the row, broadcast from its owner lane
                        key = w.bcast(ckey[uq], own);
                        nn[0] = w.bcast(cnx[uq], own); nn[1] = w.bcast(cny[uq], own); nn[2] = w.bcast(cnz[uq], own);
                        rh = w.bcast(cd[uq], own);
                    }
                }
                row_rhs = rh;
            } else if (row < Kl) {  // a row in LDS (every lane reads the same words)
                key = lkey[row - Kc];
                nn[0] = lnx[row - Kc]; nn[1] = lny[row - Kc]; nn[2] = lnz[row - Kc];
                row_rhs = w.uni(lrhs[row - Kc]);
            } else {
                key = Rkey[row];
                nn[0] = Rnx[row]; nn[1] = Rny[row]; nn[2] = Rnz[row];
                uint32_t sl = key >> 16;
                if (sl != 0xFFFFu && (int)sl >= nso) key |= 0xFFFF0000u;
                row_rhs = Rd[row] + (double)nn[0] * (double)Rpx[row] + (double)nn[1] * (double)Rpy[row] +
                          (dim == 3 ? (double)nn[2] * (double)Rpz[row] : 0.0);
            }
            const int m = key & 0xFF, i = (key >> 8) & 0xF;
            np0 = np1 = np2 = m * n1 + i;
            ncf[0] = (double)nn[0]; ncf[1] = (double)nn[1]; ncf[2] = dim == 3 ? (double)nn[2] : 0.0;
            const uint32_t sl = key >> 16;
            if (sl != 0xFFFFu) sidx_row = (int)sl * M + m;
            double val = ncf[0] * x[np0] + ncf[1] * x[P + np0] + (dim == 3 ? ncf[2] * x[2 * P + np0] : 0.0);
            val -= sidx_row >= 0 ? x[nxc + sidx_row] : 0.0;
            viol = w.uni(row_rhs - val);
        } else if (unit < US0) {
            const int s = unit - UX0, side = icand & 1;
            int sidx3[3], scoef3[3], bs;
            if (ALL_CACHED || s < Sc) {  // from the owner lane's registers
                const int own = s & 63, u = s >> 6;
                uint32_t pi = 0, pc = 0;
#pragma unroll
                for (int uq = 0; uq < SU; uq++) {
                    const uint32_t a0 = w.bcast(sidx[uq], own), a1 = w.bcast(scb[uq], own);
                    pi = uq == u ? a0 : pi;
                    pc = uq == u ? a1 : pc;
                }
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    sidx3[t] = (pi >> (10 * t)) & 0x3FF;
                    if (sidx3[t] == 0x3FF) sidx3[t] = -1;
                    scoef3[t] = (int8_t)((pc >> (8 * t)) & 0xFF);
                }
                bs = pc >> 24;
            } else {
                const Slot sl = slots[s];  // uniform load of 16 B (L2-resident table)
#pragma unroll
                for (int t = 0; t < 3; t++) { sidx3[t] = sl.idx[t]; scoef3[t] = sl.coef[t]; }
                bs = sl.bsel;
            }
            const double sg = side == 0 ? 1.0 : -1.0;  // val >= lo   |   -val >= -hi
            int ek[3], ep[3];
            double cf[3], val = 0.0;
#pragma unroll
            for (int t = 0; t < 3; t++) {  // fixed positions, unused entries keep ek = -1 / cf = 0
                const bool used = sidx3[t] >= 0;
                const int k = used ? sidx3[t] / P : -1;
                ek[t] = k;
                ep[t] = used ? sidx3[t] - k * P : 0;
                cf[t] = used ? sg * (double)scoef3[t] : 0.0;
                val += used ? (double)scoef3[t] * x[sidx3[t]] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                new_line = new_line || (ek[t] >= 0 && ep[t] != ep[0]);
                new_axis = ek[t] >= 0 ? ek[t] : new_axis;  // LINE rows: all used entries share it (qp_tables.cpp)
            }
            if (new_line) {
                ncf[0] = cf[0]; ncf[1] = cf[1]; ncf[2] = cf[2];
                np0 = ep[0]; np1 = ep[1]; np2 = ep[2];
            } else {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    ncf[0] = ek[t] == 0 ? cf[t] : ncf[0];
                    ncf[1] = ek[t] == 1 ? cf[t] : ncf[1];
                    ncf[2] = ek[t] == 2 ? cf[t] : ncf[2];
                }
                np0 = np1 = np2 = ep[0];
            }
            viol = w.uni(side == 0 ? blo[bs] - val : val - bhi[bs]);
            // back to the reference's row scaling: n/dt for velocity rows, n(n-1)/dt^2 for acceleration rows
            row_scale = bs == dim * M ? 1.0 / kp.vel_scale : (bs == dim * M + 1 ? 1.0 / kp.acc_scale : 1.0);
        } else {
            sidx_row = unit - US0;
            viol = w.uni(x[nxc + sidx_row]);
        }
        const uint32_t pk_new = new_line ? ((uint32_t)np0 | ((uint32_t)np1 << 8) | ((uint32_t)np2 << 16) | ((uint32_t)new_axis << 24) | (1u << 26))
                                         : (uint32_t)np0 * 0x010101u;
        DQ_TRACE(1, (double)ip, viol, top, (double)q);
        double sig2_row = 0.0;  // (a build without slack variables has none: no dependent scalar load in front of the rows of G)
        if (SLK != 0) {
            const double sig_row = kp.sigma[(sidx_row < 0 ? 0 : sidx_row) % M];
            sig2_row = sidx_row >= 0 ? sig_row * sig_row : 0.0;
        }
        // ---- the rows of G_T at the control points the new row touches (G a: lane p holds control point p) --------------------
        // requested HERE and used after d1 and rv below: their trip to L2 runs under those two.  The Gram table is never copied:
        // an iteration needs one row of it (POINT row) or three (LINE row), 8 P bytes each, coalesced.
        RegD gq[3];
        if (P64) {
            w.each([&](int l) {
                const int pc = l < P ? l : P - 1;
                // (three loads whatever the row: np1 = np2 = np0 for a POINT row, whose second and third read hit the line the first
                // one brings.  Copies of the first register instead made the wavefront wait for it HERE, a whole trip to L2, where the
                // point of asking early is that d1 and rv run under that trip -- found in the ISA in round 6.)
                gq[0].at(l) = G[np0 * ldg + pc];
                gq[1].at(l) = G[np1 * ldg + pc];
                gq[2].at(l) = G[np2 * ldg + pc];
            });
        }
        // where the new row's entries sit in a Z column (x-space index = axis P + control point), as z_squared() has them: a POINT
        // row in 2-D has a zero third coefficient and no third axis
        const int ze0 = (new_line ? new_axis : 0) * P + np0;
        const int ze1 = (new_line ? new_axis : (dim > 1 ? 1 : 0)) * P + np1;
        const int ze2 = (new_line ? new_axis : (dim > 2 ? 2 : 0)) * P + np2;
        const double zc2 = (new_line || dim > 2) ? ncf[2] : 0.0;
        const int zes = nxc + (sidx_row >= 0 ? sidx_row : 0);
        w.stamp(3);  // decode

        double uq = 0.0;
        double dall = 0.0;  // squared whitened length of the row, a' G a (set in the loop, behind the reads of G)
        bool resolved = false;
        while (!resolved) {
            if (++iters > max_iter) { status = DLSCQP_ITERATION_LIMIT; break; }
            const int q_in = q < qz ? q : qz;  // active rows whose records are in the working set (the others: spill record)
            // ---- d1 = Z' a, rv = S d1 = (A G A')^-1 A G a ---------------------------------------------------------------------
            // Z = G A' S holds, column by column, the active rows' normals made orthonormal in the whitened space and mapped back
            // to control-point space (Z' G^-1 Z = I on the range of G): lane j reads ITS column at the new row's three entries --
            // no decode of the active rows, no gather from the Gram table, no product with S'.  (nd1 = -d1; zero beyond the
            // active rows.  Nothing to do while no row is active: rv keeps its zeros.)
            RegD nd1[RC];
            w.each([&](int l) {
#pragma unroll
                for (int c = 0; c < RC; c++) nd1[c].at(l) = 0.0;
            });
            if (q > 0) {  // uniform
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int j = l + 64 * c;
                        const bool on = j < q;
                        const double *zj = colw + ((on & (j < qz)) ? j : 0) * cs + zo;  // (a lane without a row reads record 0 and drops it)
                        double z0 = zj[ze0], z1 = zj[ze1], z2 = zj[ze2], zs = SLK != 0 ? zj[zes] : 0.0;
                        DQ_KEEP(z0); DQ_KEEP(z1); DQ_KEEP(z2);
                        if (SLK != 0) DQ_KEEP(zs);
                        if (q > qz) {  // uniform: some records live in the spill record
                            const bool sp = on & (j >= qz);
                            const double *zg = colsp + (size_t)blend_idx(sp, j - qz, 0) * cs + zo;  // (mask arithmetic: as ?: this came out as an EXEC-masked branch)
                            double y0 = zg[ze0], y1 = zg[ze1], y2 = zg[ze2], ys = SLK != 0 ? zg[zes] : 0.0;
                            DQ_KEEP(y0); DQ_KEEP(y1); DQ_KEEP(y2);
                            if (SLK != 0) DQ_KEEP(ys);
                            z0 = sp ? y0 : z0; z1 = sp ? y1 : z1; z2 = sp ? y2 : z2; zs = sp ? ys : zs;
                        }
                        double sd = fma(zc2, z2, fma(ncf[1], z1, ncf[0] * z0));
                        sd -= (SLK != 0 && sidx_row >= 0) ? zs : 0.0;  // the slack variable's coefficient is -1
                        nd1[c].at(l) = on ? -sd : 0.0;
                    }
                });
                w.stamp(4);  // d1
                // rv = S d1: lane i accumulates row i of S over the records, in blocks of QB with every read of a block issued before
                // its first use (S is zero below its diagonal: stored so; a block's records beyond the active rows meet a zero of nd1)
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        const int ir = i < r_max ? i : r_max - 1;
                        double a0 = 0.0, a1 = 0.0;
                        for (int k0 = 0; k0 < q_in; k0 += QB) {
                            double sv[QB];
#pragma unroll
                            for (int u = 0; u < QB; u++) sv[u] = colw[(k0 + u < q_in ? k0 + u : q_in - 1) * cs + ir];  // uniform: a record that exists
#pragma unroll
                            for (int u = 0; u < QB; u++) DQ_KEEP(sv[u]);
#pragma unroll
                            for (int u = 0; u < QB; u += 2) {
                                a0 = fma(sv[u], DQ_BC(nd1, k0 + u), a0);
                                a1 = fma(sv[u + 1], DQ_BC(nd1, k0 + u + 1), a1);
                            }
                        }
                        for (int k0 = qz; k0 < q; k0 += QB) {  // records in the spill record (rare: more than qz active rows)
                            double sv[QB];
#pragma unroll
                            for (int u = 0; u < QB; u++) sv[u] = colsp[(size_t)((k0 + u < q ? k0 + u : q - 1) - qz) * cs + ir];
#pragma unroll
                            for (int u = 0; u < QB; u++) DQ_KEEP(sv[u]);
#pragma unroll
                            for (int u = 0; u < QB; u += 2) {
                                a0 = fma(sv[u], DQ_BC(nd1, k0 + u), a0);
                                a1 = fma(sv[u + 1], DQ_BC(nd1, k0 + u + 1), a1);
                            }
                        }
                        rv[c].at(l) = i < q ? -(a0 + a1) : 0.0;
                    }
                });
                w.stamp(5);  // rv = S d1
            }
            // squared whitened length of the row, a' G a: the entries of G at the row's own control points
            // (the rows of G through an opaque copy made HERE: everything computed from them is invariant in this loop, and
            // hoisted in front of it -- as the compiler did -- it waits for their trip to L2 before d1 and rv instead of after)
            RegD gk[3];
            if (P64) {
                w.each([&](int l) {
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        gk[t].at(l) = gq[t].at(l);
                        DQ_KEEP(gk[t].at(l));
                    }
                });
            }
            {
                double g00, g01 = 0.0, g02 = 0.0, g11 = 0.0, g12 = 0.0, g22 = 0.0;
                if (P64) {
                    g00 = w.bcast(gk[0], np0);
                    if (new_line) {  // uniform
                        g01 = w.bcast(gk[0], np1); g02 = w.bcast(gk[0], np2);
                        g11 = w.bcast(gk[1], np1); g12 = w.bcast(gk[1], np2); g22 = w.bcast(gk[2], np2);
                    }
                } else {
                    g00 = G[np0 * ldg + np0];
                    if (new_line) {  // uniform
                        g01 = G[np0 * ldg + np1]; g02 = G[np0 * ldg + np2];
                        g11 = G[np1 * ldg + np1]; g12 = G[np1 * ldg + np2]; g22 = G[np2 * ldg + np2];
                    }
                }
                if (new_line) {  // uniform
                    const double line = ncf[0] * (ncf[0] * g00 + 2.0 * (ncf[1] * g01 + ncf[2] * g02)) +
                                        ncf[1] * (ncf[1] * g11 + 2.0 * ncf[2] * g12) + ncf[2] * ncf[2] * g22;
                    dall = w.uni(line + sig2_row);
                } else {
                    dall = w.uni((ncf[0] * ncf[0] + ncf[1] * ncf[1] + ncf[2] * ncf[2]) * g00 + sig2_row);
                }
            }
            // ---- step direction in control-point space: dx = G a - Z d1 = G (a - A' rv) ----------------------------------------
            // lane p: its control point's entry of G a, then one AXPY per active row on that row's Z column -- three reads and
            // three FMAs per row, whatever kind of row it is
            // DXREG (static shapes without slack variables, P <= 64): lane p keeps dx of control point p in registers; the
            // step is taken from there (lanes beyond P hold control point P - 1 once more)
            RegD dkr[3];
            constexpr int DD = DIMS > 0 ? DIMS : 3;
            w.each([&](int l) {
                for (int pb = 0; pb < P; pb += 64) {  // uniform trip count
                    const int p = pb + l;
                    const bool in = p < P;
                    const int pc = in ? p : P - 1;
                    double dk[3];
                    {
                        double g0, g1, g2;
                        if (P64) {
                            g0 = gk[0].at(l); g1 = gk[1].at(l); g2 = gk[2].at(l);
                        } else {
                            g0 = G[np0 * ldg + pc]; g1 = G[np1 * ldg + pc]; g2 = G[np2 * ldg + pc];
                            DQ_KEEP(g0); DQ_KEEP(g1); DQ_KEEP(g2);
                        }
                        const double sl = ncf[0] * g0 + ncf[1] * g1 + ncf[2] * g2;
                        dk[0] = new_line ? (new_axis == 0 ? sl : 0.0) : ncf[0] * g0;
                        dk[1] = new_line ? (new_axis == 1 ? sl : 0.0) : ncf[1] * g0;
                        dk[2] = new_line ? (new_axis == 2 ? sl : 0.0) : ncf[2] * g0;
                    }
                    for (int j0 = 0; j0 < q_in; j0 += JB) {  // uniform; every read of a block issued before its first use
                        double zc[JB][DD];
#pragma unroll
                        for (int u = 0; u < JB; u++) {
                            const double *zj = colw + (j0 + u < q_in ? j0 + u : q_in - 1) * cs + zo + pc;
#pragma unroll
                            for (int k = 0; k < DD; k++) zc[u][k] = zj[(k < dim ? k : 0) * P];
                        }
#pragma unroll
                        for (int u = 0; u < JB; u++) {
#pragma unroll
                            for (int k = 0; k < DD; k++) DQ_KEEP(zc[u][k]);
                        }
#pragma unroll
                        for (int u = 0; u < JB; u++) {
                            const double wj = DQ_BC(nd1, j0 + u);
#pragma unroll
                            for (int k = 0; k < DD; k++) dk[k] = fma(wj, zc[u][k], dk[k]);
                        }
                    }
                    for (int j0 = qz; j0 < q; j0 += JB) {  // records in the spill record
                        double zc[JB][DD];
#pragma unroll
                        for (int u = 0; u < JB; u++) {
                            const double *zj = colsp + (size_t)((j0 + u < q ? j0 + u : q - 1) - qz) * cs + zo + pc;
#pragma unroll
                            for (int k = 0; k < DD; k++) zc[u][k] = zj[(k < dim ? k : 0) * P];
                        }
#pragma unroll
                        for (int u = 0; u < JB; u++) {
#pragma unroll
                            for (int k = 0; k < DD; k++) DQ_KEEP(zc[u][k]);
                        }
#pragma unroll
                        for (int u = 0; u < JB; u++) {
                            const double wj = DQ_BC(nd1, j0 + u);
#pragma unroll
                            for (int k = 0; k < DD; k++) dk[k] = fma(wj, zc[u][k], dk[k]);
                        }
                    }
                    if (DXREG) {
                        dkr[0].at(l) = dk[0]; dkr[1].at(l) = dk[1]; dkr[2].at(l) = dim > 2 ? dk[2] : 0.0;
                    } else {
#pragma unroll
                        for (int k = 0; k < DD; k++)
                            if (k < dim) base[blend_idx(in, oDx + k * P + p, odump + (l & 31))] = dk[k];  // uniform condition
                    }
                }
                // slack variables: G is diag(sigma^2) there and every slack coefficient is -1
                for (int s0 = 0; s0 < nslk; s0 += 64) {  // uniform trip count
                    const int s = s0 + l;
                    const int sc = s < nslk ? s : nslk - 1;
                    const double sg = kp.sigma[sc % M];
                    double v = s == sidx_row ? -(sg * sg) : 0.0;
                    for (int j = 0; j < q_in; j++) v = fma(DQ_BC(nd1, j), colw[j * cs + zo + nxc + sc], v);
                    for (int j = qz; j < q; j++) v = fma(DQ_BC(nd1, j), colsp[(size_t)(j - qz) * cs + zo + nxc + sc], v);
                    base[blend_idx(s < nslk, oDx + nxc + s, odump + (l & 31))] = v;
                }
            });
            w.sync();
            // ---- |z|^2 = (a - A' rv)' G (a - A' rv) = a.dx - sum_i rv_i (a_i.dx): the squared length of the step in the whitened
            // space, as a quadratic form (its error is quadratic in the error of rv: a dependent row gives ~1e-30, not ~1e-15)
            auto z_squared = [&]() -> double {  // reads dx from LDS
                double adx;
                {
                    const double x0 = dxl[(new_line ? new_axis : 0) * P + np0];
                    const double x1 = dxl[(new_line ? new_axis : (dim > 1 ? 1 : 0)) * P + np1];
                    const double x2 = dxl[(new_line ? new_axis : (dim > 2 ? 2 : 0)) * P + np2];
                    const double xs = dxl[nxc + (sidx_row >= 0 ? sidx_row : 0)];
                    double s = ncf[0] * x0 + ncf[1] * x1 + (new_line || dim > 2 ? ncf[2] * x2 : 0.0);
                    s -= (SLK != 0 && sidx_row >= 0) ? xs : 0.0;
                    adx = w.uni(s);
                }
                if (q == 0) return adx;  // uniform
                RegD tr;
                w.each([&](int l) {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const uint32_t pk = a_pk[c].at(l);
                        const int ka = (pk >> 24) & 3;
                        const bool line = (pk >> 26) & 1;
                        const int sl = a_sl[c].at(l);
                        double x0 = dxl[(line ? ka : 0) * P + (int)(pk & 0xFF)];
                        double x1 = dxl[(line ? ka : (dim > 1 ? 1 : 0)) * P + (int)((pk >> 8) & 0xFF)];
                        double x2 = dxl[(line ? ka : (dim > 2 ? 2 : 0)) * P + (int)((pk >> 16) & 0xFF)];
                        double xs = SLK != 0 ? dxl[nxc + (sl >= 0 ? sl : 0)] : 0.0;
                        DQ_KEEP(x0); DQ_KEEP(x1); DQ_KEEP(x2); DQ_KEEP(xs);
                        // weights -rv x coefficient (zero beyond the active rows); POINT rows in 2-D carry a zero third coefficient
                        const double mr = l + 64 * c < q ? -rv[c].at(l) : 0.0;
                        double s = (mr * (double)a_cf[c][0].at(l)) * x0 + (mr * (double)a_cf[c][1].at(l)) * x1 + (mr * (double)a_cf[c][2].at(l)) * x2;
                        s += (SLK != 0 && sl >= 0 && l + 64 * c < q) ? rv[c].at(l) * xs : 0.0;
                        acc += s;
                    }
                    tr.at(l) = acc;
                });
                return adx + w.sum(tr);
            };
            // Whether the row DEPENDS on the active rows is decided in the whitened space itself.  dx = G (a - A' rv) of a row
            // that depends on them modulo the equality constraints is G u with u in the range of the equalities' transpose:
            // zero in exact arithmetic, eps |G| |u| in floating point -- and both forms of |z|^2 above inherit that: the
            // quotient |z|^2 / |N'a|^2 of a dependent row comes out between 1e-16 and 1e-7 (the reduced Hessian's condition
            // number is 2e5 .. 3.5e6), no threshold separates it from a row that is merely close to the active ones, and a
            // dependent row taken for an independent one gets a step of 1e13 times noise (found by tests/fuzz_gpu.py: 6 of
            // 17 622 random QPs accepted although infeasible).  So a row that looks nearly dependent is measured again the way
            // the dense method measures every row: u = a - A' rv in the variables' own space (every lane the same
            // read-modify-write, one active row after the other), z = N_T' u with the whitening map itself (read from HBM / L2,
            // here only), |z|^2 = sum z_j^2 -- the error is (eps |N| |u|)^2, a dependent row gives ~1e-28.
            double noise2 = 0.0;  // (set by whitened_length) sum_j (sum_p |N_pj| |u_p|)^2: eps^2 times it is the rounding noise of |z|^2
            auto whitened_length = [&]() -> double {
                double *u = base + lay.oU;
                const double *tN = kp.tabN + (size_t)(T - 1) * P * f;
                const int nx = nxc + nslk;
                w.each([&](int l) {
                    for (int j0 = 0; j0 < nx; j0 += 64) base[blend_idx(j0 + l < nx, lay.oU + j0 + l, odump + (l & 31))] = 0.0;  // uniform trip count
                });
                w.sync();
                for (int i = 0; i <= q; i++) {  // uniform; i == q: the new row itself
                    const bool act = i < q;
                    const int ic = act ? i : 0;
                    const uint32_t pk = act ? (uint32_t)DQ_BC(a_pk, ic) : pk_new;
                    const double mrv = -DQ_BC(rv, ic);  // active row i enters u with the weight -rv_i
                    const double c0 = act ? mrv * (double)DQ_BC2(a_cf, 0, ic) : ncf[0], c1 = act ? mrv * (double)DQ_BC2(a_cf, 1, ic) : ncf[1],
                                 c2 = act ? mrv * (double)DQ_BC2(a_cf, 2, ic) : ncf[2];
                    const int sl = act ? DQ_BCI(a_sl, ic) : sidx_row;
                    const double csl = act ? DQ_BC(rv, ic) : -1.0;  // the slack variable's coefficient is -1: -rv x -1
                    const int ka = (pk >> 24) & 3;
                    const bool line = (pk >> 26) & 1;
                    const int p0 = pk & 0xFF, p1 = (pk >> 8) & 0xFF, p2 = (pk >> 16) & 0xFF;
                    // a POINT row in 2-D has a zero third coefficient and no third axis: it is added to the first axis' entry
                    const int j0 = (line ? ka : 0) * P + p0, j1 = (line ? ka : (dim > 1 ? 1 : 0)) * P + (line ? p1 : p0);
                    const int j2 = (line ? ka : (dim > 2 ? 2 : 0)) * P + (line ? p2 : p0);
                    const double c2u = (line || dim > 2) ? c2 : 0.0;
                    w.once([&]() {
                        u[j0] += c0;
                        u[j1] += c1;
                        u[j2] += c2u;
                        if (SLK != 0 && sl >= 0) u[nxc + sl] += csl;  // uniform
                    });
                    w.sync();
                }
                RegD acc, accn;
                w.each([&](int l) {
                    double s = 0.0, sn = 0.0;
                    for (int jb = 0; jb < f; jb += 64) {  // uniform trip count
                        const int j = jb + l;
                        const int jc = j < f ? j : f - 1;
                        // (a rolled loop: this path is rare, and the kernel has no registers to lend it; one read of the map per control
                        // point serves the three axes -- the reads come from L2 one after the other and are what this path costs)
                        double z0 = 0.0, z1 = 0.0, z2 = 0.0, n0 = 0.0, n1s = 0.0, n2 = 0.0;
#pragma nounroll
                        for (int p = 0; p < P; p++) {
                            const double nv = tN[(size_t)p * f + jc], na = fabs(nv);
                            const double u0 = u[p], u1 = u[(dim > 1 ? P : 0) + p], u2 = u[(dim > 2 ? 2 * P : 0) + p];
                            z0 = fma(nv, u0, z0); n0 = fma(na, fabs(u0), n0);
                            z1 = fma(nv, u1, z1); n1s = fma(na, fabs(u1), n1s);
                            z2 = fma(nv, u2, z2); n2 = fma(na, fabs(u2), n2);
                        }
                        const double zz3 = z0 * z0 + (dim > 1 ? z1 * z1 : 0.0) + (dim > 2 ? z2 * z2 : 0.0);
                        const double nn3 = n0 * n0 + (dim > 1 ? n1s * n1s : 0.0) + (dim > 2 ? n2 * n2 : 0.0);
                        s += j < f ? zz3 : 0.0;
                        sn += j < f ? nn3 : 0.0;
                    }
                    for (int s0 = 0; s0 < nslk; s0 += 64) {  // uniform trip count
                        const int sidx = s0 + l;
                        const int sc = sidx < nslk ? sidx : 0;
                        const double v = kp.sigma[sc % M] * u[nxc + sc];
                        s += sidx < nslk ? v * v : 0.0;
                        sn += sidx < nslk ? v * v : 0.0;  // (a slack variable's own term: its absolute value is itself)
                    }
                    acc.at(l) = s;
                    accn.at(l) = sn;
                });
                double zz;
                w.sum2(acc, accn, zz, noise2);
                // nothing is left of the normal even in absolute values (DEP_XSPACE, qp_types.hpp): the answer is "zero" -- decided here, so
                // that the iteration outside is the code it was (round 4: a select on a flag of the parameter block in the iteration
                // cost the S64 launch 3 % -- register allocation --, the threshold as a number of the block 0.6 %)
                return noise2 > DEP_XSPACE * dall ? zz : 0.0;
            };
            double dd2;
            if (DXREG) {
                // a.dx from the registers: the rate at which the step moves the row it is taken for.  Equal to |z|^2 up to the
                // rounding of rv (~1e-16 x the conditioning of the active rows, relative to |N'a|^2); only a row that looks
                // nearly dependent that way is measured again as the quadratic form
                // (a POINT row: one control point, its three axes -- three broadcasts; a LINE row: three points of one axis)
                double y0, y1, y2;
                if (!new_line) {  // uniform
                    y0 = w.bcast(dkr[0], np0); y1 = w.bcast(dkr[1], np0); y2 = w.bcast(dkr[2], np0);
                } else {
                    const double e00 = w.bcast(dkr[0], np0), e01 = w.bcast(dkr[1], np0), e02 = w.bcast(dkr[2], np0);
                    const double e10 = w.bcast(dkr[0], np1), e11 = w.bcast(dkr[1], np1), e12 = w.bcast(dkr[2], np1);
                    const double e20 = w.bcast(dkr[0], np2), e21 = w.bcast(dkr[1], np2), e22 = w.bcast(dkr[2], np2);
                    y0 = new_axis == 0 ? e00 : (new_axis == 1 ? e01 : e02);
                    y1 = new_axis == 0 ? e10 : (new_axis == 1 ? e11 : e12);
                    y2 = new_axis == 0 ? e20 : (new_axis == 1 ? e21 : e22);
                }
                dd2 = ncf[0] * y0 + ncf[1] * y1 + ncf[2] * y2;
            } else {
                dd2 = z_squared();
            }
#ifdef DLSCQP_TRACE
            // trace builds (tests/test_dependency_margins.py): EVERY row is also measured in the whitened space, whether it raised
            // suspicion or not -- the truth the thresholds are judged against -- without touching what the solve goes on with
            const double dd2_first = dd2;
            double zz_truth = 0.0, noise_truth = 0.0;
            if (q > 0) {
                const double keep_noise = noise2;
                zz_truth = whitened_length();
                noise_truth = noise2;
                noise2 = keep_noise;
            }
            bool remeasured = false;
#endif
            // (suspect_now: the shape's constant, or what the worst pivot so far asks for -- DEP_PIVOT eps kappa / pivot ratio -- once that is
            // beyond the wide constant; beyond 1: every row)
            if (q > 0 && !(dd2 > suspect_now * dall)) {
                DQ_TRACE(5, dd2, dall, 0.0, 0.0);  // (the step rate that raised the suspicion)
                dd2 = whitened_length();  // uniform, a few per cent of the iterations
#ifdef DLSCQP_TRACE
                remeasured = true;
#endif
            }  // uniform, a few per cent of the iterations
            w.stamp(6);  // step direction, its whitened length
            // step lengths: t1 = min u_k / r_k over r_k > 0 (dual), t2 = viol / |z|^2 (primal)
            double t1 = 1e300;
            int ldrop = -1;
            RegI rpos;
            w.each([&](int l) {
                bool any_pos = false;
#pragma unroll
                for (int c = 0; c < RC; c++) any_pos |= (l + 64 * c < q) & (rv[c].at(l) > 0);
                rpos.at(l) = any_pos ? 1 : 0;
            });
            if (q > 0 && w.any(rpos)) {  // uniform: without a positive entry of rv no dual step is blocked
                RegD lt1;
                RegI lk;
                w.each([&](int l) {
                    double bt = 1e300;
                    int bk = -1;
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int k = l + 64 * c;
                        const double rk = rv[c].at(l);
                        const bool cand = k < q && rk > 0;
                        const double t = uu[c].at(l) / (cand ? rk : 1.0);
                        const bool better = cand && t < bt;
                        bt = better ? t : bt;
                        bk = better ? k : bk;
                    }
                    lt1.at(l) = bt;
                    lk.at(l) = bk;
                });
                t1 = w.min(lt1);
                if (t1 < 1e300) {
                    RegI is_min;
                    w.each([&](int l) { is_min.at(l) = (lt1.at(l) == t1 && lk.at(l) >= 0) ? 1 : 0; });
                    ldrop = w.bcast(lk, w.first_lane(is_min));
                }
            }
            // q == r: the active rows span the whitened space, every further row depends on them
            // (noise2 > 0: |z|^2 was measured in the whitened space: it must also stand out of the rounding noise of that measure)
            // (whitened_length answers 0 where nothing is left of the normal even in absolute values: DEP_XSPACE)
            const bool dependent = !(dd2 > DEP_TOL * dall) || !(dd2 > DEP_NOISE * noise2) || !(dd2 > 0) || q >= r;
            const double t2 = dependent ? 1e300 : viol / dd2;
            DQ_TRACE(2, dd2, dall, t1, (double)ldrop);
#ifdef DLSCQP_TRACE
            if (q > 0 && q < r) {
                // (6: a.dx / |N'a|^2 as first measured, |z|^2 / |N'a|^2 and |z|^2 / noise in the whitened space, flags: 1 = the solve
                // took the row for dependent, 2 = it was measured again, 4 = dependent by the whitened measure, 8 = the narrow suspicion
                // threshold DEP_SUSPECT applies (static shape without slack variables at the calibrated parameter values; else
                // DEP_SUSPECT_SLACK))
                const bool truth = !(zz_truth > DEP_TOL * dall) || !(zz_truth > DEP_NOISE * noise_truth) || !(zz_truth > 0) ||
                                   !(noise_truth > DEP_XSPACE * dall);
                DQ_TRACE(6, dd2_first / dall, zz_truth / dall, noise_truth > 0 ? zz_truth / noise_truth : -1.0,
                         (double)((dependent ? 1 : 0) | (remeasured ? 2 : 0) | (truth ? 4 : 0) | ((DXREG && kp.dep_calibrated) ? 8 : 0)));
                DQ_TRACE(7, noise_truth / dall, zz_truth / dall, trace_min_ratio, truth ? 1.0 : 0.0);  // what DEP_XSPACE is compared with; [3]: the smallest |z|^2 / |N'a|^2 a row was added with so far
            }
#endif
            w.stamp(7);  // ratio test
            if (t1 >= 1e300 && t2 >= 1e300) {
                // no step possible: infeasible, unless within CPLEX's feasibility tolerance
                if (viol * row_scale <= CPLEX_FEAS_TOL) {
                    set_skip(ip, true, 0.0);
                    resolved = true;
                    break;
                }
                status = DLSCQP_INFEASIBLE;
                if (out_conflict) {  // uniform
                    // The certificate (Farkas): no step is possible means  a - A' rv  lies in the range of the equalities' transpose
                    // (|z| = 0) and rv <= 0 on every active row (no dual step is blocked).  With lambda = -rv >= 0 on the active rows
                    // (which x meets with equality) and 1 on the blocked row:  sum lambda_i a_i . x'  is the same number for every x'
                    // that satisfies the equalities, and at x it is  sum lambda_i b_i - viol:  no such x' satisfies all these rows.
                    // Row ids that do not depend on the instantiation: type << 24 | index << 3 | sub -- 0: row `sub` (2 kind + side) of
                    // coordinate `index`; 1: LSC row `index` of the agent's list; 2: side `sub` of slot `index` of the context's table;
                    // 3: the bound eps <= 0 of slack variable `index`.
                    auto conflict_id = [&](int code) -> double {
                        // (sums of comparisons and flat selects: nested ?: on a lane's code came out as EXEC-masked branches)
                        const int un = code >> 3, cc = code & 7;
                        const int g1 = un >= UL0 ? 1 : 0, g2 = un >= UX0 ? 1 : 0, g3 = un >= US0 ? 1 : 0;
                        const int ty = g1 + g2 + g3;
                        int ix = un;
                        ix = g1 ? un - UL0 : ix;
                        ix = g2 ? un - UX0 + slot0 : ix;
                        ix = g3 ? un - US0 : ix;
                        return (double)((ty << 24) | (ix << 3) | (cc & ((ty & 1) - 1)));  // (types 1 and 3 have no sub-row)
                    };
                    // (every lane stores: lanes beyond the active rows to the blocked row's pair, which is written after them)
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int i = l + 64 * c;
                            const int t_ = i < q ? i : q;
                            out_conflict[4 + 2 * t_] = conflict_id(act[c].at(l) < 0 ? 0 : act[c].at(l));
                            out_conflict[5 + 2 * t_] = -rv[c].at(l);
                        }
                    });
                    w.sync();
                    w.each([&](int) {  // every lane, the same words, the same values
                        out_conflict[4 + 2 * q] = conflict_id(ip);
                        out_conflict[5 + 2 * q] = 1.0;
                        out_conflict[0] = (double)(q + 1);
                        out_conflict[1] = viol;
                        out_conflict[2] = row_scale;
                        out_conflict[3] = 0.0;
                    });
                }
                break;
            }
            const double t = t1 < t2 ? t1 : t2;
            if (!dependent) {
                // primal step  x += t dx
                w.each([&](int l) {
                    constexpr int NXC = DIMS > 0 && SLK == 0 ? (DIMS * MS * 6 + 63) / 64 : 1;
                    const int nx = nxc + nslk;
                    if (DXREG) {
                        const int pc = l < P ? l : P - 1;
                        double xo[3];
#pragma unroll
                        for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++) xo[k] = x[k * P + pc];
#pragma unroll
                        for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++) DQ_KEEP(xo[k]);
#pragma unroll
                        for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++)
                            base[blend_idx(l < P, oX + k * P + l, odump + (l & 31))] = fma(t, dkr[k].at(l), xo[k]);
                    } else if (DIMS > 0 && SLK == 0) {
                        double xo[NXC], dv[NXC];
#pragma unroll
                        for (int u = 0; u < NXC; u++) {
                            const int j = 64 * u + l;
                            const int jc = j < nx ? j : nx - 1;
                            xo[u] = x[jc]; dv[u] = dxl[jc];
                        }
#pragma unroll
                        for (int u = 0; u < NXC; u++) { DQ_KEEP(xo[u]); DQ_KEEP(dv[u]); }
#pragma unroll
                        for (int u = 0; u < NXC; u++) {
                            const int j = 64 * u + l;
                            base[blend_idx(j < nx, oX + j, odump + (l & 31))] = fma(t, dv[u], xo[u]);
                        }
                    } else {
                        for (int j0 = 0; j0 < nx; j0 += 64) {  // uniform trip count
                            const int j = j0 + l;
                            const int jc = j < nx ? j : nx - 1;
                            const double xn = fma(t, dxl[jc], x[jc]);
                            base[blend_idx(j < nx, oX + j, odump + (l & 31))] = xn;
                        }
                    }
                });
                viol -= t * dd2;
                // |t dx|_H = t sqrt(a.dx); single precision and 2 % on top (a.dx of a nearly dependent row carries its rounding)
                if (!ALL_CACHED) path += 1.02 * t * (double)sqrtf((float)dd2);
            }
            w.each([&](int l) {
#pragma unroll
                for (int c = 0; c < RC; c++) {
                    const double un = fma(-t, rv[c].at(l), uu[c].at(l));
                    uu[c].at(l) = l + 64 * c < q ? un : uu[c].at(l);
                }
            });
            uq += t;
            w.sync();
            w.stamp(8);  // primal / dual step
            if (!dependent && t2 <= t1) {
                // ---- add row ip: R gains the column (d1; alpha), alpha = |z|  <=>  S = R^-1 gains (-S d1 / alpha; 1 / alpha; zeros),
                // and Z = G A' S the column G (a - A' rv) / alpha = dx / alpha: record q, every entry of it written ----
                const double inv_alpha = 1.0 / sqrt(dd2);
                {
                    // the threshold this pivot asks for (1e300: every row from here on -- dd2 > 1e300 dall never holds)
                    double by_pivot = kp.dep_pivot * (dall * inv_alpha * inv_alpha);
                    by_pivot = by_pivot >= 1.0 ? 1e300 : by_pivot;
                    suspect_now = (by_pivot > DEP_SUSPECT_SLACK && by_pivot > suspect_now) ? by_pivot : suspect_now;
                }
#ifdef DLSCQP_TRACE
                trace_min_ratio = dd2 / dall < trace_min_ratio ? dd2 / dall : trace_min_ratio;
#endif
                // (called with a pointer into the working set or into the spill record: inlined at both places, so that each copy knows
                // its address space -- as a function of its own it would take a generic pointer, and the flat-address code that goes
                // with one hit "Illegal instruction detected: Operand has incorrect register class" in this compiler)
                auto write_record = [&](double *rec) __attribute__((always_inline)) {
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int i = l + 64 * c;
                            const int ic = i < r_max ? i : r_max - 1;  // (lanes beyond r_max: entry r_max - 1 once more, the same value)
                            const double sv = ic < q ? -rv[c].at(l) * inv_alpha : (ic == q ? inv_alpha : 0.0);
                            rec[ic] = sv;
                        }
                        if (DXREG) {
                            const int pc = l < P ? l : P - 1;
#pragma unroll
                            for (int k = 0; k < DD; k++) rec[zo + k * P + pc] = dkr[k].at(l) * inv_alpha;
                        } else {
                            const int nx = nxc + nslk;
                            for (int j0 = 0; j0 < nx; j0 += 64) {  // uniform trip count
                                const int jc = j0 + l < nx ? j0 + l : nx - 1;
                                rec[zo + jc] = dxl[jc] * inv_alpha;
                            }
                        }
                    });
                };
                if (q < qz) write_record(colw + q * cs);  // uniform
                else write_record(colsp + (size_t)(q - qz) * cs);
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        const bool me = i == q;
                        uu[c].at(l) = me ? uq : uu[c].at(l);
                        act[c].at(l) = me ? ip : act[c].at(l);
                        a_pk[c].at(l) = me ? pk_new : a_pk[c].at(l);
                        a_sl[c].at(l) = me ? sidx_row : a_sl[c].at(l);
                        a_rhs[c].at(l) = me ? row_rhs : a_rhs[c].at(l);
#pragma unroll
                        for (int t = 0; t < 3; t++) a_cf[c][t].at(l) = me ? (float)ncf[t] : a_cf[c][t].at(l);
                    }
                });
                set_skip(ip, true, 0.0);
                q++;
                n_add++;
                resolved = true;
                w.stamp(9);  // add
            } else {
                // ---- drop active row ldrop: delete column l of R, restore the triangle with Givens ----
                const int lcol = ldrop;
                set_skip(DQ_BCI(act, lcol), false, DQ_BC(a_rhs, lcol));
                if (!ALL_CACHED) w.each([&](int l) { expiry.at(l) = -1e300; });  // the row is priced again: its group, whichever it is
                // entry `idx` of record k (k uniform): from the working set or from the spill record
                auto rec_ld = [&](int k, int idx) -> double {
                    if (k < qz) return colw[k * cs + idx];
                    return colsp[(size_t)(k - qz) * cs + idx];
                };
                auto rec_st = [&](int k, bool cond, int idx, double v, int l) {  // lanes with nothing to store: to a dump slot
                    if (k < qz) base[blend_idx(cond, oC + k * cs + idx, odump + (l & 31))] = v;
                    else colsp[blend_idx(cond, (k - qz) * cs + idx, -32 + (l & 31))] = v;
                };
                // Deleting column lcol of R: the rotations of columns (t, t+1), t = lcol..q-2, that restore the
                // triangle are those that sweep row lcol of S into its last entry; S' = (S without row lcol) G'
                // without its last column, and Z = G A' S takes the same rotations of its columns (what row lcol of S
                // multiplied -- G a_lcol -- is gone from all but the last).  All angles come from one prefix sum of squares.
                RegD srow[RC], pre[RC], gc[RC], gs[RC];
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int k = l + 64 * c;
                        const bool in = k >= lcol && k < q;
                        double sv = colw[((in & (k < qz)) ? k : 0) * cs + lcol];  // S(lcol, k): lane k reads ITS record
                        DQ_KEEP(sv);
                        if (q > qz) {  // uniform
                            const bool sp = in & (k >= qz);
                            double sg = colsp[(size_t)blend_idx(sp, k - qz, 0) * cs + lcol];
                            DQ_KEEP(sg);
                            sv = sp ? sg : sv;
                        }
                        const double v = in ? sv : 0.0;
                        srow[c].at(l) = v;
                        pre[c].at(l) = v * v;
                    }
                });
                w.prefix_sum(pre, RC);
                {
                    RegD pprev[RC];
#pragma unroll
                    for (int c = 0; c < RC; c++) w.shift_up(pre, c, RC, pprev[c]);  // element k-1
                    const double s_first = DQ_BC(srow, lcol);  // the first pivot keeps its sign, later ones are norms
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            // lane k = t+1 holds the rotation of step t: (a, b) = (|s_lcol..s_t|, s_{t+1}) -> (0, |..|)
                            const int k = l + 64 * c;
                            const double sig = sqrt(pre[c].at(l));
                            double sp = sqrt(pprev[c].at(l));  // prefix sums of squares: never negative
                            DQ_KEEP(sp);
                            double a = k > lcol ? sp : 0.0;
                            a = k == lcol + 1 ? s_first : a;
                            const bool ok = k > lcol && k < q && sig > 0;
                            gc[c].at(l) = ok ? srow[c].at(l) / sig : 1.0;
                            gs[c].at(l) = ok ? a / sig : 0.0;
                        }
                    });
                }
                RegD yS[RC];
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        const double s0 = rec_ld(lcol, i <= lcol ? i : 0);
                        yS[c].at(l) = i <= lcol ? s0 : 0.0;
                    }
                });
                for (int t = lcol; t < q - 1; t++) {
                    const double rc_ = DQ_BC(gc, t + 1), rs_ = DQ_BC(gs, t + 1);
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int i = l + 64 * c;
                            // S: old column t+1 has rows 0..t+1; row lcol is deleted, rows below it move up
                            const double cS0 = rec_ld(t + 1, i <= t + 1 ? i : 0);
                            const double cS = i <= t + 1 ? cS0 : 0.0;
                            const double oS = rc_ * yS[c].at(l) - rs_ * cS;
                            yS[c].at(l) = rs_ * yS[c].at(l) + rc_ * cS;
                            const int inew = i < lcol ? i : i - 1;
                            rec_st(t, (i != lcol) & (inew <= t), inew, oS, l);
                        }
                    });
                    w.sync();
                }
                // the same rotations on the columns of Z, 64 entries of a column per trip (a lane keeps its entry of the column
                // that is carried along in a register; no lane touches another lane's entries: no sync between the steps)
                w.each([&](int l) {
                    const int nx = nxc + nslk;
                    for (int j0 = 0; j0 < nx; j0 += 64) {  // uniform trip count
                        const bool in = j0 + l < nx;
                        const int e = zo + (in ? j0 + l : nx - 1);
                        double y = rec_ld(lcol, e);
                        for (int t = lcol; t < q - 1; t++) {
                            const double rc_ = DQ_BC(gc, t + 1), rs_ = DQ_BC(gs, t + 1);
                            const double cz = rec_ld(t + 1, e);
                            const double oz = rc_ * y - rs_ * cz;
                            y = rs_ * y + rc_ * cz;
                            rec_st(t, in, e, oz, l);
                        }
                    }
                });
                w.sync();
                // compact the per-lane state of the active rows: element k takes element k+1, k >= lcol
                {
                    RegD nu[RC], nrh[RC];
                    RegI nact[RC], nsl[RC];
                    RegU npk[RC];
                    RegF ncfr[RC][3];
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        w.shift_down(uu, c, RC, nu[c]);
                        w.shift_down(a_rhs, c, RC, nrh[c]);
                        w.shift_down(act, c, RC, nact[c]);
                        w.shift_down(a_sl, c, RC, nsl[c]);
                        w.shift_down(a_pk, c, RC, npk[c]);
                    }
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        RegF col[RC];
                        w.each([&](int l) {
#pragma unroll
                            for (int c = 0; c < RC; c++) col[c].at(l) = a_cf[c][t].at(l);
                        });
#pragma unroll
                        for (int c = 0; c < RC; c++) w.shift_down(col, c, RC, ncfr[c][t]);
                    }
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int k = l + 64 * c;
                            const bool mv = k >= lcol && k < q - 1;
                            uu[c].at(l) = mv ? nu[c].at(l) : uu[c].at(l);
                            a_rhs[c].at(l) = mv ? nrh[c].at(l) : a_rhs[c].at(l);
                            act[c].at(l) = mv ? nact[c].at(l) : act[c].at(l);
                            a_sl[c].at(l) = mv ? nsl[c].at(l) : a_sl[c].at(l);
                            a_pk[c].at(l) = mv ? npk[c].at(l) : a_pk[c].at(l);
#pragma unroll
                            for (int t = 0; t < 3; t++) a_cf[c][t].at(l) = mv ? ncfr[c][t].at(l) : a_cf[c][t].at(l);
                        }
                    });
                }
                q--;
                n_drop++;
                w.stamp(10);  // Givens drop
                // loop again with the same row ip (step 2(a) of Goldfarb-Idnani)
            }
        }
        if (status != DLSCQP_OK) break;
    }

    // ================================ read-out ==================================================
    if (status != DLSCQP_OK) {
        exit_stores(status, true, iters, n_add, n_drop, q, false, 0.0);
        return;
    }
    // objective before the guarded stores below, so that nothing computed later is live across them.  An agent that never
    // iterated sits at its unconstrained minimiser: its objective is a 4 x 4 quadratic form per axis (qp_tables.cpp: tabPhi)
    double fval = 0.0;
    if (out_cost) {
        if (iters == 0 && kp.phi_ok && DIMS > 0) {  // uniform
            fval = fval0;
        } else if (iters == 0 && kp.phi_ok) {
            const double *ph = kp.tabPhi + (size_t)(T - 1) * 10;
            double pq[10];
#pragma unroll
            for (int j = 0; j < 10; j++) pq[j] = ph[j];
            for (int k = 0; k < dim; k++) {
                const double s0 = (double)pos[k], s1 = (double)vel[k], s2 = (double)acc[k], s3 = (double)goal[k];
                fval += s0 * (pq[0] * s0 + 2.0 * (pq[1] * s1 + pq[2] * s2 + pq[3] * s3)) + s1 * (pq[4] * s1 + 2.0 * (pq[5] * s2 + pq[6] * s3)) +
                        s2 * (pq[7] * s2 + 2.0 * pq[8] * s3) + pq[9] * s3 * s3;
            }
        } else {
            fval = objective();
        }
    }
    // Per-lane results: the only lane-guarded instructions of the kernel are these final global stores.
    w.each([&](int l) {
        // traj_optimizer.cpp:69-83: point3d(float) control points, z = world_z_2d in 2-D
        if (out_ctrl)
            for (int j0 = 0; j0 < M * n1 * 3; j0 += 64) {
                const int j = j0 + l;
                const bool in = j < M * n1 * 3;
                const int jc = in ? j : 0;
                const int p = jc / 3, k = jc - p * 3;
                const double xv = x[(k < dim ? k : 0) * P + p];
                const float v = k < dim ? (float)xv : (float)kp.z_2d;
                if (in) out_ctrl[j] = v;
            }
        if (out_x)
            for (int j0 = 0; j0 < nxc + nslk; j0 += 64) {
                const int j = j0 + l;
                const bool in = j < nxc + nslk;
                const double v = x[in ? j : 0];
                if (in) out_x[j] = v;
            }
        // traj_optimizer.cpp:85-106
        if (out_slack_cost)
            for (int s0 = 0; s0 < max_slack_obs; s0 += 64) {
                const int s = s0 + l;
                const bool in = s < max_slack_obs;
                const bool live = in & (s < nso);
                double sc = 0;
                for (int m = 0; m < M; m++) sc += fabs(x[nxc + (live ? s * M + m : 0)]);
                if (in) out_slack_cost[s] = live ? sc : 0.0;
            }
    });
    exit_stores(status, true, iters, n_add, n_drop, q, true, fval);
    w.stamp(11);  // read-out
}

#undef DQ_BC
#undef DQ_BCI
#undef DQ_BC2

}  // namespace dlscqp
