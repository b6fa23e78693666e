// gi_core.hpp -- per-agent goal LP + trajectory QP, one agent per wavefront.
//
// Replaces GoalOptimizer::solve (goal_optimizer.cpp:7-198) and TrajOptimizer::solve
// (traj_optimizer.cpp:18-165, model at :225-527) of the reference.  The equalities are eliminated
// and the Hessian whitened on the host (qp_tables.cpp), which leaves, per agent, a projection
// problem  min 1/2 |w|^2  s.t. sparse half-spaces in control-point space, solved here with a
// Goldfarb-Idnani dual active-set method:
//   * every constraint row the wave will ever look at is decoded ONCE into registers: lane l owns
//     LSC rows l, l+64, ... (coalesced SoA loads) and structured slots l, l+64, ...; a scan is then
//     pure LDS reads of control points + FMAs, followed by one arg-max butterfly;
//   * the small vectors of the method (whitened normal, d = Q'a, step direction z, R^-1 d, duals,
//     active list) are one register per lane, broadcast with readlane -- no LDS round trips;
//   * Q (r x r orthogonal) and S = R^-1 (packed upper triangle; R = triangular factor of the active
//     normals) live in LDS and are set up lazily: the ~60 % of agents whose unconstrained optimum is
//     feasible never touch them.  Keeping the INVERSE turns the O(q)-deep back substitution R^-1 d into
//     one lane-parallel product, appending a row into writing one column (-S d1 / alpha, 1 / alpha), and
//     deleting a row into a column-rotation sweep whose angles come from one prefix sum of row l of S;
//   * a row is added with ONE Householder reflection (rank-1 update of the trailing block of Q).
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
// DQ_MARK(text): a comment in the generated assembly (tests/divergence_check.py recognises the exit stores by it)
#ifndef DQ_MARK
#define DQ_MARK(text) ((void)0)
#endif

namespace dlscqp {

// KU: LSC rows cached in registers per lane (64 KU rows; the rest is read from HBM on every scan)
// SU: structured slots cached per lane (64 SU slots; the rest from HBM).  Template parameters of
// plan_agent: the register budget of the wave (256 VGPRs at 2 waves/SIMD) is shared with the unrolled loops.

struct MemLayout {
    int r_max, nx_max, ldq, nflag;
    int oQ, oRp, oN, oX, oBlo, oBhi;  // in doubles
    int oDump;                        // 32 dump slots (lane & 31) for the lanes that have nothing to store: the spare row of Q
    int oFlag;                        // byte offset of the flags of rows beyond the register cache
    int kmax;                         // most LSC rows of one agent the layout was sized for (dlscqp_device_batch.max_rows_per_agent)
    int bytes;
    int s_ext_doubles;                // > 0: S = R^-1 is not part of this block but lives in a buffer of that many doubles (dump slots included)
};

DQ_FN MemLayout make_layout(const KParams &kp, int max_slack_obs, int kmax, int KU, int SU, bool n_in_lds = true, bool s_in_block = true) {
    MemLayout L;
    const int nslk = kp.M * max_slack_obs;
    L.r_max = kp.dim * kp.f + nslk;
    L.nx_max = kp.dim * kp.P + nslk;
    L.ldq = L.r_max | 1;  // odd leading dimension: conflict-free column and row sweeps
    const int cold_rows = kmax > 64 * KU ? kmax - 64 * KU : 0;
    const int cold_slots = kp.n_slots > 64 * SU ? kp.n_slots - 64 * SU : 0;
    L.nflag = cold_rows + 2 * cold_slots;
    L.kmax = kmax;
    int o = 0;
    L.oQ = o; o += (L.r_max + 1) * L.ldq;  // + one row for the lanes beyond r (see "no divergence" below)
    L.oDump = L.oQ + L.r_max * L.ldq;      // that row doubles as the dump slots: its content is never used
    if (L.ldq < 32) o += 32 - L.ldq;
    L.oRp = o; o += s_in_block ? L.r_max * (L.r_max + 1) / 2 : 0;
    L.s_ext_doubles = s_in_block ? 0 : L.r_max * (L.r_max + 1) / 2 + 32;
    L.oN = o; o += n_in_lds ? kp.P * kp.f : 0;  // instantiations that keep N_T's rows in registers need no copy
    L.oX = o; o += L.nx_max;
    L.oBlo = o; o += kp.nb;
    L.oBhi = o; o += kp.nb;
    L.oFlag = o * 8;
    L.bytes = (L.oFlag + L.nflag + 1 + 15) & ~15;  // + one flag byte as the dump of the flag array
    return L;
}

// Instantiations of plan_agent: id, RC, DIMS, FS, MS, KU, SU, SLK.  Static shapes for the reference's launch
// files (3-D M=5: f=13; 2-D / 3-D M=10: f=28) with register caches sized to the row counts, with and without
// slack (dynamic) obstacles; runtime-shape instantiations for everything else.  One list for the HIP launcher
// and for the lane-serial test build.
#ifndef DQ_G2_KU  // register caches of the two-register runtime-shape instantiation (also the one with HBM working-set variants)
#define DQ_G2_KU 2
#define DQ_G2_SU 6
#endif
#define DQ_INSTANCES(X)          \
    X(0, 1, 3, 13, 5, 2, 4, 0)   \
    X(1, 1, 3, 13, 5, 4, 5, 0)   \
    X(2, 1, 3, 13, 5, 4, 5, -1)  \
    X(3, 1, 2, 28, 10, 3, 5, 0)  \
    X(4, 2, 3, 28, 10, 2, 6, 0)  \
    X(5, 2, 3, 28, 10, 2, 6, -1) \
    X(6, 1, 0, 0, 0, 2, 6, -1)   \
    X(7, 2, 0, 0, 0, DQ_G2_KU, DQ_G2_SU, -1)
constexpr int DQ_N_INSTANCES = 8;

struct InstanceChoice {
    int id, KU, SU;  // id < 0: unsupported (reduced dimension above 128)
};

// Instantiations with f <= 16 and at most 64 control points per axis (the 3-D M = 5 ones) keep lane p's row of N_T
// in registers and read the three rows of a constraint normal straight from the (L2-resident) table: no copy of N_T
// in LDS, which is what fits an eighth agent on a CU.
constexpr bool n_in_registers(int FS, int MS) { return FS > 0 && FS <= 16 && MS * 6 <= 64; }
inline bool instance_n_in_lds(int id) {
    switch (id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID: return !n_in_registers(FS, MS);
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    default: return true;
    }
}

// Which instantiation plans a batch (r_max = dim f + M max_slack_obs, kmax = most LSC rows of one agent).
// register-cache sizes of an instantiation, from the table above (the one place they are written down)
inline InstanceChoice instance_choice(int id) {
    switch (id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID: return {ID, KU, SU};
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    default: return {-1, 0, 0};
    }
}

inline InstanceChoice select_instance(const KParams &kp, int max_slack_obs, int kmax, bool force_generic) {
    const int r_max = kp.dim * kp.f + kp.M * max_slack_obs;
    const bool ns = max_slack_obs == 0;
    if (!force_generic) {
        if (kp.dim == 3 && kp.f == 13 && kp.M == 5 && r_max <= 64) {
            const InstanceChoice all_cached = instance_choice(0);  // compiled without the loops over rows / slots beyond its caches
            if (ns && kmax <= 64 * all_cached.KU && kp.n_slots <= 64 * all_cached.SU) return all_cached;
            return instance_choice(ns ? 1 : 2);
        }
        if (kp.dim == 2 && kp.f == 28 && kp.M == 10 && ns) return instance_choice(3);
        if (kp.dim == 3 && kp.f == 28 && kp.M == 10 && r_max <= 128) return instance_choice(ns ? 4 : 5);
    }
    if (r_max <= 64) return instance_choice(6);
    if (r_max <= 128) return instance_choice(7);
    return {-1, 0, 0};
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

// v[k] for a per-lane k without dynamic indexing of a register array (which would go through scratch memory)
template <class T>
DQ_FN T pick3(const T *v, int k) {
    const T v0 = v[0], v1 = v[1], v2 = v[2];  // flat selects: nested ?: came out as EXEC-masked branches
    T r = v2;
    r = k == 1 ? v1 : r;
    r = k == 0 ? v0 : r;
    return r;
}

#define DQ_RP(i, j) Rp[(j) * ((j) + 1) / 2 + (i)]
// element e of a per-lane vector spread over RC registers
#define DQ_BC(arr, e) (RC == 1 ? w.bcast((arr)[0], (e)) : ((e) < 64 ? w.bcast((arr)[0], (e)) : w.bcast((arr)[RC - 1], (e) - 64)))

// RC: registers per per-lane vector (r <= 64 RC).  DIMS/FS/MS: compile-time world dimension, free
// degrees of freedom per axis and segment count (0 = read them from KParams).  With static shapes every O(r) / O(f) loop
// below has a constant trip count, is fully unrolled, and its LDS reads are issued back to back instead
// of one dependent round trip per iteration -- the difference between ~35k and a few thousand cycles per
// active-set iteration for a single wavefront.
//
// NO DIVERGENCE.  Every lane executes every instruction of plan_agent: there is no branch and no loop bound that
// depends on the lane.  Lanes that have no element to work on use a clamped (valid) index for their loads,
// mask the result with a select, and send their stores to a dump slot (MemLayout::oDump, the extra row of Q).
// This is a correctness rule, not a style: the solver keeps ~100 VGPRs of per-lane state and wave-uniform
// values (LDS addresses, cached rows) alive across the whole iteration, and when the register allocator
// splits or spills such a live range INSIDE a lane-divergent region, the copy runs under the region's EXEC
// mask and the inactive lanes come back as garbage after the region (seen on gfx950 / ROCm 7.2: lanes
// >= r - 64 of the two-register generic instantiation, depending on unrelated source changes).  Branches on
// wave-uniform conditions do not touch EXEC and are fine.
#ifdef DLSCQP_TRACE
#define DQ_TRACE(tag, a, b, c, d) w.trace(tag, a, b, c, d)  // debug builds only (tests/trace_gpu.py)
#else
#define DQ_TRACE(tag, a, b, c, d)
#endif

// SLK: compile-time number of slack obstacles of the batch (0), or -1 = read it from the arguments.  With
// DIMS > 0 and SLK = 0 the whole LDS layout up to the bound table (ldq, offsets of S, N, x, dump) and the whitened
// dimension r are constants: LDS addresses become "lane part + immediate", no address arithmetic in the loops.
// S_EXT: the packed triangle S = R^-1 lives in `s_ext` (HBM, MemLayout::s_ext_doubles doubles of this agent's own) instead
// of the block at `mem`: for reduced dimensions whose Q AND S do not fit the LDS of a CU together (r > 110).  The sweeps
// over Q -- three of the four O(r^2) passes of an iteration -- stay in LDS.
template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK, class W, bool S_EXT = false>
DQ_FN void plan_agent(const W &w, const KParams &kp, const MemLayout &lay, unsigned char *mem,
                      const dlscqp_agent *agp, const RowsView &rows, int row_begin, int K_rows,
                      const float *sfc, int max_slack_obs, int do_goal, int do_traj,
                      float *out_goal, int *out_goal_status, float *out_ctrl, double *out_x,
                      double *out_cost, double *out_slack_cost, int *out_status, int *out_iters, double *s_ext = nullptr) {
    static_assert(RC == 1 || RC == 2, "r <= 128");
    typedef typename W::template Reg<double> RegD;
    typedef typename W::template Reg<float> RegF;
    typedef typename W::template Reg<uint32_t> RegU;
    typedef typename W::template Reg<int> RegI;

    constexpr int CB = 4;  // rows beyond the register cache are read in trips of CB blocks of 64
    // select_instance() hands the KU = 2 instantiation only batches whose rows and slots all fit the register caches
    // (kmax <= 128, n_slots <= 256): it is compiled without the loops over rows and slots beyond them
    constexpr bool ALL_CACHED = KU == 2 && SU == 4;
    const int K = ALL_CACHED && K_rows > 64 * KU ? 64 * KU : K_rows;
    // The caller's max_rows_per_agent picked this instantiation and sized the flag array: an agent with more rows than
    // declared (or, in the all-cached instantiation, more rows or slots than the register caches hold) is not planned
    // with some of its rows dropped -- it fails with DLSCQP_NUMERICAL.
    const bool bad_shape = K_rows < 0 || K_rows > lay.kmax || (ALL_CACHED && (K_rows > 64 * KU || kp.n_slots > 64 * SU));
    const int dim = DIMS > 0 ? DIMS : kp.dim, f = FS > 0 ? FS : kp.f;
    const int M = MS > 0 ? MS : kp.M;
    constexpr int n1 = 6;  // n = 5 is the only degree the reference supports (traj_optimizer.cpp:209)
    const int P = M * n1;
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
    int nso_rt = agp->n_slack_obs;
    if (nso_rt > max_slack_obs) nso_rt = max_slack_obs;
    if (nso_rt < 0) nso_rt = 0;
    const int nso = SLK == 0 ? 0 : nso_rt;
    const int nslk = M * nso;

    const float *Rnx = rows.nx + row_begin, *Rny = rows.ny + row_begin, *Rnz = rows.nz + row_begin;
    const float *Rpx = rows.px + row_begin, *Rpy = rows.py + row_begin, *Rpz = rows.pz + row_begin;
    const double *Rd = rows.d + row_begin;
    const uint32_t *Rkey = rows.key + row_begin;

    // ---- LSC rows into registers: lane l owns rows l + 64u (coalesced loads, once) --------------
    RegF cnx[KU], cny[KU], cnz[KU], cpx[KU], cpy[KU], cpz[KU];
    RegD cd[KU];
    RegU ckey[KU];
    const int Kc = ALL_CACHED ? K : (K < 64 * KU ? K : 64 * KU);  // cached rows
#pragma unroll
    for (int u = 0; u < KU; u++) {
        w.each([&](int l) {
            cnx[u].at(l) = 0; cny[u].at(l) = 0; cnz[u].at(l) = 0;
            cpx[u].at(l) = 0; cpy[u].at(l) = 0; cpz[u].at(l) = 0;
            cd[u].at(l) = 0;
            ckey[u].at(l) = KEY_SKIP;
        });
        if (64 * u < Kc) {  // uniform
            w.each([&](int l) {
                const int j = l + 64 * u;
                const bool in = j < Kc;
                const int jc = in ? j : Kc - 1;  // clamped: every lane loads, lanes beyond the rows drop the value
                const float a0 = Rnx[jc], a1 = Rny[jc], a2 = Rnz[jc], b0 = Rpx[jc], b1 = Rpy[jc], b2 = Rpz[jc];
                const double dd = Rd[jc];
                const uint32_t kk = Rkey[jc];
                cnx[u].at(l) = in ? a0 : 0.0f; cny[u].at(l) = in ? a1 : 0.0f; cnz[u].at(l) = in ? a2 : 0.0f;
                cpx[u].at(l) = in ? b0 : 0.0f; cpy[u].at(l) = in ? b1 : 0.0f; cpz[u].at(l) = in ? b2 : 0.0f;
                cd[u].at(l) = in ? dd : 0.0;
                ckey[u].at(l) = in ? kk : KEY_SKIP;
            });
        }
    }

    // ---- structured slots into registers (issued here so that all prologue loads are in flight together)
    const Slot *slots = kp.slots;
    const int n_slots = ALL_CACHED && kp.n_slots > 64 * SU ? 64 * SU : kp.n_slots;
    const int Sc = ALL_CACHED ? n_slots : (n_slots < 64 * SU ? n_slots : 64 * SU);  // cached slots
    RegU sidx[SU], scb[SU];  // 3 x 10-bit x-space indices (0x3FF = unused) | 3 x int8 coefficients + bsel
#pragma unroll
    for (int u = 0; u < SU; u++) {
        {
            w.each([&](int l) {
                sidx[u].at(l) = 0xBFFFFFFFu;  // bit 31: no slot in this register
                scb[u].at(l) = 0;
            });
            if (64 * u < Sc) {  // uniform
                w.each([&](int l) {
                    const int s = l + 64 * u;
                    const bool in = s < Sc;
                    const Slot sl = slots[in ? s : Sc - 1];
                    const uint32_t pi = (uint32_t)(sl.idx[0] & 0x3FF) | ((uint32_t)(sl.idx[1] & 0x3FF) << 10) |
                                        ((uint32_t)(sl.idx[2] & 0x3FF) << 20);
                    const uint32_t pc = (uint32_t)(uint8_t)sl.coef[0] | ((uint32_t)(uint8_t)sl.coef[1] << 8) |
                                        ((uint32_t)(uint8_t)sl.coef[2] << 16) | ((uint32_t)sl.bsel << 24);
                    sidx[u].at(l) = in ? pi : 0xBFFFFFFFu;
                    scb[u].at(l) = in ? pc : 0u;
                });
            }
        }
    }

    // ================================ goal LP ===================================================
    // min t in [0, 1+1e-5] s.t. a_r t + b_r >= 0 (goal_optimizer.cpp:138-198), closed form.
    float goal[3] = {cg[0], cg[1], cg[2]};
    int goal_status = DLSCQP_OK;
    if (do_goal && !bad_shape) {
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
            // goal_optimizer.cpp:176-195: last control point of the last segment, static obstacles.
            // One pass over the rows: per-lane bounds, then max / min / any across the wave.
            RegD llo, lhi;
            RegI lbad, lviol;
            auto lp_row = [&](uint32_t key, float nx, float ny, float nz, float qx, float qy, float qz, double dd,
                              double &lo, double &hi, int &bad, int &viol) {
                // select form (no lane-dependent branch): rows that are not part of the LP leave lo/hi/bad/viol as they are
                const int m = key & 0xFF, i = (key >> 8) & 0xF, dyn = (key >> 12) & 1;
                const bool in_lp = (key != KEY_SKIP) & (dyn == 0) & (m == M - 1) & (i == n) & !f32_norm3_below_eps(nx, ny, nz);
                const float ox = wp[0] - qx, oy = wp[1] - qy, oz = wp[2] - qz;
                double a = (double)nx * (double)diff[0] + (double)ny * (double)diff[1];
                double b = -dd + (double)nx * (double)ox + (double)ny * (double)oy;
                const double a3 = a + (double)nz * (double)diff[2], b3 = b + (double)nz * (double)oz;
                a = dim == 3 ? a3 : a;
                b = dim == 3 ? b3 : b;
                double tq = -b / a;  // not finite when a == 0: never selected then
                DQ_KEEP(tq);
                lo = (in_lp & (a > 0) & (tq > lo)) ? tq : lo;
                hi = (in_lp & (a < 0) & (tq < hi)) ? tq : hi;
                bad = (in_lp & !(a > 0) & !(a < 0) & (b < -CPLEX_FEAS_TOL)) ? 1 : bad;
                // exception branch of goal_optimizer.cpp:63-77: does current_goal_point violate the row?
                const float e0 = cg[0] - qx, e1 = cg[1] - qy, e2 = cg[2] - qz;
                const float dot = nx * e0 + ny * e1 + nz * e2;  // float32 dot of float32 differences
                viol = (in_lp & ((double)dot - dd < -SP_EPSILON_FLOAT)) ? 1 : viol;
            };
            w.each([&](int l) {
                double lo = -1e300, hi = 1e300;
                int bad = 0, viol = 0;
#pragma unroll
                for (int u = 0; u < KU; u++)
                        lp_row(ckey[u].at(l), cnx[u].at(l), cny[u].at(l), cnz[u].at(l), cpx[u].at(l), cpy[u].at(l),
                               cpz[u].at(l), cd[u].at(l), lo, hi, bad, viol);
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
                        kk[u] = Rkey[jc];
                        g[u][0] = Rnx[jc]; g[u][1] = Rny[jc]; g[u][2] = Rnz[jc];
                        g[u][3] = Rpx[jc]; g[u][4] = Rpy[jc]; g[u][5] = Rpz[jc];
                        gd[u] = Rd[jc];
                    }
#pragma unroll
                    for (int u = 0; u < CB; u++) {  // loaded by every lane: no branch around the loads of rows past the end
                        DQ_KEEP(kk[u]); DQ_KEEP(gd[u]);
#pragma unroll
                        for (int t = 0; t < 6; t++) DQ_KEEP(g[u][t]);
                    }
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        const uint32_t kq = blend_idx(j0 + 64 * u + l < K, (int)kk[u], (int)KEY_SKIP);
                        lp_row(kq, g[u][0], g[u][1], g[u][2], g[u][3], g[u][4], g[u][5], gd[u], lo, hi, bad, viol);
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
                    const float *bmin = sfc + (M - 1) * 6, *bmax = bmin + 3;
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
    w.stamp(0);  // agent record, row cache, goal LP
    if (!do_traj) {
        exit_stores(0, false, 0, 0, 0, 0, false, 0.0);
        return;
    }

    // ================================ trajectory QP =============================================
    double *base = reinterpret_cast<double *>(mem);
    // layout (make_layout): compile-time for static shapes without slack obstacles, else from the arguments
#ifndef DQ_STAGE_Z
#define DQ_STAGE_Z 1
#endif
#ifndef DQ_STAGE_S
#define DQ_STAGE_S 1
#endif
#ifndef DQ_STAGE_H
#define DQ_STAGE_H 1
#endif
    constexpr int FB = FS > 0 ? FS : 8;  // block of LDS reads issued together in the O(r) loops
    constexpr bool NREG = n_in_registers(FS, MS);  // lane p holds row p of N_T in registers, no LDS copy
    constexpr bool ST = DIMS > 0 && SLK == 0;
    constexpr int R_ST = DIMS * FS, LDQ_ST = (DIMS * FS) | 1;
    static_assert(!ST || LDQ_ST >= 32, "the spare row of Q holds the 32 dump slots");
    const int r_max = ST ? R_ST : lay.r_max;
    const int ldq = ST ? LDQ_ST : lay.ldq;
    const int oQ = 0;
    const int oRp = ST ? (R_ST + 1) * LDQ_ST : lay.oRp;
    const int oN = ST ? (R_ST + 1) * LDQ_ST + R_ST * (R_ST + 1) / 2 : lay.oN;
    const int oX = ST ? (R_ST + 1) * LDQ_ST + R_ST * (R_ST + 1) / 2 + (NREG ? 0 : MS * 6 * FS) : lay.oX;
    const int odump = ST ? R_ST * LDQ_ST : lay.oDump;  // 32 dump slots in the spare row of Q, base[odump + (lane & 31)]
    const int oBlo = ST ? (R_ST + 1) * LDQ_ST + R_ST * (R_ST + 1) / 2 + (NREG ? 0 : MS * 6 * FS) + DIMS * MS * 6 : lay.oBlo;
    const int oBhi = lay.oBhi;
    double *Q = base + oQ, *Rp = S_EXT ? s_ext : base + oRp, *Nl = base + oN, *x = base + oX;
    double *blo = base + oBlo, *bhi = base + oBhi;
    unsigned char *cold_flag = mem + lay.oFlag;
    const int rc = dim * f;      // whitened control-point dimension
    const int r = rc + nslk;     // whitened dimension
    const int nxc = dim * P;     // control-point variables
    const int cold_rows = K - Kc;
    const int qdummy = r_max;  // spare row of Q for the lanes beyond r
    const int odump_rp = S_EXT ? r_max * (r_max + 1) / 2 : odump - oRp;  // the dump slots as indices of Rp

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
    if (r > 64 * RC || dim != kp.dim || f != kp.f || M != kp.M || n1 != kp.n1) status = DLSCQP_NUMERICAL;  // the launcher picks the instantiation; defensive
    if (status != DLSCQP_OK) {
        exit_stores(status, true, 0, 0, 0, 0, false, 0.0);
        return;
    }

    const double *tN = kp.tabN + (size_t)(T - 1) * P * f;
    const double *tXs = kp.tabX0s + (size_t)(T - 1) * P * 3;
    const double *tXg = kp.tabX0g + (size_t)(T - 1) * P;
    const double *tNN = kp.tabNN + (size_t)(T - 1) * P;
    const double *tSIW = kp.tabSIW + (size_t)(T - 1) * kp.n_slots;

    // ---- unconstrained minimiser and bound table into LDS -------------------------------------------
    // (uniform trip counts; a lane past the end computes a clamped element and stores it to the dump slot)
    RegD ciw[KU], siw[SU], nnr[KU];
    constexpr int NXP = DIMS > 0 ? (DIMS * MS * 6 + 63) / 64 : 1;
    if (DIMS > 0) {
        // static shapes: every table read of this phase (x0 factors, slot scalings, row scalings: L2 round trips) is
        // issued before the first use, so the phase pays one round trip instead of one per table
        w.each([&](int l) {
            double a0[NXP], a1[NXP], a2[NXP], ag[NXP], sv[SU], nn[KU];
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const int j = 64 * u + l;
                const int jc = j < nxc ? j : nxc - 1;
                const int p = jc - (jc / P) * P;
                a0[u] = tXs[p * 3 + 0]; a1[u] = tXs[p * 3 + 1]; a2[u] = tXs[p * 3 + 2]; ag[u] = tXg[p];
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int s = l + 64 * u;
                sv[u] = tSIW[s < Sc ? s : (Sc > 0 ? Sc - 1 : 0)];
            }
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const uint32_t key = ckey[u].at(l);
                const bool have = key != KEY_SKIP;
                const int mi = have ? (int)(key & 0xFF) * n1 + (int)((key >> 8) & 0xF) : 0;
                nn[u] = tNN[mi < P ? mi : 0];
            }
#pragma unroll
            for (int u = 0; u < NXP; u++) { DQ_KEEP(a0[u]); DQ_KEEP(a1[u]); DQ_KEEP(a2[u]); DQ_KEEP(ag[u]); }
#pragma unroll
            for (int u = 0; u < SU; u++) DQ_KEEP(sv[u]);
#pragma unroll
            for (int u = 0; u < KU; u++) DQ_KEEP(nn[u]);
#pragma unroll
            for (int u = 0; u < NXP; u++) {
                const int j = 64 * u + l;
                const bool in = j < nxc;
                const int k = (in ? j : nxc - 1) / P;
                const double v = a0[u] * (double)pick3(pos, k) + a1[u] * (double)pick3(vel, k) + a2[u] * (double)pick3(acc, k) +
                                 ag[u] * (double)pick3(goal, k);
                base[blend_idx(in, oX + j, odump + (l & 31))] = v;
            }
#pragma unroll
            for (int u = 0; u < SU; u++) siw[u].at(l) = (l + 64 * u < Sc) ? sv[u] : 0.0;
#pragma unroll
            for (int u = 0; u < KU; u++) nnr[u].at(l) = nn[u];
        });
    }
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
            const double hv = max_vel * kp.vel_scale, ha = max_acc * kp.acc_scale;
            const double hc = 0.5 * kp.comm_range - radius;  // traj_optimizer.cpp:495-500
            const int kw = bc - (dim * M + 3);
            const double cw = 0.5 * kp.comm_range - SP_EPSILON_FLOAT;  // traj_optimizer.cpp:507-510
            const double wpk = (double)pick3(wp, kw < 0 ? 0 : kw);
            double lo = wpk - cw, hi = wpk + cw;
            lo = bc == dim * M + 2 ? -hc : lo; hi = bc == dim * M + 2 ? hc : hi;
            lo = bc == dim * M + 1 ? -ha : lo; hi = bc == dim * M + 1 ? ha : hi;
            lo = bc == dim * M ? -hv : lo;     hi = bc == dim * M ? hv : hi;
            lo = is_box ? blo_box : lo;        hi = is_box ? bhi_box : hi;
            base[blend_idx(in, oBlo + b, odump + (l & 31))] = lo;
            base[blend_idx(in, oBhi + b, odump + (l & 31))] = hi;
        }
    });
    // finish the LSC cache: rhs = d + n.p (traj_optimizer.cpp:427-439), drop rows that are not in the model
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
            double sg = kp.sigma[m < M ? m : 0];
            DQ_KEEP(sg);
            wsq = sl != 0xFFFFu ? wsq + sg * sg : wsq;
            DQ_KEEP(rhs);
            cd[u].at(l) = in_model ? rhs : cd[u].at(l);
            ckey[u].at(l) = in_model ? key2 : KEY_SKIP;
            // the SQUARED whitened length for now: 1 / sqrt of it (a square root and a division, ~70 instructions a row) is
            // taken when the first pricing pass finds a violated cached row -- for most agents never
            ciw[u].at(l) = (in_model & (wsq > 0)) ? wsq : 0.0;
        });
    }
    w.sync();
    w.stamp(1);  // slot cache, x0, bounds

    // per-lane state of the method
    RegU fl_lsc, fl_slo, fl_shi, fl_slk;  // "skip" bits of the cached rows: active or accepted-infeasible
    RegD na[RC], dv[RC], zv[RC], rv[RC], uu[RC];
    RegI act[RC];
    w.each([&](int l) {
        fl_lsc.at(l) = 0; fl_slo.at(l) = 0; fl_shi.at(l) = 0; fl_slk.at(l) = 0;
#pragma unroll
        for (int c = 0; c < RC; c++) {
            uu[c].at(l) = 0; act[c].at(l) = -1; rv[c].at(l) = 0;
            na[c].at(l) = 0; dv[c].at(l) = 0; zv[c].at(l) = 0;
        }
    });

    int q = 0, iters = 0, n_add = 0, n_drop = 0;
    bool factors_ready = false;
    bool ciw_ready = false;  // ciw holds 1 / whitened length (true) or still the squared length (false)
    // The Householder update of Q that an added row asks for is postponed until Q is read again: the last row an agent
    // adds is usually followed by a pricing pass that finds nothing, and then the O(r^2) sweep is never needed.
    bool q_pending = false;
    bool q_identity = true;  // Q has not been written yet: it is the identity (d = na, z = d2), materialised with the first update
    int q_pend_col = 0;
    double q_pend_alpha = 0.0, q_pend_beta = 0.0;
    RegD q_pend_v[RC], q_pend_z[RC];
    w.each([&](int l) {
#pragma unroll
        for (int c = 0; c < RC; c++) { q_pend_v[c].at(l) = 0.0; q_pend_z[c].at(l) = 0.0; }
    });
    RegD nrow[NREG ? FB : 1];  // NREG: row (lane) of N_T, loaded with the factors
    w.each([&](int l) { for (int u = 0; u < (NREG ? FB : 1); u++) nrow[u].at(l) = 0.0; });
    const int max_iter = kp.max_iter > 0 ? kp.max_iter : 20 * (r + 8) + 200;
    // unified row codes: [0,K) LSC rows, K + 2 s + side structured slots, K + 2 n_slots + s slack bounds
    const int code_slack0 = K + 2 * n_slots;

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

    auto set_skip = [&](int code, bool on) {  // code and on are uniform; the owner lane is picked by a select
        auto upd = [&](RegU &fr, int owner, int u) {
            w.each([&](int l) {
                const uint32_t cur = fr.at(l), bit = 1u << u;
                const uint32_t nv = on ? (cur | bit) : (cur & ~bit);
                fr.at(l) = l == owner ? nv : cur;
            });
        };
        if (code < Kc) {
            upd(fl_lsc, code & 63, code >> 6);
        } else if (code < K) {
            w.each([&](int) { cold_flag[code - Kc] = on ? 1 : 0; });  // every lane, same byte, same value
        } else if (code < code_slack0) {
            const int s = (code - K) >> 1, side = (code - K) & 1;
            if (ALL_CACHED || s < Sc) {
                if (side) upd(fl_shi, s & 63, s >> 6);
                else upd(fl_slo, s & 63, s >> 6);
            } else {
                w.each([&](int) { cold_flag[cold_rows + 2 * (s - Sc) + side] = on ? 1 : 0; });
            }
        } else {
            const int s = code - code_slack0;
            upd(fl_slk, s & 63, s >> 6);
        }
        w.sync();
    };

    auto apply_pending_q = [&]() {
        const int qc = q_pend_col;
        if (q_identity) {  // uniform: first update, Q = I first (and zeros in the spare row qdummy that lanes beyond r work on)
            w.each([&](int l0) {
                int l = l0;
                DQ_KEEP(l);  // addresses computed here, not hoisted out of the iteration loop (see the factor set-up)
                const int nq = (r_max + 1) * ldq;
                for (int j0 = 0; j0 < nq; j0 += 64) {
                    const int j = j0 + l;
                    const int i = j / ldq, c = j - i * ldq;
                    base[blend_idx(j < nq, oQ + j, odump + (l & 31))] = ((i == c) & (i < r)) ? 1.0 : 0.0;
                }
            });
            w.sync();
            q_identity = false;
        }
                w.each([&](int l) {
#pragma unroll
            for (int c = 0; c < RC; c++) {
                const int i = l + 64 * c;
                double *qi = Q + (i < r ? i : qdummy) * ldq;  // lanes beyond r update the spare row
                const double gi = -q_pend_beta * (q_pend_z[c].at(l) - q_pend_alpha * qi[qc]);
#if DQ_STAGE_H
                for (int j0 = 0; j0 < rc; j0 += FB) {  // Q[i][:] += gi v', in blocks of FB columns
                    if (j0 + FB <= qc) continue;  // uniform: v is zero in the first q columns
                    const int nv = rc - j0 < FB ? rc - j0 : FB;  // uniform; == FB for static shapes
                    double tq[FB];
#pragma unroll
                    for (int u = 0; u < FB; u++) tq[u] = qi[j0 + (u < nv ? u : 0)];
#pragma unroll
                    for (int u = 0; u < FB; u++) DQ_KEEP(tq[u]);
#pragma unroll
                    for (int u = 0; u < FB; u++) tq[u] = fma(gi, DQ_BC(q_pend_v, j0 + (u < nv ? u : 0)), tq[u]);
#pragma unroll
                    for (int u = 0; u < FB; u++)
                        if (u < nv) qi[j0 + u] = tq[u];  // uniform condition
                }
#else
#pragma unroll
                for (int j = 0; j < rc; j++) qi[j] = fma(gi, DQ_BC(q_pend_v, j), qi[j]);  // Q[i][:] += gi v'
#endif

                for (int j = rc; j < r; j++) qi[j] = fma(gi, DQ_BC(q_pend_v, j), qi[j]);
            }
        });
        w.sync();
        q_pending = false;
    };

    for (;;) {
        // ---- most violated inactive row ---------------------------------------------------------
        RegD bestv, bests;
        RegI bestc;
        // cached rows: ALL LDS reads of the pass first and unconditionally (DQ_KEEP), then select-based arithmetic -- no
        // branches, and one exposed LDS round trip per pass instead of one per row.
        RegD lxr[KU][4], sxr[SU][5], lvr[KU];
        RegI lsc_viol;
        w.each([&](int l) {
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const uint32_t key = ckey[u].at(l);
                const bool in_model = key != KEY_SKIP;
                const int m = key & 0xFF, i = (key >> 8) & 0xF;
                const int p = in_model ? m * n1 + i : 0;
                const uint32_t sl = key >> 16;
                const bool has_slack = in_model && sl != 0xFFFFu;
                lxr[u][0].at(l) = x[p];
                lxr[u][1].at(l) = x[P + p];
                lxr[u][2].at(l) = dim == 3 ? x[2 * P + p] : 0.0;
                lxr[u][3].at(l) = SLK == 0 ? 0.0 : x[nxc + (has_slack ? (int)sl * M + m : 0)];
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const uint32_t pi = sidx[u].at(l), pc = scb[u].at(l);
                int i0 = pi & 0x3FF, i1 = (pi >> 10) & 0x3FF, i2 = (pi >> 20) & 0x3FF;
                i0 = i0 == 0x3FF ? 0 : i0;  // unused entries carry coefficient 0
                i1 = i1 == 0x3FF ? 0 : i1;
                i2 = i2 == 0x3FF ? 0 : i2;
                const int bsel = pc >> 24;
                sxr[u][0].at(l) = x[i0]; sxr[u][1].at(l) = x[i1]; sxr[u][2].at(l) = x[i2];
                sxr[u][3].at(l) = blo[bsel]; sxr[u][4].at(l) = bhi[bsel];
            }
#pragma unroll
            for (int u = 0; u < KU; u++) { DQ_KEEP(lxr[u][0].at(l)); DQ_KEEP(lxr[u][1].at(l)); DQ_KEEP(lxr[u][2].at(l)); DQ_KEEP(lxr[u][3].at(l)); }
#pragma unroll
            for (int u = 0; u < SU; u++) {
                DQ_KEEP(sxr[u][0].at(l)); DQ_KEEP(sxr[u][1].at(l)); DQ_KEEP(sxr[u][2].at(l)); DQ_KEEP(sxr[u][3].at(l)); DQ_KEEP(sxr[u][4].at(l));
            }
            bool violated = false;
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const uint32_t key = ckey[u].at(l);
                const bool in_model = key != KEY_SKIP;
                const uint32_t sl = key >> 16;
                const bool has_slack = in_model && sl != 0xFFFFu;
                const bool valid = in_model && !((fl_lsc.at(l) >> u) & 1);
                double val = (double)cnx[u].at(l) * lxr[u][0].at(l);
                val = fma((double)cny[u].at(l), lxr[u][1].at(l), val);
                if (dim == 3) val = fma((double)cnz[u].at(l), lxr[u][2].at(l), val);
                val -= has_slack ? lxr[u][3].at(l) : 0.0;
                const double v = valid ? cd[u].at(l) - val : -1.0;
                lvr[u].at(l) = v;
                violated |= v > VIOL_TOL;
            }
            lsc_viol.at(l) = violated ? 1 : 0;
        });
        // pivot scaling of the cached rows: 1 / sqrt(squared whitened length), taken when a pass first finds one of them
        // violated (wave-uniform branch, at most once per agent; most agents never get here)
        if (!ciw_ready && w.any(lsc_viol)) {
            w.each([&](int l) {
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    const double wsq = ciw[u].at(l);
                    double iw = 1.0 / sqrt(wsq > 0 ? wsq : 1.0);
                    DQ_KEEP(iw);
                    ciw[u].at(l) = wsq > 0 ? iw : 0.0;
                }
            });
            ciw_ready = true;
        }
        w.each([&](int l) {
            double bv = 0.0, bs = 0.0;  // violation and pivot score (violation / whitened row length) of the lane's best row
            int bc = -1;
            double sx[SU][5];
#pragma unroll
            for (int u = 0; u < SU; u++)
                for (int t = 0; t < 5; t++) sx[u][t] = sxr[u][t].at(l);
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const double v = lvr[u].at(l);
                const double sc = v > VIOL_TOL ? v * ciw[u].at(l) : -1.0;  // scaling not taken yet: no lane has v > VIOL_TOL
                bc = sc > bs ? l + 64 * u : bc;
                bv = sc > bs ? v : bv;
                bs = sc > bs ? sc : bs;
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const uint32_t pi = sidx[u].at(l), pc = scb[u].at(l);
                const bool valid = !(pi & 0x80000000u);
                double val = (double)(int8_t)(pc & 0xFF) * sx[u][0];
                val = fma((double)(int8_t)((pc >> 8) & 0xFF), sx[u][1], val);
                val = fma((double)(int8_t)((pc >> 16) & 0xFF), sx[u][2], val);
                const double vlo = (valid && !((fl_slo.at(l) >> u) & 1)) ? sx[u][3] - val : -1.0;
                const double vhi = (valid && !((fl_shi.at(l) >> u) & 1)) ? val - sx[u][4] : -1.0;
                const int s = l + 64 * u;
                const double slo = vlo > VIOL_TOL ? vlo * siw[u].at(l) : -1.0;
                bc = slo > bs ? K + 2 * s : bc;
                bv = slo > bs ? vlo : bv;
                bs = slo > bs ? slo : bs;
                const double shi = vhi > VIOL_TOL ? vhi * siw[u].at(l) : -1.0;
                bc = shi > bs ? K + 2 * s + 1 : bc;
                bv = shi > bs ? vhi : bv;
                bs = shi > bs ? shi : bs;
            }
            for (int s0 = 0; s0 < nslk; s0 += 64) {  // eps <= 0 (traj_optimizer.cpp:276); uniform trip count
                const int s = s0 + l;
                const bool in = s < nslk;
                const int sc2 = in ? s : 0;
                double v = x[nxc + sc2];
                double sg = kp.sigma[sc2 % M];
                DQ_KEEP(v); DQ_KEEP(sg);
                const bool cand = in & (v > VIOL_TOL) & (sg > 0) & !((fl_slk.at(l) >> (s0 >> 6)) & 1);
                double qv = v / (sg > 0 ? sg : 1.0);
                DQ_KEEP(qv);
                const double sc = cand ? qv : -1.0;
                bc = sc > bs ? code_slack0 + s : bc;
                bv = sc > bs ? v : bv;
                bs = sc > bs ? sc : bs;
            }
            // rows beyond the register cache: straight from HBM (K > 64 KU), CB blocks of 64 rows per trip: first every
            // row's record, then the loads that depend on it (scaling, slack variable), then the arithmetic
            if (!ALL_CACHED)
            for (int j0 = Kc; j0 < K; j0 += 64 * CB) {
                uint32_t key[CB];
                float g[CB][6];
                double gd[CB];
                int cf0[CB];
#pragma unroll
                for (int u = 0; u < CB; u++) {
                    const int j = j0 + 64 * u + l;
                    const int jc = j < K ? j : K - 1;
                    key[u] = Rkey[jc];
                    g[u][0] = Rnx[jc]; g[u][1] = Rny[jc]; g[u][2] = Rnz[jc];
                    g[u][3] = Rpx[jc]; g[u][4] = Rpy[jc]; g[u][5] = Rpz[jc];
                    gd[u] = Rd[jc];
                    cf0[u] = cold_flag[jc - Kc];
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
                    inm[u] = (j < K) & (cf0[u] == 0) & !((m == 0) & (i < kp.phi)) & !f32_norm3_below_eps(g[u][0], g[u][1], g[u][2]);
                    const int p = inm[u] ? m * n1 + i : 0;
                    const uint32_t sl = key[u] >> 16;
                    hsl[u] = inm[u] & (sl != 0xFFFFu) & ((int)sl < nso);
                    xv[u][0] = x[p]; xv[u][1] = x[P + p]; xv[u][2] = x[(dim == 3 ? 2 : 0) * P + p];
                    xv[u][3] = x[nxc + (hsl[u] ? (int)sl * M + m : 0)];
                    tn[u] = tNN[p];
                    sg[u] = kp.sigma[inm[u] ? m : 0];
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
                    val = dim == 3 ? val3 : val;
                    rhs = dim == 3 ? rhs3 : rhs;
                    val = hsl[u] ? val - xv[u][3] : val;
                    const double v = rhs - val;
                    double wsq = ((double)nx * (double)nx + (double)ny * (double)ny + (dim == 3 ? (double)nz * (double)nz : 0.0)) * tn[u];
                    wsq = hsl[u] ? wsq + sg[u] * sg[u] : wsq;
                    double qv = v / sqrt(wsq > 0 ? wsq : 1.0);
                    DQ_KEEP(qv);
                    const double sc = (inm[u] & (v > VIOL_TOL) & (wsq > 0)) ? qv : -1.0;
                    bc = sc > bs ? j : bc;
                    bv = sc > bs ? v : bv;
                    bs = sc > bs ? sc : bs;
                }
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
                const double vlo = blo[sl.bsel] - val, vhi = val - bhi[sl.bsel];
                const double iw = tSIW[sc2];
                const int fo = cold_rows + 2 * (sc2 - Sc);
                int f_lo = cold_flag[fo], f_hi = cold_flag[fo + 1];
                DQ_KEEP(f_lo); DQ_KEEP(f_hi);
                const double slo = (in & (vlo > VIOL_TOL) & (f_lo == 0)) ? vlo * iw : -1.0;
                bc = slo > bs ? K + 2 * s : bc;
                bv = slo > bs ? vlo : bv;
                bs = slo > bs ? slo : bs;
                const double shi = (in & (vhi > VIOL_TOL) & (f_hi == 0)) ? vhi * iw : -1.0;
                bc = shi > bs ? K + 2 * s + 1 : bc;
                bv = shi > bs ? vhi : bv;
                bs = shi > bs ? shi : bs;
            }
            bestv.at(l) = bv;
            bests.at(l) = bs;
            bestc.at(l) = bc;
        });
        const double top = w.max(bests);  // best pivot score over all violated rows
        w.stamp(2);                       // row scan + arg-max
        if (!(top > 0.0)) break;          // no row violated by more than VIOL_TOL: optimal
        // ---- infeasibility by weak duality ------------------------------------------------------
        // x minimises the objective over the active rows with non-negative multipliers, so f(x) is a lower bound
        // of the optimum.  Every feasible point obeys the box and acceleration rows, which bound the objective
        // from above by f_max (objective_upper_bound).  f(x) > f_max therefore proves the QP infeasible, usually
        // after 5-20 rows instead of the ~r rows the active set needs to run out of dimensions (the launch is as
        // slow as its slowest agent, and those were the infeasible ones).  Checked every second pricing pass from the
        // eighth on.
        n_price++;
        if (kp.wd_cut && nslk == 0 && n_price >= 8 && (n_price & 1) == 0) {
            if (n_price == 8) f_max = objective_upper_bound();  // first check of this agent
            const double f_now = objective();
            DQ_TRACE(4, f_now, f_max, (double)n_price, (double)q);
            if (f_now > f_max + 1.0) {  // margin 1 (about 1 % of f_max with the reference's weights); f(x) grows by orders of magnitude
                status = DLSCQP_INFEASIBLE;
                break;
            }
        }
        RegI is_top;
        w.each([&](int l) { is_top.at(l) = (bests.at(l) == top && bestc.at(l) >= 0) ? 1 : 0; });
        const int owner_lane = w.first_lane(is_top);
        const int ip = w.bcast(bestc, owner_lane);
        double viol = w.bcast(bestv, owner_lane);  // = -(a.x - b) of the chosen row
        DQ_TRACE(1, (double)ip, viol, top, (double)q);
#ifdef DLSCQP_TRACE
        DQ_TRACE(3, objective(), 0.0, 0.0, 0.0);
#endif

        if (!factors_ready) {
            // first violated row: set up the factors (agents that never get here skip all of this)
            w.each([&](int l0) {
                // the lane index through an opaque copy: the addresses below must be computed here, not hoisted
                // out of the iteration loop (where ~25 loop-invariant address registers were kept alive in scratch)
                int l = l0;
                DQ_KEEP(l);
                if (!NREG) {
                    for (int j0 = 0; j0 < P * f; j0 += 64) {
                        const int j = j0 + l;
                        const bool in = j < P * f;
                        const double v = tN[in ? j : P * f - 1];
                        base[blend_idx(in, oN + j, odump + (l & 31))] = v;
                    }
                }
            });
            if (NREG) {
                w.each([&](int l) {
                    const int pc = l < P ? l : P - 1;
#pragma unroll
                    for (int u = 0; u < (NREG ? FB : 1); u++) nrow[u].at(l) = tN[pc * f + u];
                });
            }
            w.sync();
            factors_ready = true;
            w.stamp(1);
        }

        if (q_pending) apply_pending_q();  // uniform
        // ---- decode the row into sparse x-space entries (uniform) -------------------------------
        int ek[3] = {-1, -1, -1}, ep[3] = {0, 0, 0}, sidx_row = -1;
        double cf[3] = {0, 0, 0}, row_scale = 1.0;
        int axes_mask = 0;
        if (ip < K) {
            uint32_t key = 0;
            float nn[3] = {0, 0, 0};
            if (ALL_CACHED || ip < Kc) {
                const int own = ip & 63, u = ip >> 6;
#pragma unroll
                for (int uq = 0; uq < KU; uq++) {  // uniform selects, no branches
                    const uint32_t kq = w.bcast(ckey[uq], own);
                    const float a0 = w.bcast(cnx[uq], own), a1 = w.bcast(cny[uq], own), a2 = w.bcast(cnz[uq], own);
                    key = uq == u ? kq : key;
                    nn[0] = uq == u ? a0 : nn[0]; nn[1] = uq == u ? a1 : nn[1]; nn[2] = uq == u ? a2 : nn[2];
                }
            } else {
                key = Rkey[ip];
                nn[0] = Rnx[ip]; nn[1] = Rny[ip]; nn[2] = Rnz[ip];
                uint32_t sl = key >> 16;
                if (sl != 0xFFFFu && (int)sl >= nso) key |= 0xFFFF0000u;
            }
            const int m = key & 0xFF, i = (key >> 8) & 0xF;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (k < dim) {
                    ek[k] = k; ep[k] = m * n1 + i; cf[k] = (double)nn[k];
                    axes_mask |= 1 << k;
                }
            }
            const uint32_t sl = key >> 16;
            if (sl != 0xFFFFu) sidx_row = (int)sl * M + m;
        } else if (ip < code_slack0) {
            const int s = (ip - K) >> 1, side = (ip - K) & 1;
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
#pragma unroll
            for (int t = 0; t < 3; t++) {  // fixed positions, unused entries keep ek = -1 / cf = 0
                const bool used = sidx3[t] >= 0;
                const int k = used ? sidx3[t] / P : -1;
                ek[t] = k;
                ep[t] = used ? sidx3[t] - k * P : 0;
                cf[t] = used ? sg * (double)scoef3[t] : 0.0;
                axes_mask |= used ? (1 << k) : 0;
            }
            // back to the reference's row scaling: n/dt for velocity rows, n(n-1)/dt^2 for acceleration rows
            row_scale = bs == dim * M ? 1.0 / kp.vel_scale : (bs == dim * M + 1 ? 1.0 / kp.acc_scale : 1.0);
        } else {
            sidx_row = ip - code_slack0;
        }
        // whitened normal na = N' a  (slack part: -sigma), one element per lane
        w.each([&](int l) {
#pragma unroll
            for (int c = 0; c < RC; c++) {
                const int e = l + 64 * c;
                double v = 0;
                const bool in_ctrl = e < rc;
                const int ec = in_ctrl ? e : 0;
                const int k = ec / f, j = ec - k * f;
                double nv[3];  // the three loads first (NREG: L2 round trips), then their uses
#pragma unroll
                for (int t = 0; t < 3; t++) nv[t] = NREG ? tN[ep[t] * f + j] : Nl[ep[t] * f + j];  // ep[t] is a valid row of N for unused entries too (0)
#pragma unroll
                for (int t = 0; t < 3; t++) DQ_KEEP(nv[t]);
#pragma unroll
                for (int t = 0; t < 3; t++) v = (in_ctrl & (ek[t] == k)) ? fma(cf[t], nv[t], v) : v;
                double sgm = -kp.sigma[(sidx_row < 0 ? 0 : sidx_row) % M];
                DQ_KEEP(sgm);
                v = (!in_ctrl & (e - rc == sidx_row)) ? sgm : v;
                na[c].at(l) = v;
            }
        });
        w.stamp(3);  // decode + whitened normal

        double uq = 0.0;
        bool resolved = false;
        while (!resolved) {
            if (++iters > max_iter) { status = DLSCQP_ITERATION_LIMIT; break; }
            // d = Q' na   (na is non-zero only in the touched axis blocks and one slack entry)
            if (q_identity) {  // uniform
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) dv[c].at(l) = l + 64 * c < r ? na[c].at(l) : 0.0;
                });
            } else
            w.each([&](int l) {
#pragma unroll
                for (int c = 0; c < RC; c++) {
                    const int j = l + 64 * c;
                    double a0 = 0.0, a1 = 0.0;
                    const int jc = j < r ? j : 0;  // lanes beyond r read column 0 and drop the sum
#pragma unroll
                    for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++) {
                        if (k < dim && (axes_mask & (1 << k))) {  // uniform
                            // blocks of FB rows: all LDS reads of a block are issued before the first use
                            for (int i0 = 0; i0 < f; i0 += FB) {
                                const int nv = f - i0 < FB ? f - i0 : FB;  // uniform; == FB for static shapes
                                double tq[FB];
#pragma unroll
                                for (int u = 0; u < FB; u++) tq[u] = Q[(k * f + i0 + (u < nv ? u : 0)) * ldq + jc];
#pragma unroll
                                for (int u = 0; u < FB; u++) DQ_KEEP(tq[u]);
#pragma unroll
                                for (int u = 0; u < FB; u += 2) {
                                    if (u < nv) a0 = fma(tq[u], DQ_BC(na, k * f + i0 + u), a0);
                                    if (u + 1 < FB && u + 1 < nv) a1 = fma(tq[u + 1], DQ_BC(na, k * f + i0 + u + 1), a1);
                                }
                            }
                        }
                    }
                    if (sidx_row >= 0) {  // uniform
                        const int i = rc + sidx_row;
                        a0 = fma(Q[i * ldq + jc], DQ_BC(na, i), a0);
                    }
                    dv[c].at(l) = j < r ? a0 + a1 : 0.0;
                }
            });
            w.stamp(4);  // d = Q' na
            // z = Q2 d2, |d2|^2, |d|^2.  d2 = d with the first q entries zeroed, as a register of its own, so
            // that the unrolled loop below is select- and branch-free (its LDS reads then issue back to back)
            RegD dz[RC];
            w.each([&](int l) {
#pragma unroll
                for (int c = 0; c < RC; c++) dz[c].at(l) = (l + 64 * c >= q) ? dv[c].at(l) : 0.0;
            });
            if (q_identity) {  // uniform: Q2 = I (q = 0)
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) zv[c].at(l) = dz[c].at(l);
                });
            } else
            w.each([&](int l) {
#pragma unroll
                for (int c = 0; c < RC; c++) {
                    const int i = l + 64 * c;
                    double a0 = 0.0, a1 = 0.0;
                    const double *qi = Q + (i < r ? i : qdummy) * ldq;  // lanes beyond r: the spare row
#if DQ_STAGE_Z
                    for (int j0 = 0; j0 < rc; j0 += FB) {
                        if (j0 + FB <= q) continue;  // uniform: d2 is zero in the first q columns
                        const int nv = rc - j0 < FB ? rc - j0 : FB;  // uniform; == FB for static shapes
                        double tq[FB];
#pragma unroll
                        for (int u = 0; u < FB; u++) tq[u] = qi[j0 + (u < nv ? u : 0)];
#pragma unroll
                        for (int u = 0; u < FB; u++) DQ_KEEP(tq[u]);
#pragma unroll
                        for (int u = 0; u < FB; u += 2) {
                            if (u < nv) a0 = fma(tq[u], DQ_BC(dz, j0 + u), a0);
                            if (u + 1 < FB && u + 1 < nv) a1 = fma(tq[u + 1], DQ_BC(dz, j0 + u + 1), a1);
                        }
                    }
#else
#pragma unroll
                    for (int j = 0; j < rc; j += 2) {
                        a0 = fma(qi[j], DQ_BC(dz, j), a0);
                        if (j + 1 < rc) a1 = fma(qi[j + 1], DQ_BC(dz, j + 1), a1);
                    }
#endif

                    for (int j = rc; j < r; j++) a0 = fma(qi[j], DQ_BC(dz, j), a0);  // slack columns
                    zv[c].at(l) = i < r ? a0 + a1 : 0.0;
                }
            });
            RegD s_d2, s_all;
            w.each([&](int l) {
                double a2 = 0, aa = 0;
#pragma unroll
                for (int c = 0; c < RC; c++) {
                    const int j = l + 64 * c;
                    const double t = dv[c].at(l) * dv[c].at(l);  // dv is zero beyond r
                    aa += t;
                    a2 += j >= q ? t : 0.0;
                }
                s_d2.at(l) = a2;
                s_all.at(l) = aa;
            });
            double dd2, dall;
            w.sum2(s_d2, s_all, dd2, dall);
            w.stamp(5);  // z, |d2|^2, |d|^2
            // rv = R^{-1} d1 = S d1: lane i accumulates row i of S; no dependence between lanes
            {
                RegD dm[RC];
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) dm[c].at(l) = (l + 64 * c < q) ? dv[c].at(l) : 0.0;
                });
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        double a0 = 0.0, a1 = 0.0;
                        int k = 0;
                        for (; k + 1 < q; k += 2) {  // S(i,k) = 0 below the diagonal (packed storage: mask it)
                            const double s0 = DQ_RP(i <= k ? i : 0, k), s1 = DQ_RP(i <= k + 1 ? i : 0, k + 1);
                            a0 = fma(i <= k ? s0 : 0.0, DQ_BC(dm, k), a0);
                            a1 = fma(i <= k + 1 ? s1 : 0.0, DQ_BC(dm, k + 1), a1);
                        }
                        if (k < q) {
                            const double s0 = DQ_RP(i <= k ? i : 0, k);
                            a0 = fma(i <= k ? s0 : 0.0, DQ_BC(dm, k), a0);
                        }
                        rv[c].at(l) = a0 + a1;
                    }
                });
            }
            w.stamp(6);  // back substitution
            // step lengths: t1 = min u_k / r_k over r_k > 0 (dual), t2 = viol / |d2|^2 (primal)
            double t1 = 1e300;
            int ldrop = -1;
            if (q > 0) {
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
            const bool dependent = !(dd2 > DEP_TOL * dall) || !(dd2 > 0);
            const double t2 = dependent ? 1e300 : viol / dd2;
            DQ_TRACE(2, dd2, dall, t1, (double)ldrop);
            w.stamp(7);  // ratio test
            if (t1 >= 1e300 && t2 >= 1e300) {
                // no step possible: infeasible, unless within CPLEX's feasibility tolerance
                if (viol * row_scale <= CPLEX_FEAS_TOL) {
                    set_skip(ip, true);
                    resolved = true;
                    break;
                }
                status = DLSCQP_INFEASIBLE;
                break;
            }
            const double t = t1 < t2 ? t1 : t2;
            if (!dependent) {
                // primal step  x += t N z  (a row of N is read once and used for every axis)
                w.each([&](int l) {
                    for (int p0 = 0; p0 < P; p0 += 64) {  // uniform trip count
                        const int p = p0 + l;
                        const bool in = p < P;
                        const int pc = in ? p : P - 1;
                        double sk[3] = {0.0, 0.0, 0.0};
#if DQ_STAGE_S
                        if (NREG) {  // row p of N_T is in this lane's registers (p = lane: P <= 64)
#pragma unroll
                            for (int u = 0; u < (NREG ? FB : 1); u++)
#pragma unroll
                                for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++)
                                    if (k < dim) sk[k] = fma(nrow[NREG ? u : 0].at(l), DQ_BC(zv, k * f + u), sk[k]);
                        } else
                        for (int c0 = 0; c0 < f; c0 += FB) {
                            const int nvb = f - c0 < FB ? f - c0 : FB;  // uniform; == FB for static shapes
                            double tn[FB];
#pragma unroll
                            for (int u = 0; u < FB; u++) tn[u] = Nl[pc * f + c0 + (u < nvb ? u : 0)];
#pragma unroll
                            for (int u = 0; u < FB; u++) DQ_KEEP(tn[u]);
#pragma unroll
                            for (int u = 0; u < FB; u++) {
                                if (u < nvb) {  // uniform
#pragma unroll
                                    for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++)
                                        if (k < dim) sk[k] = fma(tn[u], DQ_BC(zv, k * f + c0 + u), sk[k]);
                                }
                            }
                        }
#else
#pragma unroll
                        for (int c = 0; c < f; c++) {
                            const double nv = Nl[pc * f + c];
#pragma unroll
                            for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++)
                                if (k < dim) sk[k] = fma(nv, DQ_BC(zv, k * f + c), sk[k]);
                        }
#endif

#pragma unroll
                        for (int k = 0; k < (DIMS > 0 ? DIMS : 3); k++)
                            if (k < dim) {  // uniform
                                const double xn = fma(t, sk[k], x[k * P + pc]);
                                base[blend_idx(in, oX + k * P + p, odump + (l & 31))] = xn;
                            }
                    }
                });
                if (nslk > 0) {  // uniform
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int e = l + 64 * c;
                            const bool in = e >= rc && e < r;
                            const int s = in ? e - rc : 0;
                            const double xn = fma(t * kp.sigma[s % M], zv[c].at(l), x[nxc + s]);
                            base[blend_idx(in, oX + nxc + s, odump + (l & 31))] = xn;
                        }
                    });
                }
                viol -= t * dd2;
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
                // ---- add row ip: Householder reflection of d2 onto e_1 ---------------------------
                const double dq = DQ_BC(dv, q);
                const double nrm = sqrt(dd2);
                const double alpha = dq > 0 ? -nrm : nrm;
                const double v0 = dq - alpha;
                const double beta = 1.0 / (nrm * (nrm + fabs(dq)));  // 2 / v'v
                const double inv_alpha = 1.0 / alpha;
                // Householder vector v = (0,..,0, d_q - alpha, d_{q+1}, ..) as a register; Q <- Q H is applied by
                // apply_pending_q() when Q is read next
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int j = l + 64 * c;
                        q_pend_v[c].at(l) = j > q ? dv[c].at(l) : (j == q ? v0 : 0.0);
                        q_pend_z[c].at(l) = zv[c].at(l);
                    }
                });
                q_pending = true;
                q_pend_col = q;
                q_pend_alpha = alpha;
                q_pend_beta = beta;
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        // R gains the column (d1; alpha)  <=>  S = R^-1 gains the column (-S d1 / alpha; 1 / alpha)
                        const double sv = i < q ? -rv[c].at(l) * inv_alpha : inv_alpha;
                        Rp[blend_idx(i <= q, q * (q + 1) / 2 + i, odump_rp + (l & 31))] = sv;
                        uu[c].at(l) = i == q ? uq : uu[c].at(l);
                        act[c].at(l) = i == q ? ip : act[c].at(l);
                    }
                });
                set_skip(ip, true);
                q++;
                n_add++;
                resolved = true;
                w.stamp(9);  // Householder add
            } else {
                // ---- drop active row ldrop: delete column l of R, restore the triangle with Givens ----
                const int lcol = ldrop;
                set_skip(DQ_BC(act, lcol), false);
                // Deleting column lcol of R: the rotations G of columns (t, t+1), t = lcol..q-2, that restore the
                // triangle are those that sweep row lcol of S into its last entry; S' = (S without row lcol) G'
                // without its last column, Q' = Q G'.  All angles come from one prefix sum of squares.
                RegD srow[RC], pre[RC], gc[RC], gs[RC];
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int k = l + 64 * c;
                        const bool in = k >= lcol && k < q;
                        const double sv = DQ_RP(lcol, in ? k : lcol);
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
                RegD yS[RC], yQ[RC];
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        const double s0 = DQ_RP(i <= lcol ? i : 0, lcol), q0 = Q[(i < r ? i : qdummy) * ldq + lcol];
                        yS[c].at(l) = i <= lcol ? s0 : 0.0;
                        yQ[c].at(l) = q0;
                    }
                });
                for (int t = lcol; t < q - 1; t++) {
                    const double cs = DQ_BC(gc, t + 1), sn = DQ_BC(gs, t + 1);
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int i = l + 64 * c;
                            // S: old column t+1 has rows 0..t+1; row lcol is deleted, rows below it move up
                            const double cS0 = DQ_RP(i <= t + 1 ? i : 0, t + 1);
                            const double cS = i <= t + 1 ? cS0 : 0.0;
                            const double oS = cs * yS[c].at(l) - sn * cS;
                            yS[c].at(l) = sn * yS[c].at(l) + cs * cS;
                            const int inew = i < lcol ? i : i - 1;
                            Rp[blend_idx((i != lcol) & (inew <= t), t * (t + 1) / 2 + inew, odump_rp + (l & 31))] = oS;
                            double *qi = Q + (i < r ? i : qdummy) * ldq;
                            const double cQ = qi[t + 1];
                            qi[t] = cs * yQ[c].at(l) - sn * cQ;
                            yQ[c].at(l) = sn * yQ[c].at(l) + cs * cQ;
                        }
                    });
                    w.sync();
                }
                w.each([&](int l) {
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        const int i = l + 64 * c;
                        Q[(i < r ? i : qdummy) * ldq + q - 1] = yQ[c].at(l);
                    }
                });
                w.sync();
                // compact the per-lane duals and the active list: element k takes element k+1, k >= lcol
                {
                    RegD nu[RC];
                    RegI nact[RC];
#pragma unroll
                    for (int c = 0; c < RC; c++) {
                        w.shift_down(uu, c, RC, nu[c]);
                        w.shift_down(act, c, RC, nact[c]);
                    }
                    w.each([&](int l) {
#pragma unroll
                        for (int c = 0; c < RC; c++) {
                            const int k = l + 64 * c;
                            const bool mv = k >= lcol && k < q - 1;
                            uu[c].at(l) = mv ? nu[c].at(l) : uu[c].at(l);
                            act[c].at(l) = mv ? nact[c].at(l) : act[c].at(l);
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
    // objective before the guarded stores below, so that nothing computed later is live across them
    const double fval = out_cost ? objective() : 0.0;
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

#undef DQ_RP
#undef DQ_BC

}  // namespace dlscqp
