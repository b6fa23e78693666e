// dlscqp.hip -- gfx950 kernel + C ABI (include/dlscqp.h) of the batched DLSC-GC trajectory optimizer.
//
// One agent per wavefront (64 lanes), one wavefront per workgroup; the agent's working set
// (Gram table G_T, packed S = R^-1 of the active rows, control points, step direction, bound table, row flags) is in
// LDS, the stacked LSC rows are read from HBM in structure-of-arrays form (64 consecutive rows per
// load instruction), reductions are DPP/shuffle butterflies.  The algorithm itself is in
// gi_core.hpp; this file supplies the wave primitives, the launch and the host marshalling.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define DQ_FN __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define DQ_WAVE_ALL(c) (__ballot(!(c)) == 0ull)
// a double every lane computes from wave-uniform inputs, moved to a scalar register pair: a division's result is a vector register
// to the compiler whatever went in, and a handful of such quotients hoisted in front of a loop are registers the loop cannot use
namespace {
__device__ __forceinline__ double dq_uniform_f64(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
}  // namespace
#define DQ_UNIFORM_F64(x) dq_uniform_f64(x)
#define DQ_FRESH_INT(i) asm volatile("" : "+v"(i))
#endif
#define DQ_KEEP(v) asm volatile("" : "+v"(v))
#define DQ_MARK(text) asm volatile("; " text)
#if defined(__HIP_DEVICE_COMPILE__)
#define DQ_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double dq_score_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ double dq_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);  // v_rsq_f64: ~2^-26 relative
    return y * fma(-0.5 * x, y * y, 1.5);
}
#define DQ_SCORE_MAX(a, b) dq_score_max((a), (b))
#define DQ_RSQRT(x) dq_rsqrt(x)
#endif
#ifdef DLSCQP_TRACE_LDS
#define DLSCQP_TRACE
#endif
#include "gi_core.hpp"
#include "lsc_core.hpp"
#include "model_rows.hpp"
#include "qp_tables.hpp"
#include "sfc_core.hpp"
#include "sfc_world.hpp"

using namespace dlscqp;

// ---------------------------------------------------------------------------------------------
// wave primitives
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ double uniform_f64(double v) {
    // the butterflies below leave the same value in every lane; tell the compiler so that the
    // control flow that depends on it is scalar
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One wavefront = one agent.  Reg<T> is a VGPR; each() is the lane itself; cross-lane traffic is
// v_readlane (broadcasts), ds_bpermute butterflies (reductions) and ballots.
// FRESH_LANE: every each() block starts from a lane number the compiler cannot see through (see each()).
template <bool FRESH_LANE>
struct WaveGpuT {
    int lane;
    // Helper wavefronts (WPA > 1): the workgroup of an agent has nw wavefronts; wavefront 0 plans, the others price their
    // share of the rows beyond the register cache in every pricing pass.  mail (LDS, 8 doubles): [0] = 1.0 when the planning
    // wavefront is done, [1] = the agent's number of terminal segments T, [1 + v] = best tagged score of wavefront v >= 1.
    int nw = 1;
    double *mail = nullptr;
    __device__ __forceinline__ void pass_begin(int T) const {
        mail[lane & 1] = (lane & 1) ? (double)T : 0.0;  // every lane, two words, the same values (no lane-guarded store)
        __syncthreads();                                 // the control points in LDS are current: helpers may read
    }
    __device__ __forceinline__ double pass_end(double top) const {
        __syncthreads();  // every helper has written its score
        for (int v = 1; v < nw; v++) {
            const double r = uniform_f64(mail[1 + v]);
            top = r > top ? r : top;
        }
        return top;
    }
    template <class T>
    struct Reg {
        T v;
        __device__ __forceinline__ T &at(int) { return v; }
        __device__ __forceinline__ const T &at(int) const { return v; }
    };
#ifdef DLSCQP_STAMPS
    // diagnostic build only (profiles/stamps.py): cycles per phase of plan_agent, never in the product .so
    mutable unsigned long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    mutable unsigned long long last = 0;
    __device__ __forceinline__ void stamp(int p) const {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[p] += t - last;
        last = t;
    }
#elif defined(DLSCQP_ISA_MARKS)
    // diagnostic compile only (tests/tools/isa_phases.py): the phase boundaries as comments in the assembly
    __device__ __forceinline__ void stamp(int p) const {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("; DQ_STAMP %0" ::"i"(p));
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    __device__ __forceinline__ void stamp(int) const {}
#endif
#ifdef DLSCQP_TRACE
    // debug build only: lane 0 logs (tag, 4 doubles) records of this agent's solve into the stamps buffer
    double *tr = nullptr;
    mutable int trn = 0;
    __device__ __forceinline__ void trace(int tag, double a, double b, double c, double d) const {
        if (tr && lane == 0 && trn < 200) {
            double *o = tr + 1 + trn * 5;
            o[0] = tag; o[1] = a; o[2] = b; o[3] = c; o[4] = d;
            tr[0] = trn + 1;
        }
        trn++;
    }
#endif
    // FRESH_LANE: the lane number goes through an empty statement first.  Everything a block derives from it -- LDS addresses (lane x 8
    // + constant), row codes, clamped indices -- is then computed where it is used, one or two instructions each, instead of being
    // hoisted in front of the iteration loop and kept in registers across it: in the S64 instantiation that was ~45 of 250 vector
    // registers (tests/tools/isa_liveness.py), the difference between two and three wavefronts per SIMD.
    unsigned int grid_gen = 0;  // dlscqp_fused_kernel: the generation of the grid barrier this workgroup read before it arrived
    template <class F>
    __device__ __forceinline__ void each(F f) const {
        int l = lane;
        if constexpr (FRESH_LANE) asm volatile("" : "+v"(l));
        f(l);
    }
    // wave-uniform work on the working set: every lane reads, computes and writes the same (lock step: all lanes read before any writes)
    template <class F>
    __device__ __forceinline__ void once(F f) const { f(); }
    __device__ __forceinline__ void sync() const {
#ifdef DLSCQP_SAFE_SYNC
        __syncthreads();
#else
        // one wavefront per agent: LDS instructions of a wave execute in order, so only the
        // compiler has to be kept from moving LDS accesses across this point
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    }
    // broadcasts: the source lane is wave-uniform, so v_readlane_b32 (SGPR result, no LDS)
    __device__ __forceinline__ double bcast(const Reg<double> &r, int l) const {
        return readlane_f64(r.v, __builtin_amdgcn_readfirstlane(l));
    }
    __device__ __forceinline__ float bcast(const Reg<float> &r, int l) const {
        return readlane_f32(r.v, __builtin_amdgcn_readfirstlane(l));
    }
    __device__ __forceinline__ uint32_t bcast(const Reg<uint32_t> &r, int l) const {
        return (uint32_t)__builtin_amdgcn_readlane((int)r.v, __builtin_amdgcn_readfirstlane(l));
    }
    __device__ __forceinline__ int bcast(const Reg<int> &r, int l) const {
        return __builtin_amdgcn_readlane(r.v, __builtin_amdgcn_readfirstlane(l));
    }
    // Wave reductions on the DPP data path (no LDS crossbar): inclusive combine within each row of 16
    // lanes (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals up; lane 63 ends up
    // with the result, which v_readlane makes uniform.
    template <int CTRL, int ROW_MASK>
    static __device__ __forceinline__ double dpp_f64(double identity, double v) {
        int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ double max(const Reg<double> &r) const {
        const double id = -1.7976931348623157e308;
        double s = r.v;
        s = fmax(s, dpp_f64<0x111, 0xf>(id, s));
        s = fmax(s, dpp_f64<0x112, 0xf>(id, s));
        s = fmax(s, dpp_f64<0x114, 0xf>(id, s));
        s = fmax(s, dpp_f64<0x118, 0xf>(id, s));
        s = fmax(s, dpp_f64<0x142, 0xa>(id, s));
        s = fmax(s, dpp_f64<0x143, 0xc>(id, s));
        return readlane_f64(s, 63);
    }
    // Maximum of values that are all >= +0 (tagged pivot scores): their bit patterns order like unsigned integers, so
    // two 32-bit reductions -- the high words, then the low words of the lanes that hold the largest high word -- with
    // the DPP step folded into v_max_u32 replace six steps of two DPP moves and a v_max_f64 each.
    static __device__ __forceinline__ uint32_t max_u32_wave(uint32_t v) {
        v = ::max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
        v = ::max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
        v = ::max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
        v = ::max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
        v = ::max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
        v = ::max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
        return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    }
    __device__ __forceinline__ double max_pos(const Reg<double> &r) const {
        const uint32_t hi = (uint32_t)__double2hiint(r.v), lo = (uint32_t)__double2loint(r.v);
        const uint32_t hmax = max_u32_wave(hi);
        const uint32_t lmax = max_u32_wave(hi == hmax ? lo : 0u);
        return __hiloint2double((int)hmax, (int)lmax);
    }
    __device__ __forceinline__ double min(const Reg<double> &r) const {
        const double id = 1.7976931348623157e308;
        double s = r.v;
        s = fmin(s, dpp_f64<0x111, 0xf>(id, s));
        s = fmin(s, dpp_f64<0x112, 0xf>(id, s));
        s = fmin(s, dpp_f64<0x114, 0xf>(id, s));
        s = fmin(s, dpp_f64<0x118, 0xf>(id, s));
        s = fmin(s, dpp_f64<0x142, 0xa>(id, s));
        s = fmin(s, dpp_f64<0x143, 0xc>(id, s));
        return readlane_f64(s, 63);
    }
    __device__ __forceinline__ void sum2(const Reg<double> &a, const Reg<double> &b, double &sa, double &sb) const {
        double x = a.v, y = b.v;  // two independent chains, interleaved
        x += dpp_f64<0x111, 0xf>(0.0, x); y += dpp_f64<0x111, 0xf>(0.0, y);
        x += dpp_f64<0x112, 0xf>(0.0, x); y += dpp_f64<0x112, 0xf>(0.0, y);
        x += dpp_f64<0x114, 0xf>(0.0, x); y += dpp_f64<0x114, 0xf>(0.0, y);
        x += dpp_f64<0x118, 0xf>(0.0, x); y += dpp_f64<0x118, 0xf>(0.0, y);
        x += dpp_f64<0x142, 0xa>(0.0, x); y += dpp_f64<0x142, 0xa>(0.0, y);
        x += dpp_f64<0x143, 0xc>(0.0, x); y += dpp_f64<0x143, 0xc>(0.0, y);
        sa = readlane_f64(x, 63);
        sb = readlane_f64(y, 63);
    }
    // inclusive prefix sum (the same DPP sequence is a Hillis-Steele scan), carried across registers
    __device__ __forceinline__ void prefix_sum(Reg<double> *arr, int nchunks) const {
        double carry = 0.0;
        for (int c = 0; c < nchunks; c++) {
            double x = arr[c].v;
            x += dpp_f64<0x111, 0xf>(0.0, x);
            x += dpp_f64<0x112, 0xf>(0.0, x);
            x += dpp_f64<0x114, 0xf>(0.0, x);
            x += dpp_f64<0x118, 0xf>(0.0, x);
            x += dpp_f64<0x142, 0xa>(0.0, x);
            x += dpp_f64<0x143, 0xc>(0.0, x);
            x += carry;
            arr[c].v = x;
            carry = readlane_f64(x, 63);
        }
    }
    template <class T>
    __device__ __forceinline__ void shift_up(const Reg<T> *arr, int c, int nchunks, Reg<T> &out) const {
        (void)nchunks;
        const T prv = __shfl_up(arr[c].v, 1, 64);
        const T tail = c > 0 ? __shfl(arr[c - 1].v, 63, 64) : T(0);
        out.v = lane == 0 ? tail : prv;
    }
    __device__ __forceinline__ double sum(const Reg<double> &a) const {
        double x = a.v;
        x += dpp_f64<0x111, 0xf>(0.0, x);
        x += dpp_f64<0x112, 0xf>(0.0, x);
        x += dpp_f64<0x114, 0xf>(0.0, x);
        x += dpp_f64<0x118, 0xf>(0.0, x);
        x += dpp_f64<0x142, 0xa>(0.0, x);
        x += dpp_f64<0x143, 0xc>(0.0, x);
        return readlane_f64(x, 63);
    }
    // a value that every lane computed identically (e.g. from LDS reads at a uniform address): into SGPRs, so that the
    // control flow that depends on it is scalar
    __device__ __forceinline__ double uni(double v) const { return uniform_f64(v); }
    __device__ __forceinline__ uint64_t ballot(const Reg<int> &r) const { return __ballot(r.v != 0); }
    __device__ __forceinline__ bool any(const Reg<int> &r) const { return __ballot(r.v != 0) != 0ull; }
    __device__ __forceinline__ int first_lane(const Reg<int> &r) const {
        unsigned long long m = __ballot(r.v != 0);
        return m ? __builtin_ctzll(m) : -1;
    }
    template <class T>
    __device__ __forceinline__ void shift_down(const Reg<T> *arr, int c, int nchunks, Reg<T> &out) const {
        T nxt = __shfl_down(arr[c].v, 1, 64);
        if (c + 1 < nchunks) {  // uniform
            const T head = __shfl(arr[c + 1].v, 0, 64);
            nxt = lane == 63 ? head : nxt;
        }
        out.v = nxt;
    }
};
typedef WaveGpuT<false> WaveGpu;

// ---- between two launches (include/dlscqp.h: dlscqp_advance): one wavefront per agent ----
struct AdvanceArgs {
    dlscqp_advance a;
    int M, n1;
    float s1, s2;  // n / dt and (n - 1) / dt in float32 (Trajectory::getStateAt differences)
};

// one wavefront per agent; E = M (n+1) 3 floats of a plan, lane l owns elements l, l + 64, ...
// (shared by dlscqp_advance_kernel and the plan kernel's epilogue: dlscqp_device_batch.advance_after)
// stage: E floats of LDS that are this wavefront's (round 6).  The new plan goes to memory AND to `stage`; everything that is derived
// from other lanes' elements of it -- the initial trajectory of the next step (the plan shifted by a segment), the boxes around that,
// the state at dt -- is read from there.  Before, each of those was a store, a wait for it to land, and a dependent load from L2:
// eight load-wait-store pairs one after the other for the initial trajectory alone (the compiler cannot batch loads of `plan` across
// stores to `initial_traj`: for all it knows they alias), ~6 us per agent behind every solve of a closed loop.
__device__ __forceinline__ void advance_agent(const AdvanceArgs &g, int ag, int l, float *stage) {
    const dlscqp_advance &a = g.a;
    const int M = g.M, n1 = g.n1, seg = n1 * 3, E = M * seg;
    float *plan = a.plans + (size_t)ag * E;
    const float *ctrl = a.ctrl + (size_t)ag * E;
    const bool ok = a.status[ag] == DLSCQP_OK;
    // element e of a plan shifted by one segment: segment m+1, the last segment repeats its end point
    auto shifted = [&](int e) {
        const int m = e / seg, k = e % 3;
        return m < M - 1 ? e + seg : (M - 1) * seg + (n1 - 1) * 3 + k;
    };
    constexpr int U = 8;  // E <= 512 (M <= 28 at n = 5)
    float v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int e = l + 64 * u;
        v[u] = e < E ? (ok ? ctrl[e] : plan[shifted(e)]) : 0.0f;  // every read of the old plan happens before ...
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int e = l + 64 * u;
        if (e < E) {
            plan[e] = v[u];  // ... the first write (one wavefront: memory operations in program order)
            stage[e] = v[u];
        }
    }
    // (one wavefront: its LDS operations are performed in order; the compiler is told not to move them across this point)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (a.out_host_pack) {  // uniform: the plan and the alerts for the host-side waypoint planner (pinned host memory)
        float *hp = a.out_host_pack + (size_t)a.n_agents * 6;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = l + 64 * u;
            if (e < E) hp[(size_t)ag * E + e] = v[u];
        }
        hp += (size_t)a.n_agents * E;
        for (int j = l; j < a.n_alert_obs; j += 64)
            hp[(size_t)ag * a.n_alert_obs + j] = a.alert_slack_cost[(size_t)ag * a.alert_slack_stride + j] > a.alert_threshold ? 1.0f : 0.0f;
    }
    if (a.initial_traj) {
        float *init = a.initial_traj + (size_t)ag * E;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = l + 64 * u;
            if (e < E) init[e] = stage[shifted(e)];
        }
        if (a.out_boxes) {  // uniform
            for (int j = l; j < M * 3; j += 64) {  // face (segment m, axis k): minimum and maximum over the segment's control points
                const int m = j / 3, k = j - 3 * m;
                float mn = stage[shifted(m * seg + k)], mx = mn;  // (the initial trajectory's element, from the new plan)
                for (int i = 1; i < n1; i++) {
                    const float v = stage[shifted(m * seg + 3 * i + k)];
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
                const float lo = mn - a.box_margin, hi = mx + a.box_margin;
                float *box = a.out_boxes + ((size_t)ag * M + m) * 6;
                box[k] = lo > a.box_min[k] ? lo : a.box_min[k];
                box[3 + k] = hi < a.box_max[k] ? hi : a.box_max[k];
            }
        }
    }
    if (l < 3) {
        const int k = l;
        const bool gok = a.goal_status[ag] != DLSCQP_INFEASIBLE;
        const float cg = gok ? a.goal[(size_t)ag * 3 + k] : a.current_goal[(size_t)ag * 3 + k];
        a.current_goal[(size_t)ag * 3 + k] = cg;
        // state at dt: first three control points of segment 1 (M >= 2, dlscqp_create)
        const int b = seg;
        const float c0 = stage[b + k], c1 = stage[b + 3 + k], c2 = stage[b + 6 + k];
        const float v0 = (c1 - c0) * g.s1, v1 = (c2 - c1) * g.s1;
        const float p = c0, vel = v0, acc = (v1 - v0) * g.s2;
        dlscqp_agent *rec = a.agents + ag;
        rec->position[k] = p;
        rec->velocity[k] = vel;
        rec->acceleration[k] = acc;
        rec->current_goal_point[k] = cg;
        if (a.out_host_pack) {
            a.out_host_pack[(size_t)ag * 3 + k] = p;
            a.out_host_pack[(size_t)a.n_agents * 3 + (size_t)ag * 3 + k] = cg;
        }
        if (a.waypoints) {
            rec->next_waypoint[k] = a.waypoints[(size_t)ag * 3 + k];
        } else if (a.desired_goal) {
            const float *dg = a.desired_goal + (size_t)ag * 3;
            // the other two axes of the new position: recomputed by this lane (three lanes, three axes each)
            float d3[3], p3[3];
            for (int j = 0; j < 3; j++) {
                p3[j] = stage[b + j];
                d3[j] = dg[j] - p3[j];
            }
            const double dist = dlscqp::f32_norm3(d3[0], d3[1], d3[2]);
            const float stepf = (float)a.waypoint_step, distf = (float)(dist > 1e-9 ? dist : 1e-9);
            float wpn = dist <= a.waypoint_step ? dg[k] : p3[k] + d3[k] / distf * stepf;
            // an agent whose goal LP found no point towards its waypoint aims at its sub-goal for one step (the synthetic swarms'
            // rule, workloads.generate; in the reference the grid planner hands out waypoints that can be reached)
            if (a.waypoint_hold_on_lp_failure && !gok) wpn = cg;
            rec->next_waypoint[k] = wpn;
        }
    }
}

struct KernelArgs {
    int n_agents, max_slack_obs, do_goal, do_traj;
    int slot0, n_slots_launch;  // launch slots [slot0, slot0 + n_slots_launch)
    const dlscqp_agent *agents;
    const int32_t *row_offsets;
    const int32_t *row_counts;       // optional: rows of agent a = row_counts[a] from row_offsets[a] on
    RowsView rows;
    const float *sfc;
    float *out_goal;
    int32_t *out_goal_status;
    float *out_ctrl;
    double *out_x;
    double *out_cost;
    double *out_slack_cost;
    int32_t *out_status;
    int32_t *out_iters;
    const int32_t *order;            // optional launch permutation
    const int32_t *hint[3];          // launch order found by the wavefront itself (hinted_agent): out_iters of up to three previous steps
    int n_blocks;                    // blocks of 64 consecutive agents, (n_agents + 63) / 64
    unsigned int blocks_magic;       // ceil(2^32 / n_blocks) where slot / n_blocks = umulhi(slot, magic) for every slot of the launch, else 0
    int row_pitch;                   // > 0: agent a's rows start at a * row_pitch (row_offsets is not read)
    int ctrl_stride, goal_stride, status_stride;  // words from agent to agent in out_ctrl / out_goal / out_status (dense: M (n+1) 3, 3, 1)
    unsigned long long *out_stamps;  // [n_agents][12], diagnostic build only
    unsigned char *work;             // HBM working sets [n_agents][work_stride] of the WS_HBM kernel, else null
    size_t work_stride;
    double *spill;                   // spill records [launch slots][spill_stride doubles] (MemLayout::spill_doubles), or null: every column
    size_t spill_stride;             // record of the active rows fits the working set
    const unsigned long long *mask_in;  // launch order from class masks (masked_agent): [n_blocks][8], written by the previous step's launch
    unsigned long long *mask_out;       // ... this launch's own: every agent sets its bit in the mask of its class
    unsigned long long *mask_zero;      // ... and a third table that this launch clears (for a later launch to write)
    unsigned int *grid;                 // dlscqp_fused_kernel: [0] arrivals, [1] generation of the grid-wide barrier in front of the epilogue, or null
    int grid_n;                         // ... workgroups of the launch
    int fresh_inputs;                   // ... the agent's row counts were written by THIS kernel: no load through the scalar cache
    double *out_conflict;               // dlscqp_device_batch.out_conflict: [n_agents][conflict_stride] doubles, or null
    int conflict_stride;
    int adv_on;                         // dlscqp_device_batch.advance_after: the planning wavefront also does the agent's bookkeeping between
    AdvanceArgs adv;                    // two launches (advance_agent), right behind its final stores
};

// A 32-bit word (two consecutive ones) at a wave-uniform index of a table that no store of the running kernel touches, through
// the scalar cache: s_load_dword, its result in scalar registers.  (The tables are written by EARLIER launches at most: the scalar
// cache is invalidated when a kernel starts.  A plain statement, neither volatile nor with a memory clobber: behind either the
// compiler no longer knows that the agent's record is unchanged and reads it with vector loads instead of scalar ones.)
__device__ __forceinline__ int scalar_load_i32(const int32_t *p, int idx) {
    int v;
    asm("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "s"(idx * 4));
    return v;
}
__device__ __forceinline__ void scalar_load_i32x2(const int32_t *p, int idx, int &v0, int &v1) {
    unsigned long long v;
    asm("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "s"(idx * 4));
    v0 = (int)(unsigned int)v;
    v1 = (int)(unsigned int)(v >> 32);
}

// dlscqp_device_batch.hint_iters: the launch order without a sort between two launches.  The agents in blocks of 64 consecutive
// ids; slot s belongs to block s % n_blocks and plans the agent of rank s / n_blocks among the block's agents, most expensive
// first.  "Expensive" is the key of dlscqp_launch_order_history_device -- max_k (3/4)^k iterations_k over the previous steps --
// in eight classes (0, 1, 2-3, 4-7, ..., 64 and more iterations); ties by id.  Lane l holds the key of agent 64 block + l: eight
// ballots give the class sizes, the class the rank falls into and, inside it, the lane (v_mbcnt).  Every wavefront of a
// workgroup computes the same answer from the same data.  Returns -1 for a slot beyond the block's agents (a last block of
// fewer than 64).
__device__ __forceinline__ int hinted_agent(const KernelArgs &a, int slot, int lane) {
    const int nb = a.n_blocks;
    const int rank = a.blocks_magic ? (int)__umulhi((unsigned int)slot, a.blocks_magic) : slot / nb;
    const int blk = slot - rank * nb;
    const int idx = blk * 64 + lane;
    const bool valid = idx < a.n_agents;
    const size_t ic = (size_t)(valid ? idx : a.n_agents - 1) * 4;
    int it = a.hint[0][ic];
    if (a.hint[1]) { const int v = (3 * a.hint[1][ic]) / 4; it = v > it ? v : it; }   // uniform branches
    if (a.hint[2]) { const int v = (9 * a.hint[2][ic]) / 16; it = v > it ? v : it; }
    int cls = it <= 0 ? 0 : 32 - __clz(it);  // 0 | 1 | 2-3 | 4-7 | ...
    cls = cls > 7 ? 7 : cls;
    int left = rank, pick = -1;  // wave-uniform from here on
#pragma unroll
    for (int c = 7; c >= 0; c--) {
        const unsigned long long m = __ballot(valid && cls == c);
        const int cnt = __popcll(m);
        if (pick < 0 && left < cnt) {
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            const unsigned long long hit = __ballot(((m >> lane) & 1ull) != 0ull && below == left);  // exactly one lane
            pick = __builtin_ctzll(hit);
        }
        left -= pick < 0 ? cnt : 0;
    }
    return pick < 0 ? -1 : blk * 64 + pick;
}

// dlscqp_device_batch.hint_mask_in: the same order -- blocks of 64 agents, slot s plans the (s / n_blocks)-th most expensive agent
// of block s % n_blocks -- from a table the PREVIOUS step's launch left behind: per block eight 64-bit masks, one per class of
// iteration counts, in which every agent set its own bit when it finished (one atomic OR without a return value, nothing waits
// for it).  The wavefront reads its block's 64 bytes with scalar loads and walks the classes with population counts: no vector
// load of 64 keys in front of the agent's record (that dependent trip cost every agent ~1.3 us), ~40 scalar instructions.
// The masks must partition the block's agents; if they do not (the first step: all zero) the block is planned in id order.
// cls_in: the class the agent was scheduled with (its own history: the class it reports decays by one per step).
__device__ __forceinline__ int masked_agent(const unsigned long long *mask_in, int nb, unsigned int blocks_magic, int n_agents, int slot, int lane, int &cls_in) {
    const int rank = blocks_magic ? (int)__umulhi((unsigned int)slot, blocks_magic) : slot / nb;
    const int blk = slot - rank * nb;
    const unsigned long long *m = mask_in + (size_t)blk * 8;
    unsigned long long mk[8];
#pragma unroll
    for (int c = 0; c < 8; c++) mk[c] = __builtin_nontemporal_load(m + c);
    const int nvalid = n_agents - 64 * blk >= 64 ? 64 : n_agents - 64 * blk;
    const unsigned long long vmask = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
    unsigned long long all = 0ull;
    int tot = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        mk[c] = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(mk[c] >> 32)) << 32) |
                (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)mk[c]);  // (every lane read the same words)
        all |= mk[c];
        tot += __popcll(mk[c]);
    }
    cls_in = 0;
    if (all != vmask || tot != nvalid) return rank < nvalid ? 64 * blk + rank : -1;  // not a partition of the block: id order
    int left = rank, pick = -1;
#pragma unroll
    for (int c = 7; c >= 0; c--) {
        const int cnt = __popcll(mk[c]);
        if (pick < 0 && left < cnt) {
            const unsigned long long mm = mk[c];
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mm, 0u));
            const unsigned long long hit = __ballot(((mm >> lane) & 1ull) != 0ull && below == left);  // exactly one lane
            pick = __builtin_ctzll(hit);
            cls_in = c;
        }
        left -= pick < 0 ? cnt : 0;
    }
    return pick < 0 ? -1 : blk * 64 + pick;
}
// the class of an iteration count: 0 | 1 | 2-3 | 4-7 | ... | 64 and more
__device__ __forceinline__ int iteration_class(int it) {
    const int c = it <= 0 ? 0 : 32 - __clz(it);
    return c > 7 ? 7 : c;
}

// Two wavefronts per SIMD (256 VGPRs each) where eight agents fit the LDS of a CU; where the working set allows two or three
// agents per CU anyway (the static shapes with reduced dimension above 64, and the 2-D M = 10 shape with its 54 KB) one
// wavefront per SIMD is all that can run: those instantiations get the whole register file, and what spilled to scratch
// memory lives in AGPRs.  (Not the runtime-shape two-register instantiation and its HBM variants: they gained 3 %, and one
// developer-only compiler configuration of tests/flag_matrix.py, -sgpr-regalloc=fast, produced a wrong forest batch with it.)
// The instantiations that keep two wavefronts per SIMD exist a second time, WIDE, for launches too small to put a second
// wavefront on a SIMD at all (at most one agent per SIMD of the chip: every mission of the reference).
#ifndef DQ_WPA_OCC
#define DQ_WPA_OCC 1  // wavefronts per SIMD the builds with helper wavefronts are compiled for
#endif
#ifndef DQ_M10_2D_ONE  // the 2-D, M = 10 shape at one wavefront per SIMD (0: two -- an experiment of round 4, profiles/)
#define DQ_M10_2D_ONE 1
#endif
#ifndef DQ_PLAN_OCC  // wavefronts per SIMD the standard builds are compiled for (3: an experiment of round 5)
#define DQ_PLAN_OCC 2
#endif
#ifndef DQ_S64_OCC  // ... and the instantiation whose rows are all in registers (S64; gi_core.hpp: LEAN) -- round 6: three, in 168 registers without a
#define DQ_S64_OCC 3  // byte of scratch memory, since nothing lane-derived or converted is kept across the iteration loop any more
#endif
#ifndef DQ_KERNEL_ATTR
#define DQ_KERNEL_ATTR __launch_bounds__(64 * WPA, WPA > 1 ? DQ_WPA_OCC : ((WIDE || (DIMS > 0 && (RC == 2 || (DIMS == 2 && FS == 28 && DQ_M10_2D_ONE)))) ? 1 : \
                                                             ((DIMS > 0 && KU == 2 && SU == 0 && SLK == 0) ? DQ_S64_OCC : DQ_PLAN_OCC)))
#endif
// WS_HBM: the agent's working set (MemLayout) lives in a per-slot HBM buffer instead of LDS -- for batches whose working
// set exceeds the 160 KiB of a CU.  Since the solver keeps no orthogonal factor (gi_core.hpp) that is only a batch with
// a very long flag array of rows beyond the register cache (> 60 000 rows per agent); the reference's largest shape
// (3-D, M = 10, four slack obstacles: 124 active rows at most) needs 95 KB.  Same code, same layout; what was an LDS
// round trip becomes an L1/L2 round trip (the buffer of a running agent stays cache-resident) -- a capability path.
// One wavefront owns a buffer and a wavefront's memory operations are performed in program order, so no fences are
// needed between its lanes.
enum { WS_LDS = 0, WS_HBM = 1 };
// WPA (wavefronts per agent) > 1: a workgroup of WPA wavefronts per agent for launches that leave most of the chip empty
// (few agents) or whose agents have long row lists (a crowd): wavefront 0 plans as always; wavefronts 1 .. WPA-1 are HELPERS
// that price their share of the rows beyond the planning wavefront's register cache, once per pricing pass, between two
// workgroup barriers (WaveGpu::pass_begin / pass_end).  Same arithmetic per row, same maximum: the answers do not depend on
// WPA (tests/test_gpu_wide_builds.py).
// One launch slot's agent: goal LP + trajectory QP + the epilogues (the body of dlscqp_plan_kernel; also the last stage of
// dlscqp_fused_kernel).  `w`: the wavefront (lane set); smem: the working set (LDS, or the slot's HBM buffer).
template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK, int WS, bool WIDE, int WPA, class Wave>
__device__ __forceinline__ void plan_slot(const KParams &kp, const MemLayout &lay, const KernelArgs &a, unsigned char *smem, Wave &w,
                                          const int slot, const int ag, const int cls_in) {
    // the agent's rows: [rb, rb + nrows_ag).  Scalar loads, both offsets in one (the compiler reads them with vector loads, one
    // dependent trip to L2 each, although the index is uniform: it cannot tell that no store of this kernel touches them)
    int rb, nrows_ag;
    if (a.fresh_inputs) {  // uniform (dlscqp_fused_kernel: this kernel's own stores; the scalar cache may hold an older line)
        rb = a.row_pitch > 0 ? ag * a.row_pitch : __builtin_amdgcn_readfirstlane(a.row_offsets[ag]);
        const int re = a.row_pitch > 0 ? rb + a.row_pitch : __builtin_amdgcn_readfirstlane(a.row_offsets[ag + 1]);
        nrows_ag = a.row_counts ? __builtin_amdgcn_readfirstlane(__builtin_nontemporal_load(a.row_counts + ag)) : re - rb;
    } else if (a.row_pitch > 0) {
        rb = ag * a.row_pitch;
        nrows_ag = a.row_counts ? scalar_load_i32(a.row_counts, ag) : a.row_pitch;
    } else {
        int re;
        scalar_load_i32x2(a.row_offsets, ag, rb, re);
        nrows_ag = re - rb;
        if (a.row_counts) nrows_ag = scalar_load_i32(a.row_counts, ag);
    }
    if constexpr (WPA > 1) {
        static_assert(WS == WS_LDS, "the helpers' mailbox is in LDS");
        w.nw = WPA;
        w.mail = reinterpret_cast<double *>(smem) + lay.oMail;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // wave-uniform, and known to be
        if (wave > 0) {
            // ---- helper wavefront: price a share of the cold rows whenever the planning wavefront opens a pass -------------
            const int K = nrows_ag;
            const int Kc = K < 64 * KU ? K : 64 * KU;
            const dlscqp_agent *agp = a.agents + ag;
            ColdRows c;
            c.nx = a.rows.nx + rb; c.ny = a.rows.ny + rb; c.nz = a.rows.nz + rb;
            c.px = a.rows.px + rb; c.py = a.rows.py + rb; c.pz = a.rows.pz + rb;
            c.d = a.rows.d + rb; c.key = a.rows.key + rb;
            c.flag = smem + lay.oFlagRows;
            c.x = reinterpret_cast<const double *>(smem) + lay.oX;
            c.sigma = kp.sigma;
            c.K = K; c.Kc = Kc; c.M = kp.M; c.P = kp.M * kp.n1; c.dim = kp.dim; c.nxc = kp.dim * c.P;
            int nso = agp->n_slack_obs;
            nso = nso > a.max_slack_obs ? a.max_slack_obs : (nso < 0 ? 0 : nso);
            c.nso = SLK == 0 ? 0 : nso;
            c.phi = kp.phi;
            c.unit0 = DIMS > 0 ? c.nxc : 0;
            // this helper's share of the rows beyond the register cache: its part of the rows the planning wavefront has decoded into
            // LDS (round 5: the helpers price those too -- a crowd agent's pass is mostly those), its part of the rows in HBM
            const int Kl = K < Kc + lay.cold_cap ? K : Kc + lay.cold_cap;
            c.Kc = Kl;  // (the skip flags are one bit per row from Kl on)
            int jb, je;
            cold_rows_share(Kc, K, wave, WPA, jb, je);
            const int ljb = jb < Kl ? jb : Kl, lje = je < Kl ? je : Kl, hjb = jb > Kl ? jb : Kl, hje = je > Kl ? je : Kl;
            LdsRows lr;
            {
                const double *base = reinterpret_cast<const double *>(smem);
                lr.rhs = base + lay.oCold; lr.iw = lr.rhs + lay.cold_cap;
                lr.nx = reinterpret_cast<const float *>(lr.iw + lay.cold_cap); lr.ny = lr.nx + lay.cold_cap; lr.nz = lr.ny + lay.cold_cap;
                lr.key = reinterpret_cast<const uint32_t *>(lr.nz + lay.cold_cap);
                lr.x = c.x; lr.Kc = Kc; lr.P = c.P; lr.M = c.M; lr.dim = c.dim; lr.nxc = c.nxc; lr.unit0 = c.unit0;
            }
            for (;;) {
                __syncthreads();  // pass_begin (or the end of the solve)
                if (uniform_f64(w.mail[0]) != 0.0) break;  // every lane reads the same word
                c.tNN = kp.tabNN + (size_t)((int)uniform_f64(w.mail[1]) - 1) * c.P;
                typename Wave::template Reg<double> best;
                double least = 0.0;  // (the helpers price all of their shares in every pass)
                best.v = lds_rows_lane_best<SLK != 0>(lr, w.lane, ljb, lje, 0.0, least);
                best.v = cold_rows_lane_best<4>(c, w.lane, hjb, hje, best.v, least);
                const double top = w.max_pos(best);
                w.mail[1 + wave] = top;  // every lane, the same word, the same value
                __syncthreads();  // pass_end
            }
            return;
        }
    }
#ifdef DLSCQP_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w.last)::"memory");
#endif
#ifdef DLSCQP_TRACE
#ifdef DLSCQP_TRACE_LDS
    w.tr = reinterpret_cast<double *>(smem + lay.bytes);  // into LDS, dumped by DLSCQP_PM_LDS
    if (threadIdx.x == 0) w.tr[0] = 0;
#else
    if (a.out_stamps) w.tr = reinterpret_cast<double *>(a.out_stamps) + (size_t)ag * 1001;
#endif
#endif
#ifdef DLSCQP_TIMELINE
    // diagnostic build only (profiles/timeline.py): when and where (CU, SIMD, wave slot) this agent's wavefront ran
    unsigned long long tl0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0)::"memory");
#endif
    const int M = kp.M, n1 = kp.n1;
    const size_t xs = (size_t)kp.dim * M * n1 + (size_t)M * a.max_slack_obs;
    int iters_done = 0;
    plan_agent<RC, DIMS, FS, MS, KU, SU, SLK, Wave>(w, kp, lay, smem, a.spill ? a.spill + (size_t)slot * a.spill_stride : nullptr,
               a.agents + ag, a.rows, rb, nrows_ag,
               a.sfc ? a.sfc + (size_t)ag * M * 6 : nullptr, a.max_slack_obs, a.do_goal, a.do_traj,
               a.out_goal ? a.out_goal + (size_t)ag * a.goal_stride : nullptr,
               a.out_goal_status ? a.out_goal_status + ag : nullptr,
               a.out_ctrl ? a.out_ctrl + (size_t)ag * a.ctrl_stride : nullptr,
               a.out_x ? a.out_x + ag * xs : nullptr, a.out_cost ? a.out_cost + ag : nullptr,
               a.out_slack_cost ? a.out_slack_cost + (size_t)ag * a.max_slack_obs : nullptr,
               a.out_status ? a.out_status + (size_t)ag * a.status_stride : nullptr,
               a.out_iters ? a.out_iters + (size_t)ag * 4 : nullptr, &iters_done,
               a.out_conflict ? a.out_conflict + (size_t)ag * a.conflict_stride : nullptr);
    if (a.mask_out) {
        // the agent's bit into the mask of its class, for the next step's launch order (masked_agent): the class of this step's
        // iteration count, or one below the class it was scheduled with if that is higher (an agent that was expensive a step or
        // two ago is still likely to be); no return value, nothing waits
        const int c_now = iteration_class(iters_done);
        const int cls = c_now > cls_in - 1 ? c_now : cls_in - 1;
        if (threadIdx.x == 0) __hip_atomic_fetch_or(a.mask_out + (size_t)(ag >> 6) * 8 + cls, 1ull << (ag & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (WPA > 1) {  // release the helpers: the solve is over
        w.mail[0] = 1.0;
        __syncthreads();
    }
    if (a.adv_on) {  // uniform
        // what dlscqp_advance_device would do for this agent in a launch of its own: it reads the verdicts and the plan this wavefront
        // has just stored (a wavefront's memory operations are performed in program order) and touches this agent's entries only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.grid) {
            // dlscqp_fused_kernel: the epilogue overwrites this agent's plan, sub-goal and state, which the OTHER agents' half-space and
            // neighbour stages of this very launch read -- it waits until every workgroup has arrived behind those stages (they did
            // so before their solves: the wait is the slowest agent's, which the step waits for anyway)
            while (__hip_atomic_load(a.grid + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == w.grid_gen) __builtin_amdgcn_s_sleep(8);
        }
        // (its staging buffer: the column records' place in the working set, free now -- one record is longer than a plan)
        advance_agent(a.adv, ag, w.lane, reinterpret_cast<float *>(reinterpret_cast<double *>(smem) + lay.oC));
    }
#ifdef DLSCQP_STAMPS
    if (a.out_stamps && threadIdx.x == 0)
        for (int p = 0; p < 12; p++) a.out_stamps[(size_t)ag * 12 + p] = w.acc[p];
#endif
#ifdef DLSCQP_TIMELINE
    if (a.out_stamps) {
        unsigned long long tl1;
        unsigned int hw, xcc;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl1)::"memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (threadIdx.x == 0) {
            a.out_stamps[(size_t)ag * 12 + 0] = tl0;
            a.out_stamps[(size_t)ag * 12 + 1] = tl1;
            a.out_stamps[(size_t)ag * 12 + 2] = hw;
            a.out_stamps[(size_t)ag * 12 + 3] = xcc;
            a.out_stamps[(size_t)ag * 12 + 4] = (unsigned long long)slot;
        }
    }
#endif
#ifdef DLSCQP_PM_LDS
    // debug build only: image of the agent's LDS working set when the solve returns (layout header + doubles)
    if (a.out_stamps) {
        unsigned long long *o = a.out_stamps + (size_t)ag * 20992;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(smem);
        if (threadIdx.x == 0) {
            o[0] = lay.bytes; o[1] = lay.oC; o[2] = lay.cs; o[3] = lay.oDx; o[4] = lay.oX; o[5] = lay.oBlo; o[6] = lay.oBhi;
            o[7] = lay.oFlag; o[8] = lay.qz; o[9] = lay.r_max;
        }
        int nd = lay.bytes / 8;
#ifdef DLSCQP_TRACE_LDS
        nd += 1001;
#endif
        for (int i = threadIdx.x; i < nd; i += 64) o[16 + i] = src[i];
    }
#endif
}

template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK, int WS = WS_LDS, bool WIDE = false, int WPA = 1>
__global__ DQ_KERNEL_ATTR void dlscqp_plan_kernel(KParams kp, MemLayout lay, KernelArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    if ((int)blockIdx.x >= a.n_slots_launch) return;
    const int slot = a.slot0 + (int)blockIdx.x;
    unsigned char *smem;
    if constexpr (WS == WS_HBM) smem = a.work + (size_t)slot * a.work_stride;
    else smem = lds;
    // the instantiation whose rows are all in registers (S64) is compiled for three wavefronts per SIMD: gi_core.hpp "LEAN"
    constexpr bool LEAN_K = DIMS > 0 && KU == 2 && SU == 0 && SLK == 0 && !WIDE && WPA == 1 && WS == WS_LDS;
#ifndef DQ_FRESH_LANE_ALL  // 1: every instantiation starts its each() blocks from a fresh lane number, not only the one that needs it for its third wavefront
#define DQ_FRESH_LANE_ALL 1  // per SIMD.  Measured at the end of round 6: the crowd's kernel 406 -> 376 registers and 100 -> 20 B of scratch, the two-per-SIMD
#endif                       // builds spill 29-90 registers instead of 56-124; crowd step 0.2983 -> 0.2938 ms, empty60's LP + QP launch 0.0427 -> 0.0405
    typedef WaveGpuT<LEAN_K || DQ_FRESH_LANE_ALL != 0> Wave;
    Wave w;
    w.lane = (int)threadIdx.x & 63;
    int cls_in = 0;
    // (wave-uniform by construction -- ballots and scalar loads; said aloud: it indexes scalar loads below)
    const int ag = __builtin_amdgcn_readfirstlane(a.order ? a.order[slot] : (a.mask_in ? masked_agent(a.mask_in, a.n_blocks, a.blocks_magic, a.n_agents, slot, w.lane, cls_in)
                                                                                      : (a.hint[0] ? hinted_agent(a, slot, w.lane) : slot)));
    if (a.mask_zero && slot < a.n_blocks) a.mask_zero[(size_t)slot * 8 + (threadIdx.x & 7)] = 0ull;  // (uniform condition; every lane stores, eight lanes to a word: no lane-guarded store)
    if (ag < 0 || ag >= a.n_agents) return;  // the whole workgroup
    plan_slot<RC, DIMS, FS, MS, KU, SU, SLK, WS, WIDE, WPA>(kp, lay, a, smem, w, slot, ag, cls_in);
}

// one workgroup of LSC_THREADS threads per local agent, threads over its (neighbour, segment) items; every
// item is independent and writes its own 6 (3) rows of the SoA arrays (lsc_core.hpp).  Several wavefronts per agent instead
// of one: an agent of the stress batch has ~440 items of about a thousand instructions each behind scattered reads of its
// neighbours' plans, and a grid of one wavefront per agent leaves the SIMDs with two wavefronts to hide that behind.  With
// 128 VGPRs (the hull points of a lane live in LDS, lsc_core.hpp: HullStore) four wavefronts share a SIMD.  Two wavefronts
// per agent: measured on the stress batch at the end of round 2 -- 64 / 128 / 192 / 256 / 512 threads: 0.072 / 0.062 / 0.072 /
// 0.067 / 0.085 ms (440 items fill 3.4 rounds of 128 to 86 %, and eight workgroups share a CU).
constexpr int LSC_THREADS = 128;
// ... and one wavefront per agent (THREADS = 64) where an agent has at most 64 items -- the synthetic swarms: five neighbours x five
// segments; with 128 threads four fifths of the lanes had nothing to do (32 768 agents: 0.28 ms per step, the largest stage of the
// S64 closed loop)
#ifndef DQ_LSC_OCC
#define DQ_LSC_OCC 4  // wavefronts per SIMD the half-space generator is compiled for (128 VGPRs, 26 of them spilled; 3 = 158 VGPRs and no
                      // spill was measured in round 4: crowd stage 0.099 -> 0.113 ms, stress step 0.133 -> 0.144 -- the fourth wavefront is worth more)
#endif
// OCC: wavefronts per SIMD the build is compiled for.  Launches that fill the chip want four (DQ_LSC_OCC, above); a launch of a
// handful of agents has the SIMDs to itself, and a build without the spills (FEW) is what its latency asks for.
#ifndef DQ_LSC_OCC_FEW
#define DQ_LSC_OCC_FEW 2
#endif
// (the body of dlscqp_lsc_kernel as a device function of a workgroup of THREADS threads: also a stage of dlscqp_fused_kernel)
template <int THREADS>
__device__ __forceinline__ void lsc_agent(const LscGenArgs &g, const int k) {
    if (k >= g.n_local) return;
    const int self = g.local_ids ? g.local_ids[k] : k;
    const int p0 = g.nbr_offsets[k], p1 = g.nbr_counts ? p0 + g.nbr_counts[k] : g.nbr_offsets[k + 1];
    const int per = g.M * g.n1 - g.phi;
    const size_t pair0 = (size_t)p0 + (size_t)k * g.n_dyn;  // pairs (agent, obstacle) before this agent's
    // items in NEIGHBOUR-major order: consecutive lanes read consecutive segments of one neighbour's plan and write one
    // contiguous block of rows.  (Segment-major order, which would keep the lanes of a wavefront on one kind of item -- hull
    // problem vs. the last segment's segment-segment problem -- was measured with the round-1 kernel: the stress batch's
    // step went from 0.51 to 0.74 ms: the scattered plan reads and row writes cost more than the branch saves.  What does
    // pay is the pass below that computes the last segments of all pairs first, one pair per thread: 0.0955 -> 0.0877 ms.)
    const int nn = p1 - p0, items = nn * g.M;
    __shared__ float hull_pts[18 * THREADS];
    const HullStore hull = {hull_pts + threadIdx.x, THREADS};
    // the last-segment geometry of every pair first, one pair per thread (lsc_item_rows says why); pairs beyond the block's
    // capacity are computed in place
    constexpr int LAST_CAP = 160;
    __shared__ LscLast last_geo[LAST_CAP];
    for (int j = threadIdx.x; j < nn && j < LAST_CAP; j += THREADS) last_geo[j] = lsc_last_geometry(g, self, g.nbr_ids[p0 + j]);
    __syncthreads();
    if (g.row_counts) {
        // presolve: rows implied by the agent's own box rows are dropped, the others written contiguously in their original
        // order -- per round of THREADS items an exclusive prefix sum of the kept-row counts over the workgroup (LDS)
        __shared__ int wave_rows[2][THREADS / 64];
        const size_t row0 = (size_t)per * pair0;
        const int all = items + g.n_dyn * g.M;
        int base = 0;  // rows the previous rounds kept (every thread carries its own copy)
        for (int it0 = 0, round = 0; it0 < all; it0 += THREADS, round ^= 1) {
            // (every round derives its addresses from a thread number that went through an empty statement: what is derived from it is
            // computed in the round instead of hoisted in front of the loop and kept in registers -- dlscqp_lsc_pair_kernel says more)
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wave = tid >> 6;
            const HullStore hull = {hull_pts + tid, THREADS};
            const int it = it0 + tid;
            LscRows r;
            r.i0 = 6;
            bool keep[6] = {false, false, false, false, false, false};
            int cnt = 0, m = 0;
            bool tail_skip = false;  // dlscqp_lsc_gen.tail_from: a neighbour far down the list contributes its last segments only
            if (it < items) {
                const int j = it / g.M;
                m = it - j * g.M;
                tail_skip = g.tail_segments > 0 && j >= g.tail_from && m < g.M - g.tail_segments;
                if (!tail_skip) lsc_item_rows(g, self, g.nbr_ids[p0 + j], m, r, hull, j < LAST_CAP ? &last_geo[j] : nullptr);
            } else if (it < all) {
                const int q = it - items, o = q / g.M;
                m = q - o * g.M;
                lsc_dyn_item_rows(g, self, o, m, r);
            }
            if (it < all && !tail_skip) {
                const float *box = g.prune_sfc ? g.prune_sfc + ((size_t)k * g.M + m) * 6 : nullptr;
                cnt = lsc_rows_kept(g, box, r, keep);
            }
            int incl = cnt;  // inclusive scan over the wavefront, then the wavefronts' totals through LDS (one barrier a round)
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int v = __shfl_up(incl, s);
                incl += lane >= s ? v : 0;
            }
            if (lane == 63) wave_rows[round][wave] = incl;
            __syncthreads();
            int before = base;
#pragma unroll
            for (int w = 0; w < THREADS / 64; w++) {
                const int t = wave_rows[round][w];
                before += w < wave ? t : 0;
                base += t;
            }
            if (cnt) lsc_store_rows(g, r, row0 + (size_t)(before + incl - cnt), keep);
        }
        if (threadIdx.x == 0) g.row_counts[k] = base;
        return;
    }
    for (int it = threadIdx.x; it < items; it += THREADS) {
        const int j = it / g.M, m = it - j * g.M;
        lsc_item(g, self, g.nbr_ids[p0 + j], m, (size_t)per * (pair0 + j), hull, j < LAST_CAP ? &last_geo[j] : nullptr);
    }
    for (int it = threadIdx.x; it < g.n_dyn * g.M; it += THREADS) {  // dynamic obstacles after the neighbour agents
        const int o = it / g.M, m = it - o * g.M;
        lsc_dyn_item(g, self, o, m, (size_t)per * (pair0 + nn + o));
    }
}
template <int THREADS, int OCC = DQ_LSC_OCC>
__global__ __launch_bounds__(THREADS, OCC) void dlscqp_lsc_kernel(LscGenArgs g) { lsc_agent<THREADS>(g, (int)blockIdx.x); }

// Two agents per wavefront, for lists so short that one agent leaves most of a wavefront without an item (the synthetic swarms of
// BASELINE.json configs[1]: 4 x 5 + 2 = 22 items -- with a wavefront per agent three fifths of the lanes of the S64 closed loop's
// largest stage after the plan kernel had nothing to do).  Lanes 0-31 work for agent 2 b, lanes 32-63 for agent 2 b + 1; the
// presolve's scan runs over the half-wavefront.  Same items in the same order: the same rows at the same places.  The odd agent
// out repeats the last agent's work and writes nothing.  Presolve path only (row_counts).
#ifndef DQ_LSC_PAIR_OCC
#define DQ_LSC_PAIR_OCC 4  // (rounds 4-5: 3, four wavefronts per SIMD spilled 248 B a lane; round 6: 114 registers with the other agent's points read again, lsc_core.hpp)
#endif
__global__ __launch_bounds__(64, DQ_LSC_PAIR_OCC) void dlscqp_lsc_pair_kernel(LscGenArgs g) {
    const int half = (int)threadIdx.x >> 5, l32 = (int)threadIdx.x & 31;
    const int k = 2 * (int)blockIdx.x + half;
    const bool have = k < g.n_local;
    const int kc = have ? k : g.n_local - 1;
    const int self = g.local_ids ? g.local_ids[kc] : kc;
    const int p0 = g.nbr_offsets[kc], p1 = g.nbr_counts ? p0 + g.nbr_counts[kc] : g.nbr_offsets[kc + 1];
    const int per = g.M * g.n1 - g.phi;
    const size_t pair0 = (size_t)p0 + (size_t)kc * g.n_dyn;
    const int nn = p1 - p0, items = nn * g.M;
    __shared__ float hull_pts[18 * 64];
    const HullStore hull = {hull_pts + threadIdx.x, 64};
    __shared__ LscLast last_geo[2][32];
    if (l32 < nn) last_geo[half][l32] = lsc_last_geometry(g, self, g.nbr_ids[p0 + l32]);
    __syncthreads();
    const size_t row0 = (size_t)per * pair0;
    const int all = items + g.n_dyn * g.M;
    const int all0 = __shfl(all, 0), all1 = __shfl(all, 32);
    const int rounds_to = all0 > all1 ? all0 : all1;  // wave-uniform: both halves walk as many rounds as the longer list needs
    int base = 0;  // rows the previous rounds of this agent kept
    for (int it0 = 0; it0 < rounds_to; it0 += 32) {
#ifndef DQ_LSC_FRESH
#define DQ_LSC_FRESH 1  // every round derives its addresses from a thread number and an agent number that went through an empty statement:
#endif                  // what is derived from them is computed in the round instead of hoisted in front of the loop and kept in registers
#if DQ_LSC_FRESH
        int tid = (int)threadIdx.x, kcl = kc;
        asm volatile("" : "+v"(tid), "+v"(kcl));
        const int half = tid >> 5, l32 = tid & 31, kc = kcl;
        const HullStore hull = {hull_pts + tid, 64};
        const int self = g.local_ids ? g.local_ids[kc] : kc;
#endif
        const int it = it0 + l32;
        LscRows r;
        r.i0 = 6;
        bool keep[6] = {false, false, false, false, false, false};
        int cnt = 0, m = 0;
        bool tail_skip = false;
        if (it < items) {
            const int j = it / g.M;
            m = it - j * g.M;
            tail_skip = g.tail_segments > 0 && j >= g.tail_from && m < g.M - g.tail_segments;
            if (!tail_skip) lsc_item_rows(g, self, g.nbr_ids[p0 + j], m, r, hull, j < 32 ? &last_geo[half][j] : nullptr, true);
        } else if (it < all) {
            const int q = it - items, o = q / g.M;
            m = q - o * g.M;
            lsc_dyn_item_rows(g, self, o, m, r);
        }
        if (it < all && !tail_skip) {
            const float *box = g.prune_sfc ? g.prune_sfc + ((size_t)kc * g.M + m) * 6 : nullptr;
            cnt = lsc_rows_kept(g, box, r, keep);
        }
        int incl = cnt;  // inclusive scan over the half-wavefront (a lane below s never takes what the other half shifted in)
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) {
            const int v = __shfl_up(incl, s);
            incl += l32 >= s ? v : 0;
        }
        if (cnt && have) lsc_store_rows(g, r, row0 + (size_t)(base + incl - cnt), keep);
        base += __shfl(incl, 31 + 32 * half);
    }
    if (l32 == 0 && have) g.row_counts[k] = base;
}

// TrajPlanner::checkWaypointTrap: one wavefront per local agent.  The lanes test the agent's neighbour rows (ballot), the
// verdict is wave-uniform, then the lanes empty the rows of the dynamic obstacles that can reach the waypoint.
struct TrapArgs {
    LscGenArgs g;
    const dlscqp_agent *agents;
    const float *sfc;
    int32_t *out_released;
    double horizon;
};
// (one wavefront; `lane` = its lane.  Also a stage of dlscqp_fused_kernel.)
__device__ __forceinline__ void trap_agent(const TrapArgs &t, const int k, const int lane) {
    const LscGenArgs &g = t.g;
    if (k >= g.n_local) return;
    const int per = g.M * g.n1 - g.phi;
    const int p0 = g.nbr_offsets[k], p1 = g.nbr_counts ? p0 + g.nbr_counts[k] : g.nbr_offsets[k + 1];
    const size_t row0 = (size_t)per * ((size_t)p0 + (size_t)k * g.n_dyn);
    // all rows of the agent: neighbours' and dynamic obstacles' (the latter are ignored by trap_row_excludes); a generator
    // that pruned rows says how many there are
    const int n_rows = g.row_counts ? g.row_counts[k] : (p1 - p0 + g.n_dyn) * per;
    const dlscqp_agent *ag = t.agents + k;
    const float *cg = ag->current_goal_point, *wp = ag->next_waypoint;
    bool out = false;
    for (int j = lane; j < n_rows; j += 64) out |= trap_row_excludes(g, row0 + j, cg, wp);
    const float *box = t.sfc ? t.sfc + ((size_t)k * g.M + (g.M - 1)) * 6 : nullptr;
    bool trapped = __ballot(out) != 0ull;
    trapped = trapped || !point_in_boxes(g, box, wp, cg) || !point_in_boxes(g, box, wp, wp);
    int released = 0;
    if (trapped) {  // uniform
        for (int o = 0; o < g.n_dyn; o++) {
            if (!obstacle_reaches(g.dyn_position + 3 * o, g.dyn_velocity + 3 * o, g.dyn_radius[o], g.dyn_max_acc[o], wp, ag->radius,
                                  t.horizon, g.dyn_uncertainty_horizon))
                continue;
            released++;
            for (int j = lane; j < n_rows; j += 64) {  // the rows of obstacle o: dynamic flag, slack variable o
                const uint32_t key = g.key[row0 + j];
                if (((key >> 12) & 1) && (int)(key >> 16) == o) trap_empty_row(g, row0 + j);
            }
        }
    }
    if (t.out_released && lane == 0) t.out_released[k] = released;
}
__global__ __launch_bounds__(64) void dlscqp_trap_kernel(TrapArgs t) { trap_agent(t, (int)blockIdx.x, (int)threadIdx.x); }

// one WORKGROUP of SFC_THREADS threads per agent grows / shifts its corridor boxes (sfc_core.hpp).
// Control flow is workgroup-uniform (every decision comes out of any_of); the threads share the lattice points of the slab
// under test.  The routine is a chain of a few dozen trial expansions, each a table lookup per lattice point (dependent
// L2 reads): latency, not throughput -- so the points of a slab (several hundred in the reference's 2.5 m high worlds) are
// spread over eight wavefronts instead of being walked 64 at a time by one (10-agent maze step: 0.96 -> see profiles/).
constexpr int SFC_THREADS = 256;  // default: one wavefront per SIMD; DLSCQP_SFC_THREADS = 64 / 128 / 512 picks another instantiation (measurements)
template <int SFC_T>
struct WaveSfcT {
    int lane;
    int *flags;            // LDS [2][SFC_T / 64]: one verdict per wavefront, double-buffered by round
    uint32_t *axis;        // LDS [SFC_T / 64][3][SFC_AXIS_CAP]: the per-axis entries of the box under test, one copy per wavefront
    float *chainv;         // LDS [SFC_T / 64][6][SFC_CHAIN_CAP]
    mutable int chain_n = 1;
    mutable int parity = 0;
    // one word of the lattice-point table (sfc_core.hpp: sfc_corner_gaps) per point.  (A copy of the part of the table an expansion
    // can reach, as bytes in LDS, was built and measured in round 4: 0.178 ms per launch against 0.171 -- a trial expansion is
    // not waiting for this read.)
    __device__ __forceinline__ bool corner_blocked(const SfcWorldView &g, int ic, int jc, int kc, int corner, int gap_max) const {
        const uint32_t word = g.gaps[((size_t)ic * g.n[1] + jc) * g.n[2] + kc];
        return (int)((word >> (4 * corner)) & 15u) <= gap_max;
    }
    // workgroup-wide OR without a lane-divergent region (the library's __syncthreads_or has one): every lane of a
    // wavefront stores the wavefront's ballot verdict to the same word, one barrier, every thread reads all words.  The
    // buffers alternate, so a round's stores cannot meet the previous round's loads.
    __device__ __forceinline__ bool block_or(bool v) const {
        constexpr int NW = SFC_T / 64;
        int *fl = flags + parity * NW;
        parity ^= 1;
        fl[lane >> 6] = __ballot(v) != 0ull ? 1 : 0;
        __syncthreads();
        int r = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) r |= fl[k];
        return r != 0;
    }
    // every thread takes its share of the points and ORs its verdicts; ONE workgroup-wide OR at the end.  (Until round 4 the
    // workgroup agreed after every SFC_T points and left at the first blocked one: a box of 11 000 lattice points was 22 table
    // reads one after the other, each with a barrier behind it, and almost every box that is tested is free.)
    template <class F>
    __device__ __forceinline__ bool any_of(int n, F f) const {
        bool acc = false;
#ifndef SFC_ANY_UNROLL
#define SFC_ANY_UNROLL 1  // (rounds 2-5: 4.  The point test is ~200 instructions and any_of has a dozen call sites: unrolled four times the
#endif                    // kernel was 70 k instructions of which a ten-workgroup launch fetches most cold; 1: corridor stage of the maze / forest step 0.038 -> 0.035 ms, library 2.8 MB smaller)
#pragma unroll SFC_ANY_UNROLL
        for (int base = 0; base < n; base += SFC_T) {
            const int idx = base + lane;
            acc |= f(idx < n ? idx : n - 1) & (idx < n);  // every thread evaluates a valid point
        }
        return block_or(acc);
    }
    // sfc_core.hpp: sfc_axis_entry of every lattice index of the box, per axis.  Each wavefront keeps its own copy (lane l computes
    // index l of the three axes): no barrier, a wavefront's LDS accesses are served in order.
    // sfc_expand's lanes: every wavefront of the workgroup evaluates the same 64 trials (the control flow stays workgroup-uniform;
    // the wavefronts sit on different SIMDs, the repetition costs nothing) and gets the same two masks
    template <class F>
    __device__ __forceinline__ void lanes(F f, uint64_t &m0, uint64_t &m1) const {
        const int fl = f(lane & 63);
        m0 = __ballot((fl & 1) != 0);
        m1 = __ballot((fl & 2) != 0);
    }
    // where a face of the box is after c moves (sfc_core.hpp: SfcTurns): lane d % 6 walks direction d, one copy per wavefront
    __device__ __forceinline__ void chain_fill(const float *init, double res, int n) const {
        float *t = chainv + (lane >> 6) * (6 * SFC_CHAIN_CAP);
        const int d = (lane & 63) % 6;
        float v = init[5];
        v = d == 4 ? init[4] : v;
        v = d == 3 ? init[3] : v;
        v = d == 2 ? init[2] : v;
        v = d == 1 ? init[1] : v;
        v = d == 0 ? init[0] : v;
        const double step = d < 3 ? -res : res;
        for (int c = 0; c < n; c++) {
            t[d * SFC_CHAIN_CAP + c] = v;
            v = (float)((double)v + step);
        }
        chain_n = n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ float chain(int d, int c) const {
        c = c < 0 ? 0 : (c >= chain_n ? chain_n - 1 : c);
        return chainv[(lane >> 6) * (6 * SFC_CHAIN_CAP) + d * SFC_CHAIN_CAP + c];
    }
    __device__ __forceinline__ float rcp(float x) const { return __builtin_amdgcn_rcpf(x); }
    __device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ float uniform_f(float v) const { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
    __device__ __forceinline__ void axis_fill(const SfcWorldView &g, double inv_res, const float *b) const {
        uint32_t *t = axis + (lane >> 6) * (3 * SFC_AXIS_CAP);
        const int l = lane & 63;
#pragma unroll
        for (int k = 0; k < 3; k++) t[k * SFC_AXIS_CAP + l] = sfc_axis_entry(g, inv_res, k, b[k], l);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ uint32_t axis_entry(int k, int it) const {
        return axis[(lane >> 6) * (3 * SFC_AXIS_CAP) + k * SFC_AXIS_CAP + it];
    }
    // largest v (0..8) over the lanes of the wavefront, as a wave-uniform value: ballots and scalar selects
    __device__ __forceinline__ int uniform_max8(int v) const {
        int r = 0;
#pragma unroll
        for (int t = 1; t <= 8; t++) r = __ballot(v >= t) != 0ull ? t : r;
        return r;
    }
    __device__ __forceinline__ void load6(const float *src, float *dst) const {
        for (int k = 0; k < 6; k++) dst[k] = src[k];
    }
    __device__ __forceinline__ void store6(float *dst, const float *src) const {
        // src is the same in every lane; every lane stores component lane % 6 (no lane-guarded region: the uniform state
        // of sfc_core.hpp lives in VGPRs, see "NO DIVERGENCE" in gi_core.hpp)
        const int k = lane % 6;
        float v = src[5];
        v = k == 4 ? src[4] : v;
        v = k == 3 ? src[3] : v;
        v = k == 2 ? src[2] : v;
        v = k == 1 ? src[1] : v;
        v = k == 0 ? src[0] : v;
        dst[k] = v;
    }
    __device__ __forceinline__ void fence() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // the agent's boxes at hand (sfc_construct): old[M][6] as read, kept[M][6] as they will be written.  Every wavefront reads
    // only `old` and writes all of `kept` itself (the same values as the others): no barrier after the one that ends the load.
    float *boxes;  // LDS [2][SFC_BOXES_CAP][6]
    __device__ __forceinline__ void boxes_load(const float *sfcs, int M) const {
        const int n = 6 * M;
        for (int i0 = 0; i0 < n; i0 += SFC_T) {  // uniform trip count; a lane past the end repeats the last element
            const int i = i0 + lane < n ? i0 + lane : n - 1;
            boxes[i] = sfcs[i];
        }
        fence();
    }
    __device__ __forceinline__ void box(int m, float *out) const {
        for (int k = 0; k < 6; k++) out[k] = boxes[6 * m + k];
    }
    __device__ __forceinline__ void box_keep(int m, const float *in) const {
        for (int k = 0; k < 6; k++) boxes[6 * SFC_BOXES_CAP + 6 * m + k] = in[k];  // every lane, the same words, the same values
    }
    __device__ __forceinline__ void boxes_store(float *sfcs, int M) const {
        const int n = 6 * M;
        for (int i0 = 0; i0 < n; i0 += SFC_T) {
            const int i = i0 + lane < n ? i0 + lane : n - 1;
            sfcs[i] = boxes[6 * SFC_BOXES_CAP + i];
        }
    }
};

struct SfcGenArgs {
    SfcWorldView world;
    int n_agents, initialize, M, n1;
    double grid_resolution, dt;
    const double *radius, *max_vel;
    const float *position, *initial_traj, *goals, *waypoints;
    float *sfcs;
    int32_t *out_code;
};

// (the body of dlscqp_sfc_kernel as a device function of a workgroup of SFC_T threads: also a stage of dlscqp_fused_kernel)
template <int SFC_T>
__device__ __forceinline__ void sfc_agent(const SfcGenArgs a, const int ag) {  // (by value: by reference this compiler's register allocator crashed on the 64-thread instantiation)
    if (ag >= a.n_agents) return;
    __shared__ int sfc_flags[2 * (SFC_T / 64)];
    __shared__ uint32_t sfc_axis[(SFC_T / 64) * 3 * SFC_AXIS_CAP];
    WaveSfcT<SFC_T> w;
    w.lane = (int)threadIdx.x;
    w.flags = sfc_flags;
    __shared__ float sfc_chain[(SFC_T / 64) * 6 * SFC_CHAIN_CAP];
    __shared__ float sfc_boxes[2 * SFC_BOXES_CAP * 6];
    w.axis = sfc_axis;
    w.chainv = sfc_chain;
    w.boxes = sfc_boxes;
    SfcParams p;
    p.margin = a.radius[ag];
    p.grid_resolution = a.grid_resolution;
    p.agent_max_vel = a.max_vel[ag];
    p.dt = a.dt;
    sfc_prepare(a.world, p);
    float *sfcs = a.sfcs + (size_t)ag * a.M * 6;
    int code;
    if (a.initialize) {
        code = sfc_initialize(w, a.world, p, a.position + 3 * (size_t)ag, a.M, sfcs) ? 0 : -1;
    } else {
        code = sfc_construct(w, a.world, p, a.initial_traj + (size_t)ag * a.M * a.n1 * 3, a.M, a.n1, a.goals + 3 * (size_t)ag,
                             a.waypoints + 3 * (size_t)ag, sfcs);
    }
    if (a.out_code) a.out_code[ag] = code;  // every lane, same word
}
template <int SFC_T>
__global__ __launch_bounds__(SFC_T) void dlscqp_sfc_kernel(SfcGenArgs a) { sfc_agent<SFC_T>(a, (int)blockIdx.x); }

typedef void (*kernel_t)(KParams, MemLayout, KernelArgs);
struct KernelChoice {
    kernel_t fn;
    int id, KU, SU;
    bool by_kind = false;
    kernel_t fn_wide = nullptr;  // the one-wavefront-per-SIMD build of the same instantiation, where there is one
    kernel_t fn_wpa = nullptr;   // the build with helper wavefronts (DQ_WPA per agent), for instantiations that price rows beyond their cache
};
constexpr int DQ_WPA = 4;
constexpr int DQ_WPA_MAX_SLOTS = 512;  // launches of at most this many agents use the builds with helper wavefronts
constexpr int DQ_WIDE_MAX_SLOTS = 1024;  // launches of at most this many agents use it: one agent per SIMD of the chip
constexpr int DQ_ID_HBM = 7;  // the runtime-shape two-register instantiation is the one that also exists as WS_HBM
#ifdef DLSCQP_ONLY_S64
const kernel_t dlscqp_plan_kernel_hbm = nullptr;
#else
const kernel_t dlscqp_plan_kernel_hbm = dlscqp_plan_kernel<2, 0, 0, 0, DQ_G2_KU, DQ_G2_SU, -1, WS_HBM>;
#endif

// The WIDE build of an instantiation, for those that have one: two wavefronts per SIMD normally, and register caches
// beyond the all-in-registers instantiation (which does not spill).
template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK>
kernel_t wide_kernel() {
    if constexpr (RC == 1 && !(DIMS == 2 && FS == 28) && !(KU == 2 && SU == 0))
        return dlscqp_plan_kernel<RC, DIMS, FS, MS, KU, SU, SLK, WS_LDS, true>;
    else
        return nullptr;
}

template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK>
kernel_t wpa_kernel() {
    if constexpr (!(KU == 2 && SU == 0))
        return dlscqp_plan_kernel<RC, DIMS, FS, MS, KU, SU, SLK, WS_LDS, false, DQ_WPA>;
    else
        return nullptr;
}

// Kernel instantiation for a batch: the shared table in gi_core.hpp (DQ_INSTANCES / select_instance).
KernelChoice select_kernel(const KParams &kp, int max_slack_obs, int kmax, bool force_runtime_shape = false) {
    static const bool force_env = std::getenv("DLSCQP_FORCE_GENERIC") != nullptr;  // debugging aid
    const bool force_generic = force_env || force_runtime_shape;
#ifdef DLSCQP_ONLY_S64  // developer aid (tests/tools/s64_only.sh): only the S64 instantiation -- a quarter of the compile time; run with DLSCQP_NO_SELFTEST=1
    (void)force_generic; (void)max_slack_obs; (void)kmax;
    return {dlscqp_plan_kernel<1, 3, 13, 5, 2, 0, 0>, 0, 2, 0, true};
#elif defined(DLSCQP_ONLY_G2)  // debugging aid: compile only the generic two-register instantiation
    (void)force_generic;
    return {dlscqp_plan_kernel<2, 0, 0, 0, DQ_G2_KU, DQ_G2_SU, -1>, 7, DQ_G2_KU, DQ_G2_SU, false};
#else
    const InstanceChoice ic = select_instance(kp, max_slack_obs, kmax, force_generic);
    switch (ic.id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID: return {dlscqp_plan_kernel<RC, DIMS, FS, MS, KU, SU, SLK>, ID, KU, SU, (DIMS) > 0, wide_kernel<RC, DIMS, FS, MS, KU, SU, SLK>(), \
                     wpa_kernel<RC, DIMS, FS, MS, KU, SU, SLK>()};
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    default: return {nullptr, -1, 0, 0, false};
    }
#endif
}

thread_local std::string g_last_error;
struct NcclId { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES), passed by value to ncclCommInitRank

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(expr)                              \
    do {                                           \
        if (!hip_ok((expr), #expr)) return DLSCQP_ERR_HIP; \
    } while (0)

constexpr int64_t LDS_LIMIT = 160 * 1024;

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct dlscqp_ctx {
    dlscqp_config cfg;
    int dim, M, n, phi;  // cached at construction like TrajOptimizer (traj_optimizer.cpp:7-11)
    double dt;
    HostTables tb;
    KParams kp;  // with device table pointers
    double *d_tabPhi = nullptr, *d_tabKIW = nullptr, *d_tabG = nullptr, *d_tabN = nullptr, *d_tabX0s = nullptr, *d_tabX0g = nullptr, *d_tabNN = nullptr, *d_tabSIW = nullptr;
    Slot *d_slots = nullptr;
    hipStream_t stream = nullptr;
    int max_dyn_lds[3 * (DQ_N_INSTANCES + 1)] = {};  // second third: the WIDE builds, last third: the builds with helper wavefronts
    int *d_order_scratch = nullptr;  // dlscqp_launch_order_device: 64 counts + 64 cursors
    void *work = nullptr;  // working sets of the WS_HBM kernel
    int force_instance = 0;  // dlscqp_selftest: 1 = the runtime-shape instantiation, 2 = its variant with the working set in HBM
    int force_build = 0;     // dlscqp_selftest: 1 = the standard build of the instantiation, 2 = its WIDE build, 3 = the build with helper wavefronts
    int32_t *nbr_scratch = nullptr;  // dlscqp_neighbours_device: cell counts / starts, agents' cells and slots, agents by cell
    size_t nbr_scratch_ints = 0;
    unsigned int *fused_grid = nullptr;  // dlscqp_fused_step_device: the grid barrier's two words (arrivals, generation)
    int fused_dyn_lds[DQ_N_INSTANCES + 1] = {};
    int nbr_cells_clean = -1;  // cell counts [0, nbr_cells_clean] of nbr_scratch are zero (what the last complete search left); -1: unknown
    size_t work_bytes = 0;
    void *spill = nullptr;  // spill records of a launch whose batch names no workspace (grown on demand, never inside a stream capture)
    size_t spill_bytes = 0;
    // staging for the host-buffer entry points
    unsigned char *h_stage = nullptr, *d_stage = nullptr;
    size_t stage_cap = 0;
    // event pairs for kernel timing
    std::vector<hipEvent_t> ev0, ev1;
    int ev_used = 0;
};

static int upload_tables(dlscqp_ctx *c) {
    auto up = [&](void **dptr, const void *src, size_t bytes) -> bool {
        if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
        if (!hip_ok(hipMalloc(dptr, std::max<size_t>(bytes, 16)), "hipMalloc(tables)")) return false;
        return hip_ok(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice), "hipMemcpy(tables)");
    };
    if (!up((void **)&c->d_tabG, c->tb.tabG.data(), c->tb.tabG.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabN, c->tb.tabN.data(), c->tb.tabN.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabX0s, c->tb.tabX0s.data(), c->tb.tabX0s.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabX0g, c->tb.tabX0g.data(), c->tb.tabX0g.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabNN, c->tb.tabNN.data(), c->tb.tabNN.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabSIW, c->tb.tabSIW.data(), c->tb.tabSIW.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabKIW, c->tb.tabKIW.data(), c->tb.tabKIW.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_tabPhi, c->tb.tabPhi.data(), c->tb.tabPhi.size() * 8)) return DLSCQP_ERR_HIP;
    if (!up((void **)&c->d_slots, c->tb.slots.data(), c->tb.slots.size() * sizeof(Slot))) return DLSCQP_ERR_HIP;
    c->kp = c->tb.kp;
    c->kp.tabG = c->d_tabG;
    c->kp.tabN = c->d_tabN;
    c->kp.tabX0s = c->d_tabX0s;
    c->kp.tabX0g = c->d_tabX0g;
    c->kp.tabNN = c->d_tabNN;
    c->kp.tabSIW = c->d_tabSIW;
    c->kp.tabKIW = c->d_tabKIW;
    c->kp.tabPhi = c->d_tabPhi;
    c->kp.slots = c->d_slots;
    return DLSCQP_SUCCESS;
}

extern "C" const char *dlscqp_last_error(void) { return g_last_error.c_str(); }

extern "C" int dlscqp_create(const dlscqp_config *cfg, dlscqp_ctx **out_ctx) {
    g_last_error.clear();
    if (!cfg || !out_ctx) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (cfg->abi_version != DLSCQP_ABI_VERSION) { g_last_error = "abi_version mismatch"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_last_error = "no HIP device: the trajectory optimizer has no CPU path";
        return DLSCQP_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_last_error = "device ordinal out of range"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    dlscqp_ctx *c = new dlscqp_ctx();
    c->cfg = *cfg;
    c->dim = cfg->world_dimension; c->M = cfg->M; c->n = cfg->n; c->phi = cfg->phi; c->dt = cfg->dt;
    std::string err;
    if (!build_tables(*cfg, c->dim, c->M, c->n, c->phi, c->dt, c->tb, err)) {
        g_last_error = "[TrajOptimizer] " + err;
        delete c;
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { g_last_error = "hipSetDevice failed"; delete c; return DLSCQP_ERR_HIP; }
    if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) { delete c; return DLSCQP_ERR_HIP; }
    int rc = upload_tables(c);
    if (rc != DLSCQP_SUCCESS) { dlscqp_destroy(c); return rc; }
    // (dlscqp_fused_step_device's grid barrier: two words, zero; here, so that the first fused step may already be a captured one)
    if (!hip_ok(hipMalloc((void **)&c->fused_grid, 256), "hipMalloc") || !hip_ok(hipMemset(c->fused_grid, 0, 256), "hipMemset")) { dlscqp_destroy(c); return DLSCQP_ERR_HIP; }
    if (!std::getenv("DLSCQP_NO_SELFTEST")) {
        rc = dlscqp_selftest(c);
        if (rc != DLSCQP_SUCCESS) {
            const std::string why = g_last_error;
            dlscqp_destroy(c);
            g_last_error = why;
            return rc;
        }
    }
    *out_ctx = c;
    return DLSCQP_SUCCESS;
}

extern "C" void dlscqp_destroy(dlscqp_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto e : c->ev0) (void)hipEventDestroy(e);
    for (auto e : c->ev1) (void)hipEventDestroy(e);
    if (c->d_tabG) (void)hipFree(c->d_tabG);
    if (c->d_tabN) (void)hipFree(c->d_tabN);
    if (c->d_tabX0s) (void)hipFree(c->d_tabX0s);
    if (c->d_tabX0g) (void)hipFree(c->d_tabX0g);
    if (c->d_tabNN) (void)hipFree(c->d_tabNN);
    if (c->d_tabSIW) (void)hipFree(c->d_tabSIW);
    if (c->d_tabKIW) (void)hipFree(c->d_tabKIW);
    if (c->d_tabPhi) (void)hipFree(c->d_tabPhi);
    if (c->d_slots) (void)hipFree(c->d_slots);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->work) (void)hipFree(c->work);
    if (c->spill) (void)hipFree(c->spill);
    if (c->d_order_scratch) (void)hipFree(c->d_order_scratch);
    if (c->nbr_scratch) (void)hipFree(c->nbr_scratch);
    if (c->fused_grid) (void)hipFree(c->fused_grid);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int dlscqp_update_param(dlscqp_ctx *c, const dlscqp_config *cfg) {
    g_last_error.clear();
    if (!c || !cfg) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    // TrajOptimizer::updateParam replaces param but keeps the cached dim/M/n/phi/dt
    // (traj_optimizer.cpp:167-169 vs :7-11)
    dlscqp_config merged = *cfg;
    merged.device = c->cfg.device;
    HostTables tb;
    std::string err;
    if (!build_tables(merged, c->dim, c->M, c->n, c->phi, c->dt, tb, err)) {
        g_last_error = "[TrajOptimizer] " + err;
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->cfg = merged;
    c->tb = std::move(tb);
    return upload_tables(c);
}

extern "C" int64_t dlscqp_lds_bytes(const dlscqp_ctx *c, int32_t max_slack_obs, int32_t max_rows_per_agent) {
    if (!c) return -1;
    KernelChoice kc = select_kernel(c->tb.kp, max_slack_obs, max_rows_per_agent);
    if (!kc.fn) return -1;
    // the smallest working set a launch of this shape can run with: eight column records in it, the others spilled (choose_layout)
    return make_layout(c->tb.kp, max_slack_obs, max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, 0, 8).bytes;
}

extern "C" int64_t dlscqp_workspace_bytes(const dlscqp_ctx *c, int32_t n_agents, int32_t max_slack_obs, int32_t max_rows_per_agent) {
    if (!c || n_agents < 0) return -1;
    KernelChoice kc = select_kernel(c->tb.kp, max_slack_obs, max_rows_per_agent);
    if (!kc.fn) return -1;
    // the smallest number of records the launcher ever keeps in the working set (choose_layout) gives the largest spill record
    const MemLayout lay = make_layout(c->tb.kp, max_slack_obs, max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, 0, 8);
    const size_t stride = ((size_t)lay.spill_doubles + 31) & ~(size_t)31;
    return (int64_t)(stride * 8 * (size_t)(64 * ((n_agents + 63) / 64)));
}

extern "C" int32_t dlscqp_conflict_doubles(const dlscqp_ctx *c, int32_t max_slack_obs) {
    if (!c || max_slack_obs < 0) return -1;
    return 4 + 2 * (c->tb.kp.dim * c->tb.kp.f + c->tb.kp.M * max_slack_obs + 1);
}

// ---- the reference's failure path on the host (model_rows.hpp): model export, conflict ----------------------------------------------
// (the shape the context was constructed with: TrajOptimizer caches dim / M / n / phi / dt, updateParam does not refresh them)
static dlscqp_config model_config(const dlscqp_ctx *c) {
    dlscqp_config cf = c->cfg;
    cf.world_dimension = c->dim; cf.M = c->M; cf.n = c->n; cf.phi = c->phi; cf.dt = c->dt;
    return cf;
}
extern "C" int dlscqp_write_lp(const dlscqp_ctx *c, const dlscqp_agent *agent, const dlscqp_lsc_row *rows, int32_t n_rows, const float *sfc,
                               int32_t max_slack_obs, const float *goal, const char *path) {
    g_last_error.clear();
    if (!c || !agent || !path || n_rows < 0 || (n_rows > 0 && !rows) || max_slack_obs < 0) { g_last_error = "invalid argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    HostModel md;
    std::string err;
    const dlscqp_config cf = model_config(c);
    if (!build_host_model(cf, c->tb.kp, *agent, rows, n_rows, sfc, max_slack_obs, goal, md, err) || !write_lp(md, path, err)) {
        g_last_error = "[TrajOptimizer] " + err;
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_conflict_decode(const dlscqp_ctx *c, const dlscqp_agent *agent, const dlscqp_lsc_row *rows, int32_t n_rows, const float *sfc,
                                      int32_t max_slack_obs, const float *goal, const double *record, dlscqp_conflict_member *out, int32_t capacity,
                                      int32_t *n_out, int32_t *n_con_out, double *gap_out) {
    g_last_error.clear();
    if (!c || !agent || !record || !n_out || n_rows < 0 || (n_rows > 0 && !rows) || capacity < 0 || (capacity > 0 && !out)) {
        g_last_error = "invalid argument";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    *n_out = 0;
    HostModel md;
    std::string err;
    const dlscqp_config cf = model_config(c);
    if (!build_host_model(cf, c->tb.kp, *agent, rows, n_rows, sfc, max_slack_obs, goal, md, err)) {
        g_last_error = "[TrajOptimizer] " + err;
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    if (n_con_out) *n_con_out = (int32_t)md.con.size();
    const int n = (int)record[0];
    if (gap_out) *gap_out = n > 0 ? record[1] : 0.0;
    if (n <= 0) return DLSCQP_SUCCESS;
    if (n > capacity) { g_last_error = "dlscqp_conflict_decode: capacity is smaller than the certificate"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    for (int t = 0; t < n; t++) {
        double scale = 1.0;
        if (!conflict_member(cf, c->tb.kp, c->tb.slots.data(), md, sfc, (uint32_t)record[4 + 2 * t], out[t], scale, err)) {
            g_last_error = err;
            return DLSCQP_ERR_INVALID_ARGUMENT;
        }
        out[t].lambda = record[5 + 2 * t] / scale;
    }
    *n_out = n;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_set_timing(dlscqp_ctx *c, int32_t capacity) {
    g_last_error.clear();
    if (!c || capacity < 0) return DLSCQP_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->cfg.device);
    while ((int)c->ev0.size() < capacity) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        c->ev0.push_back(a);
        c->ev1.push_back(b);
    }
    while ((int)c->ev0.size() > capacity) {
        (void)hipEventDestroy(c->ev0.back());
        (void)hipEventDestroy(c->ev1.back());
        c->ev0.pop_back();
        c->ev1.pop_back();
    }
    c->ev_used = 0;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_get_kernel_times(dlscqp_ctx *c, float *ms_out, int32_t capacity, int32_t *n_out) {
    g_last_error.clear();
    if (!c || !n_out) return DLSCQP_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->cfg.device);
    int n = std::min<int>(c->ev_used, capacity);
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipEventSynchronize(c->ev1[i]));
        HIP_TRY(hipEventElapsedTime(&ms_out[i], c->ev0[i], c->ev1[i]));
    }
    *n_out = n;
    return DLSCQP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------
// Working set of a launch.  The column records of the active rows (gi_core.hpp: MemLayout) are the largest part of it: r_max of them
// would leave a CU's LDS to two or three agents, and a launch of thousands of agents wants eight (four in the builds compiled for
// one wavefront per SIMD).  So: `budget` = the LDS one agent may have at the occupancy the launch can use (a launch of at most 256
// agents gets a CU each); if every record fits it, no spill; else qz records stay in the working set -- 16, which covers all but a
// few per mille of the agents of the synthetic swarms; 8 where the budget is shared with rows beyond the register cache (a
// crowd prices ~700 rows from LDS in every pass; DLSCQP_QZ overrides: measurements) -- and the others go to a spill record in HBM.
struct LayoutChoice {
    MemLayout lay;
    int budget;
};
static LayoutChoice choose_layout(const KParams &kp, const dlscqp_device_batch *b, const KernelChoice &kc, int n_slots, int max_per_cu,
                                  bool rows_in_lds, int64_t lds_limit = LDS_LIMIT) {
    int per_cu = (n_slots + 255) / 256;
    per_cu = per_cu < 1 ? 1 : (per_cu > max_per_cu ? max_per_cu : per_cu);
    const int budget = (int)(lds_limit / per_cu) - 512;
    const bool cold = b->max_rows_per_agent > 64 * kc.KU;
    const int row_budget = (cold && rows_in_lds) ? budget : 0;
    static const char *qz_env = getenv("DLSCQP_QZ");  // records in the working set: 0 = all of them, whatever the occupancy
    LayoutChoice out;
    out.budget = budget;
    const MemLayout full = make_layout(kp, b->max_slack_obs, b->max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, 0, 0);
    if (qz_env ? atoi(qz_env) <= 0 : full.bytes <= budget) {
        out.lay = make_layout(kp, b->max_slack_obs, b->max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, row_budget, 0);
        return out;
    }
    int qz_want = qz_env ? atoi(qz_env) : (cold ? 8 : 16);
    if (!qz_env && qz_want > 8 && make_layout(kp, b->max_slack_obs, b->max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, 0, qz_want).bytes > budget) qz_want = 8;
    out.lay = make_layout(kp, b->max_slack_obs, b->max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, row_budget, qz_want);
    return out;
}

// dlscqp_advance checked and turned into the kernel's arguments (dlscqp_advance_device, dlscqp_device_batch.advance_after)
static int advance_args(const dlscqp_ctx *c, const dlscqp_advance *adv, AdvanceArgs &g) {
    if (!adv->status || !adv->goal_status || !adv->ctrl || !adv->goal || !adv->plans || !adv->current_goal || !adv->agents) {
        g_last_error = "dlscqp_advance: null buffer";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    if (c->M * (c->n + 1) * 3 > 512) { g_last_error = "plans of more than 512 floats are not supported"; return DLSCQP_ERR_UNSUPPORTED; }
    if (adv->out_boxes && !adv->initial_traj) { g_last_error = "out_boxes are the boxes of initial_traj: null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (adv->out_host_pack && adv->n_alert_obs > 0 && (!adv->alert_slack_cost || adv->alert_slack_stride < adv->n_alert_obs)) {
        g_last_error = "out_host_pack with n_alert_obs > 0 needs alert_slack_cost [n][alert_slack_stride >= n_alert_obs]";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    if (adv->n_alert_obs < 0) { g_last_error = "n_alert_obs < 0"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    g.a = *adv;
    g.M = c->M;
    g.n1 = c->n + 1;
    g.s1 = (float)((double)c->n / c->dt);
    g.s2 = (float)((double)(c->n - 1) / c->dt);
    return DLSCQP_SUCCESS;
}

// dlscqp_fused_step_device: the plan stage of the fused kernel is prepared by launch() -- same checks, same layout rules, same arguments --
// and handed back instead of being launched
struct FusedPlan {
    int id = -1;         // instantiation (DQ_INSTANCES)
    int static_lds = 0;  // LDS the fused kernel's other stages hold (in): the working set gets what is left of a CU's
    KParams kp;
    MemLayout lay;
    KernelArgs a;
};

static int launch(dlscqp_ctx *c, const dlscqp_device_batch *b, hipStream_t stream, FusedPlan *fused = nullptr) {
    if (b->n_agents <= 0) return DLSCQP_SUCCESS;
    if (b->max_slack_obs < 0 || b->max_rows_per_agent < 0) { g_last_error = "negative size"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (c->kp.use_sfc && !b->sfc_boxes) {  // same rule as the host-buffer entry points: never plan without the corridor rows
        g_last_error = "world_use_octomap is set but sfc_boxes is NULL";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    KernelChoice kc = select_kernel(c->kp, b->max_slack_obs, b->max_rows_per_agent, c->force_instance != 0);
    if (!kc.fn) {
        g_last_error = "reduced dimension above 128 (dim*f + M*max_slack_obs) is not supported";
        return DLSCQP_ERR_UNSUPPORTED;
    }
    // (the working set is sized below, once the build and the number of launch slots are known; this first look only decides
    // whether it can live in LDS at all: with eight column records in the working set, the smallest the layout goes)
    MemLayout lay = make_layout(c->kp, b->max_slack_obs, b->max_rows_per_agent, kc.KU, kc.SU, kc.by_kind, 0, 8);
    static const bool force_hbm = getenv("DLSCQP_DEBUG_WS") && std::string(getenv("DLSCQP_DEBUG_WS")) == "hbm";  // tests
    const bool ws_hbm = lay.bytes > LDS_LIMIT || force_hbm || c->force_instance == 2;
    KernelArgs a;
    a.work = nullptr;
    a.work_stride = 0;
    a.spill = nullptr;
    a.spill_stride = 0;
    if (ws_hbm) {
        // Working set beyond the LDS of a CU: the runtime-shape kernel with its working set in a per-slot HBM buffer.  The
        // buffer belongs to the context: launches of one context that use it must not overlap.
        kc = KernelChoice{dlscqp_plan_kernel_hbm, DQ_ID_HBM, DQ_G2_KU, DQ_G2_SU, false};
        lay = make_layout(c->kp, b->max_slack_obs, b->max_rows_per_agent, kc.KU, kc.SU, false);  // (every column record in the working set)
        a.work_stride = ((size_t)lay.bytes + 255) & ~(size_t)255;
        const size_t need = a.work_stride * (size_t)(64 * ((b->n_agents + 63) / 64));  // one per launch slot (hint_iters: 64 per block)
        if (need > c->work_bytes) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(stream, &cap);
            if (cap != hipStreamCaptureStatusNone) { g_last_error = "the HBM working sets must grow: not inside a stream capture (launch once eagerly first)"; return DLSCQP_ERR_INVALID_ARGUMENT; }
            HIP_TRY(hipDeviceSynchronize());
            if (c->work) (void)hipFree(c->work);
            c->work = nullptr;
            c->work_bytes = 0;
            HIP_TRY(hipMalloc(&c->work, need));
            c->work_bytes = need;
        }
        a.work = static_cast<unsigned char *>(c->work);
    }
    a.n_agents = b->n_agents;
    a.slot0 = b->launch_count > 0 ? b->launch_offset : 0;
    a.n_slots_launch = b->launch_count > 0 ? b->launch_count : b->n_agents;
    // launch order found inside the kernel (dlscqp_device_batch.hint_iters): slots are (rank, block) pairs, 64 per block
    a.hint[0] = a.hint[1] = a.hint[2] = nullptr;
    a.n_blocks = (b->n_agents + 63) / 64;
    a.blocks_magic = 0;
    if (b->hint_steps != 0 && !b->agent_order) {
        if (b->hint_steps < 0 || b->hint_steps > 3 || !b->hint_iters[0]) { g_last_error = "hint_iters: one to three steps of iteration counts"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        if (b->launch_count > 0) { g_last_error = "hint_iters cannot be combined with launch_offset / launch_count"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        for (int k = 0; k < b->hint_steps; k++) {
            if (!b->hint_iters[k]) { g_last_error = "hint_iters: null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
            a.hint[k] = b->hint_iters[k];
        }
        a.n_slots_launch = 64 * a.n_blocks;
        // slot / n_blocks as one multiplication where that is exact: slot * n_blocks < 2^32 for every slot of the launch
        if (a.n_blocks > 1 && (unsigned long long)a.n_slots_launch * (unsigned long long)a.n_blocks < (1ull << 32))
            a.blocks_magic = (unsigned int)(((1ull << 32) + (unsigned long long)a.n_blocks - 1) / (unsigned long long)a.n_blocks);
    }
    // ... or from the class masks the previous step's launch left behind (dlscqp_device_batch.hint_mask_in): the same slots
    a.mask_in = nullptr; a.mask_out = nullptr; a.mask_zero = nullptr;
    if (b->hint_mask_in || b->hint_mask_out || b->hint_mask_zero) {
        // (launch_offset / launch_count then count LAUNCH SLOTS of the 64 n_blocks: slots [0, R n_blocks) are every block's R most
        // expensive agents -- a caller can plan those in a launch of their own, e.g. one small enough to get helper wavefronts)
        if ((b->hint_mask_out && (b->hint_mask_out == b->hint_mask_in || b->hint_mask_out == b->hint_mask_zero)) ||
            (b->hint_mask_in && b->hint_mask_in == b->hint_mask_zero)) { g_last_error = "hint_mask_in / _out / _zero must be three different tables"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        a.mask_out = reinterpret_cast<unsigned long long *>(b->hint_mask_out);
        a.mask_zero = reinterpret_cast<unsigned long long *>(b->hint_mask_zero);
        if (b->hint_mask_in && !b->agent_order && !a.hint[0]) {
            a.mask_in = reinterpret_cast<const unsigned long long *>(b->hint_mask_in);
            if (b->launch_count <= 0) a.n_slots_launch = 64 * a.n_blocks;
            if (a.n_blocks > 1 && (unsigned long long)a.n_slots_launch * (unsigned long long)a.n_blocks < (1ull << 32))
                a.blocks_magic = (unsigned int)(((1ull << 32) + (unsigned long long)a.n_blocks - 1) / (unsigned long long)a.n_blocks);
        }
    }
    a.ctrl_stride = c->M * (c->n + 1) * 3; a.goal_stride = 3; a.status_stride = 1;
    a.adv_on = 0;
    if (b->advance_after) {
        // the bookkeeping between two launches in the plan kernel's epilogue: it reads what this launch writes, as dense arrays
        const dlscqp_advance *adv = b->advance_after;
        if (b->out_record_stride != 0 || !b->do_goal || !b->do_traj || adv->n_agents != b->n_agents || adv->status != b->out_status ||
            adv->goal_status != b->out_goal_status || adv->ctrl != b->out_ctrl || adv->goal != b->out_goal ||
            (adv->n_alert_obs > 0 && adv->alert_slack_cost != b->out_slack_cost)) {
            g_last_error = "advance_after: for a launch of goal LP + QP whose dense outputs (out_status, out_goal_status, out_ctrl, out_goal, "
                           "out_slack_cost) are the structure's inputs, over the same agents";
            return DLSCQP_ERR_INVALID_ARGUMENT;
        }
        const int rc = advance_args(c, adv, a.adv);
        if (rc != DLSCQP_SUCCESS) return rc;
        a.adv_on = 1;
    }
    if (b->out_record_stride > 0) {
        if (b->out_record_stride < c->M * (c->n + 1) * 3) { g_last_error = "out_record_stride is smaller than a plan"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        a.ctrl_stride = a.goal_stride = a.status_stride = b->out_record_stride;
    }
    a.row_pitch = b->row_pitch > 0 ? b->row_pitch : 0;
    if (a.row_pitch > 0 && (long long)a.row_pitch * (long long)b->n_agents >= (1ll << 31)) { g_last_error = "row_pitch * n_agents exceeds 2^31 rows"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    // Helper wavefronts where an agent has rows beyond the planning wavefront's register cache AND the launch is so small that
    // one wavefront per agent leaves most SIMDs of the chip without work (at most two agents per CU): measured on one MI355X,
    // LP + QP launch of the empty60 mission (60 agents x 1 593 rows) 0.053 -> 0.038 ms; a 2 048-agent crowd, whose agents
    // already fill the CUs their LDS lets them share, loses (0.138 -> 0.19 ms: the builds with helpers run one workgroup per CU)
    static const char *wpa_env = getenv("DLSCQP_WPA");  // tests: "1" = never, "4" = whenever the instantiation has the build
    bool wpa = !ws_hbm && kc.fn_wpa && b->max_rows_per_agent > 64 * kc.KU && a.n_slots_launch <= DQ_WPA_MAX_SLOTS;
    if (wpa_env) wpa = !ws_hbm && kc.fn_wpa && wpa_env[0] != '1';
    bool wide = !wpa && kc.fn_wide && a.n_slots_launch <= DQ_WIDE_MAX_SLOTS;
    // dlscqp_selftest pins the build: the one a launch of thousands of agents gets (two wavefronts per SIMD: the tightest register
    // budget), the WIDE one, the one with helper wavefronts -- whatever the size of its own two-agent launch would pick
    if (c->force_build == 1) { wpa = false; wide = false; }
    else if (c->force_build == 2) { wpa = false; wide = kc.fn_wide != nullptr; }
    else if (c->force_build == 3) { wpa = !ws_hbm && kc.fn_wpa != nullptr; wide = !wpa && wide; }
    if (fused) {  // the fused kernel's plan stage is the build with helper wavefronts (256 threads, like its other stages)
        if (ws_hbm || !kc.fn_wpa) { g_last_error = "fused step: no build with helper wavefronts for this shape"; return DLSCQP_ERR_UNSUPPORTED; }
        wpa = true;
        wide = false;
    }
    kernel_t kfn = wpa ? kc.fn_wpa : (wide ? kc.fn_wide : kc.fn);
    if (!ws_hbm) {
        // Rows beyond the register cache: what the working set leaves of the agent's LDS budget holds them (gi_core.hpp: "rows in
        // LDS"); in the builds with helper wavefronts too (round 5: the helpers price their shares of those rows from LDS).
        // DLSCQP_ROWS_IN_LDS=0: every such row from HBM in every pass, as before round 4 (tests).
        static const bool rows_in_lds = !(getenv("DLSCQP_ROWS_IN_LDS") && getenv("DLSCQP_ROWS_IN_LDS")[0] == '0');
        const bool one_per_simd = wide || (kc.id == 3 && DQ_M10_2D_ONE) || kc.id == 4 || kc.id == 5;
        const int max_per_cu = wpa ? (DQ_WPA_OCC * 4) / DQ_WPA : (one_per_simd ? 4 : 4 * (kc.id == 0 ? DQ_S64_OCC : DQ_PLAN_OCC));
        const LayoutChoice lc = choose_layout(c->kp, b, kc, a.n_slots_launch, max_per_cu < 1 ? 1 : max_per_cu, rows_in_lds,
                                              fused ? LDS_LIMIT - fused->static_lds : LDS_LIMIT);
        lay = lc.lay;
        if (lay.bytes > LDS_LIMIT) { g_last_error = "working set exceeds the LDS of a CU"; return DLSCQP_ERR_UNSUPPORTED; }  // (ws_hbm above rules it out)
        if (lay.spill_doubles > 0) {
            // spill records, one per launch slot: the batch's own workspace if it names one that is large enough, else a buffer of
            // the context (launches of one context that rely on it must not overlap; it never grows inside a stream capture)
            a.spill_stride = ((size_t)lay.spill_doubles + 31) & ~(size_t)31;
            const size_t need = a.spill_stride * 8 * (size_t)(a.slot0 + a.n_slots_launch);
            if (b->workspace && b->workspace_bytes >= (int64_t)need) {
                a.spill = static_cast<double *>(b->workspace);
            } else {
                if (need > c->spill_bytes) {
                    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                    (void)hipStreamIsCapturing(stream, &cap);
                    if (cap != hipStreamCaptureStatusNone) {
                        g_last_error = "the spill records must grow: not inside a stream capture (pass dlscqp_device_batch.workspace, or launch once eagerly first)";
                        return DLSCQP_ERR_INVALID_ARGUMENT;
                    }
                    HIP_TRY(hipDeviceSynchronize());
                    if (c->spill) (void)hipFree(c->spill);
                    c->spill = nullptr;
                    c->spill_bytes = 0;
                    HIP_TRY(hipMalloc(&c->spill, need));
                    c->spill_bytes = need;
                }
                a.spill = static_cast<double *>(c->spill);
            }
        }
    }
    const int attr_slot = kc.id + (wpa ? 2 * (DQ_N_INSTANCES + 1) : (wide ? DQ_N_INSTANCES + 1 : 0));
    if (!fused && !ws_hbm && lay.bytes > c->max_dyn_lds[attr_slot]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lay.bytes));
        c->max_dyn_lds[attr_slot] = lay.bytes;
    }
    if (a.slot0 < 0 || a.slot0 + a.n_slots_launch > 64 * a.n_blocks || (!a.hint[0] && !a.mask_in && a.slot0 + a.n_slots_launch > b->n_agents)) {
        g_last_error = "launch slots out of range";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    a.max_slack_obs = b->max_slack_obs;
    a.do_goal = b->do_goal;
    a.do_traj = b->do_traj;
    a.agents = b->agents;
    a.row_offsets = b->row_offsets;
    a.row_counts = b->row_counts;
    a.rows = RowsView{b->nx, b->ny, b->nz, b->px, b->py, b->pz, b->d, b->key};
    a.sfc = c->kp.use_sfc ? b->sfc_boxes : nullptr;
    a.out_goal = b->out_goal;
    a.out_goal_status = b->out_goal_status;
    a.out_ctrl = b->out_ctrl;
    a.out_x = b->out_ctrl_f64;
    a.out_cost = b->out_cost;
    a.out_slack_cost = b->out_slack_cost;
    a.out_status = b->out_status;
    a.out_iters = b->out_iters;
    a.order = b->agent_order;
    a.out_stamps = (unsigned long long *)b->debug_stamps;
    a.grid = nullptr;
    a.grid_n = 0;
    a.fresh_inputs = 0;
    a.out_conflict = nullptr;
    a.conflict_stride = 0;
    if (b->out_conflict) {
        // the certificates of this launch's INFEASIBLE verdicts (include/dlscqp.h); records of agents without one read "0 members"
        const int need = 4 + 2 * (lay.r_max + 1);
        if (b->conflict_stride < need) { g_last_error = "conflict_stride is smaller than dlscqp_conflict_doubles()"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        if (!b->do_traj) { g_last_error = "out_conflict: the certificate is the trajectory QP's"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        a.out_conflict = b->out_conflict;
        a.conflict_stride = b->conflict_stride;
        HIP_TRY(hipMemsetAsync(b->out_conflict, 0, (size_t)b->n_agents * (size_t)b->conflict_stride * sizeof(double), stream));
    }
    if (fused) {
        fused->id = kc.id;
        fused->kp = c->kp;
        fused->lay = lay;
        fused->a = a;
        return DLSCQP_SUCCESS;
    }
    const bool timed = c->ev_used < (int)c->ev0.size();
    if (timed) HIP_TRY(hipEventRecord(c->ev0[c->ev_used], stream));
#ifdef DLSCQP_TRACE_LDS
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kc.fn), hipFuncAttributeMaxDynamicSharedMemorySize, lay.bytes + 8016));
    hipLaunchKernelGGL(kfn, dim3(a.n_slots_launch), dim3(wpa ? 64 * DQ_WPA : 64), lay.bytes + 8016, stream, c->kp, lay, a);
#else
    // DLSCQP_LDS_TOTAL (developer experiment): the workgroup asks for at least this much LDS -- fewer agents per CU than the registers would allow
    static const int lds_total = getenv("DLSCQP_LDS_TOTAL") ? atoi(getenv("DLSCQP_LDS_TOTAL")) : 0;
    const int lds_bytes = ws_hbm ? 0 : (lds_total > lay.bytes ? lds_total : lay.bytes);
    if (lds_total > lay.bytes && !ws_hbm) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(kfn, dim3(a.n_slots_launch), dim3(wpa ? 64 * DQ_WPA : 64), lds_bytes, stream, c->kp, lay, a);
#endif
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(c->ev1[c->ev_used], stream));
        c->ev_used++;
    }
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_plan_batch_device(dlscqp_ctx *c, const dlscqp_device_batch *b, void *stream) {
    g_last_error.clear();
    if (!c || !b) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (!b->agents || (!b->row_offsets && b->row_pitch <= 0) || !b->out_status) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    (void)hipSetDevice(c->cfg.device);
    // the caller's stream exactly as given: NULL is the HIP null stream (e.g. torch's default stream), not the
    // context's private stream -- launching elsewhere would race with the caller's stream-ordered work
    return launch(c, b, (hipStream_t)stream);
}

// dlscqp_lsc_gen -> the argument block of the half-space and waypoint-trap kernels
static int lsc_gen_args(dlscqp_ctx *c, const dlscqp_lsc_gen *gen, LscGenArgs *out) {
    if (!gen->goals || !gen->radius || !gen->downwash || !gen->nbr_offsets || !gen->nbr_ids || !gen->nx || !gen->d || !gen->key ||
        (gen->first_plan ? (!gen->position || !gen->velocity) : !gen->plans)) {
        g_last_error = "null buffer";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(c->cfg.device);
    LscGenArgs g;
    g.n_local = gen->n_local; g.first_plan = gen->first_plan;
    g.M = c->M; g.n1 = c->n + 1; g.phi = c->phi; g.dt = c->dt;
    g.local_ids = gen->local_ids;
    g.plans = gen->plans; g.position = gen->position; g.velocity = gen->velocity; g.goals = gen->goals;
    g.radius = gen->radius; g.downwash = gen->downwash;
    g.nbr_offsets = gen->nbr_offsets; g.nbr_ids = gen->nbr_ids; g.nbr_counts = gen->nbr_counts;
    g.nx = gen->nx; g.ny = gen->ny; g.nz = gen->nz; g.px = gen->px; g.py = gen->py; g.pz = gen->pz;
    g.d = gen->d; g.key = gen->key;
    g.n_dyn = gen->n_dyn > 0 ? gen->n_dyn : 0;
    if (g.n_dyn > 0 && (!gen->dyn_position || !gen->dyn_velocity || !gen->dyn_radius || !gen->dyn_max_acc || !gen->dyn_downwash)) {
        g_last_error = "null dynamic-obstacle buffer";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    g.dyn_size_prediction = gen->dyn_size_prediction;
    g.dyn_uncertainty_horizon = gen->dyn_uncertainty_horizon;
    g.dyn_position = gen->dyn_position; g.dyn_velocity = gen->dyn_velocity;
    g.dyn_radius = gen->dyn_radius; g.dyn_max_acc = gen->dyn_max_acc; g.dyn_downwash = gen->dyn_downwash;
    g.observed_position = gen->observed_position;
    g.reset_threshold = gen->reset_threshold;
    g.comm_range = c->cfg.communication_range;
    g.use_sfc = c->kp.use_sfc;
    g.prune_sfc = nullptr;
    g.row_counts = nullptr;
    g.tail_from = gen->tail_from;
    g.tail_segments = gen->tail_segments > 0 ? gen->tail_segments : 0;
    if (g.tail_segments > 0 && (!gen->prune || g.tail_segments >= c->M || g.tail_from < 0)) {
        g_last_error = "tail_segments: only with prune, 0 < tail_segments < M, tail_from >= 0";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    if (gen->prune) {
        if (!gen->out_row_counts || (c->kp.use_sfc && !gen->prune_sfc_boxes)) { g_last_error = "prune: null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        if (c->kp.rsfc_mode) {
            // RECIPROCALRSFC bounds the first segment's z by +-100, not by the world box (traj_optimizer.cpp:254-258): the
            // presolve's box is not the model's there
            g_last_error = "prune is not available in RECIPROCALRSFC mode";
            return DLSCQP_ERR_UNSUPPORTED;
        }
        g.prune_sfc = c->kp.use_sfc ? gen->prune_sfc_boxes : nullptr;
        g.row_counts = gen->out_row_counts;
    }
    g.dim = c->dim;
    for (int k = 0; k < 3; k++) { g.world_min[k] = c->cfg.world_min[k]; g.world_max[k] = c->cfg.world_max[k]; }
    *out = g;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_generate_lsc_device(dlscqp_ctx *c, const dlscqp_lsc_gen *gen, void *stream) {
    g_last_error.clear();
    if (!c || !gen) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (gen->n_local <= 0) return DLSCQP_SUCCESS;
    LscGenArgs g;
    const int rc = lsc_gen_args(c, gen, &g);
    if (rc != DLSCQP_SUCCESS) return rc;
    const char *pairs_env = std::getenv("DLSCQP_LSC_PAIRS");  // "0": one wavefront per agent as before (tests compare the two)
    const bool pairs = !(pairs_env && pairs_env[0] == '0');
    if (pairs && g.row_counts && gen->max_neighbours > 0 && (gen->max_neighbours + g.n_dyn) * c->M <= 32)
        hipLaunchKernelGGL(dlscqp_lsc_pair_kernel, dim3((gen->n_local + 1) / 2), dim3(64), 0, (hipStream_t)stream, g);
    else if (gen->max_neighbours > 0 && (gen->max_neighbours + g.n_dyn) * c->M <= 64 && gen->n_local <= 64)
        hipLaunchKernelGGL((dlscqp_lsc_kernel<64, DQ_LSC_OCC_FEW>), dim3(gen->n_local), dim3(64), 0, (hipStream_t)stream, g);
    else if (gen->max_neighbours > 0 && (gen->max_neighbours + g.n_dyn) * c->M <= 64)
        hipLaunchKernelGGL(dlscqp_lsc_kernel<64>, dim3(gen->n_local), dim3(64), 0, (hipStream_t)stream, g);
    else if (gen->n_local <= 64 && gen->max_neighbours > 0 && (gen->max_neighbours + g.n_dyn) * c->M > LSC_THREADS)
        // a handful of agents with more items than one round of 128 threads takes (the forest mission: 13 x 10): the chip is empty
        // anyway, a workgroup of 256 or 512 threads takes them in one round or two
    {
        if ((gen->max_neighbours + g.n_dyn) * c->M > 256)  // (empty60: 59 x 5)
            hipLaunchKernelGGL((dlscqp_lsc_kernel<512, DQ_LSC_OCC_FEW>), dim3(gen->n_local), dim3(512), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL((dlscqp_lsc_kernel<256, DQ_LSC_OCC_FEW>), dim3(gen->n_local), dim3(256), 0, (hipStream_t)stream, g);
    } else if (gen->n_local <= 64)
        hipLaunchKernelGGL((dlscqp_lsc_kernel<LSC_THREADS, DQ_LSC_OCC_FEW>), dim3(gen->n_local), dim3(LSC_THREADS), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(dlscqp_lsc_kernel<LSC_THREADS>, dim3(gen->n_local), dim3(LSC_THREADS), 0, (hipStream_t)stream, g);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_waypoint_trap_device(dlscqp_ctx *c, const dlscqp_lsc_gen *gen, const dlscqp_waypoint_trap *trap, void *stream) {
    g_last_error.clear();
    if (!c || !gen || !trap) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (gen->n_local <= 0 || gen->n_dyn <= 0) return DLSCQP_SUCCESS;  // nothing to release without dynamic obstacles (:709)
    if (!trap->agents || (c->kp.use_sfc && !trap->sfc_boxes)) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    TrapArgs t;
    const int rc = lsc_gen_args(c, gen, &t.g);
    if (rc != DLSCQP_SUCCESS) return rc;
    t.agents = trap->agents;
    t.sfc = c->kp.use_sfc ? trap->sfc_boxes : nullptr;
    t.out_released = trap->out_released;
    t.horizon = c->M * c->dt;
    hipLaunchKernelGGL(dlscqp_trap_kernel, dim3(gen->n_local), dim3(64), 0, (hipStream_t)stream, t);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}


// ---------------------------------------------------------------------------------------------
// communication-range neighbour lists (include/dlscqp.h: dlscqp_neighbours_device)
// ---------------------------------------------------------------------------------------------
namespace {
struct NbrGrid {
    float lo[3], inv_cell[3];
    int n[3];
};
struct NbrArgs {
    NbrGrid grid;
    dlscqp_neighbour_search ns;
    double range;
    int per;
    int32_t *cell_count, *cell_start, *agent_cell, *agent_slot, *by_cell;
    int n_cells;
    int direct;  // 1: few agents -- every agent is a candidate, no cells (the gather kernel alone)
};

__device__ __forceinline__ int nbr_cell_of(const NbrGrid &g, const float *p, int *c3) {
    int id = 0;
#pragma unroll
    for (int k = 2; k >= 0; k--) {
        int c = (int)floorf((p[k] - g.lo[k]) * g.inv_cell[k]);
        c = c < 0 ? 0 : (c >= g.n[k] ? g.n[k] - 1 : c);  // monotone clamp: agents within range stay in adjacent cells
        c3[k] = c;
        id = id * g.n[k] + c;
    }
    return id;
}

// 1: every agent takes a slot in its cell
__global__ __launch_bounds__(256) void dlscqp_nbr_bin_kernel(NbrArgs a) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= a.ns.n_total) return;
    int c3[3];
    const int cell = nbr_cell_of(a.grid, a.ns.position + (size_t)i * a.ns.position_stride, c3);
    a.agent_cell[i] = cell;
    a.agent_slot[i] = atomicAdd(&a.cell_count[cell], 1);
}

// 2: exclusive prefix sum of the cell counts (one workgroup; a few hundred to a few thousand cells)
__global__ __launch_bounds__(1024) void dlscqp_nbr_scan_kernel(NbrArgs a) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < a.n_cells; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int v = i < a.n_cells ? a.cell_count[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            const int t = threadIdx.x >= (unsigned)s ? part[threadIdx.x - s] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < a.n_cells) a.cell_start[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.cell_start[a.n_cells] = carry;
}

// 3: agents ordered by cell (order within a cell: as the slots were drawn -- the lists are sorted afterwards)
__global__ __launch_bounds__(256) void dlscqp_nbr_scatter_kernel(NbrArgs a) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= a.ns.n_total) return;
    a.by_cell[a.cell_start[a.agent_cell[i]] + a.agent_slot[i]] = i;
}

// 4: one wavefront per local agent: the members of the 27 cells around its own (nine runs of up to three cells that are
// contiguous in by_cell), LInfinityDistance test, compaction into LDS, rank sort by id, the first `pitch` ids out
constexpr int NBR_LIST_CAP = 4096;
constexpr int NBR_DIRECT_MAX = 256;  // (DLSCQP_NBR_DIRECT_MAX in the environment: tests compare the two ways on the same agents)
// (one wavefront does the work -- threads tid < 64 of the workgroup; the others only keep the barrier company: dlscqp_fused_kernel
// calls this with 256 threads)
__device__ __forceinline__ void nbr_gather_agent(const NbrArgs &a, const int k, const int tid) {
    __shared__ int list[NBR_LIST_CAP];
    const int lane = tid & 63;
    const bool act = tid < 64;
    if (k >= a.ns.n_local) return;  // (the whole workgroup)
    const int self = a.ns.local_ids ? a.ns.local_ids[k] : k;
    const float *ps = a.ns.position + (size_t)self * a.ns.position_stride;
    const float sx = ps[0], sy = ps[1], sz = ps[2];
    int count = 0;
    auto consider = [&](int lo, int hi) {  // candidates by_cell[lo .. hi), uniform bounds
        for (int b = lo; b < hi; b += 64) {
            const int j = b + lane;
            const int id = j < hi ? (a.direct ? j : a.by_cell[j]) : self;
            const float *pq = a.ns.position + (size_t)id * a.ns.position_stride;
            // util.hpp:131-140: float32 differences, the largest of their absolute values, compared in double
            float d = fabsf(sx - pq[0]);
            d = fmaxf(d, fabsf(sy - pq[1]));
            d = fmaxf(d, fabsf(sz - pq[2]));
            const bool in = id != self && !(a.range > 0 && (double)d > a.range);
            const unsigned long long m = __ballot(in);
            const int at = count + __popcll(m & ((1ull << lane) - 1ull));
            if (in && at < NBR_LIST_CAP) list[at] = id;
            count += __popcll(m);
        }
    };
    if (!act) {
        // (wavefronts beyond the first: nothing to gather)
    } else if (a.range > 0 && !a.direct) {
        int c3[3];
        nbr_cell_of(a.grid, ps, c3);
        for (int dz = -1; dz <= 1; dz++) {
            const int z = c3[2] + dz;
            if (z < 0 || z >= a.grid.n[2]) continue;
            for (int dy = -1; dy <= 1; dy++) {
                const int y = c3[1] + dy;
                if (y < 0 || y >= a.grid.n[1]) continue;
                const int x0 = c3[0] > 0 ? c3[0] - 1 : 0, x1 = c3[0] + 1 < a.grid.n[0] ? c3[0] + 1 : a.grid.n[0] - 1;
                const int row = (z * a.grid.n[1] + y) * a.grid.n[0];
                consider(a.cell_start[row + x0], a.cell_start[row + x1 + 1]);
            }
        }
    } else {
        consider(0, a.ns.n_total);  // no communication limit: by_cell holds every agent
    }
    // the cell counts back to zero for the next search (their last reader, the prefix sum, ran before this kernel): the search
    // has no memset of its own -- a memset node inside a captured step was seen to leave the counts standing on the second
    // replay of the graph (round 4's memory access fault: slots beyond a cell's run, a scatter out of bounds)
    if (!a.direct && act)
        for (int e = k * 64 + lane; e <= a.n_cells; e += a.ns.n_local * 64) a.cell_count[e] = 0;
    __syncthreads();
    if (!act) return;
    const int kept = count < NBR_LIST_CAP ? count : NBR_LIST_CAP;
    for (int e = lane; e < kept; e += 64) {
        const int id = list[e];
        int rank = 0;
        for (int j = 0; j < kept; j++) rank += list[j] < id ? 1 : 0;
        if (rank < a.ns.pitch) a.ns.out_nbr_ids[(size_t)k * a.ns.pitch + rank] = id;
    }
    if (lane == 0) {
        const int n = kept < a.ns.pitch ? kept : a.ns.pitch;
        a.ns.out_nbr_counts[k] = n;
        if (a.ns.out_row_counts) a.ns.out_row_counts[k] = (n + a.ns.n_dyn) * a.per;
        if (a.ns.out_overflow && (count > a.ns.pitch || count > NBR_LIST_CAP)) *a.ns.out_overflow = 1;
    }
}
__global__ __launch_bounds__(64) void dlscqp_nbr_gather_kernel(NbrArgs a) { nbr_gather_agent(a, (int)blockIdx.x, (int)threadIdx.x); }
}  // namespace

// dlscqp_neighbour_search -> the kernels' argument block (the cell scratch pointers are the caller's to fill in)
static int nbr_build_args(dlscqp_ctx *c, const dlscqp_neighbour_search *ns, NbrArgs &a) {
    if (!ns->position || !ns->out_nbr_ids || !ns->out_nbr_counts || ns->pitch <= 0 || ns->position_stride < 3 || ns->n_dyn < 0) {
        g_last_error = "null buffer or invalid size";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    a.ns = *ns;
    a.range = c->cfg.communication_range;
    a.per = c->M * (c->n + 1) - c->phi;
    // cells of one communication range (never more than 64 per axis: larger cells keep the 27-cell search valid)
    a.n_cells = 1;
    for (int k = 0; k < 3; k++) {
        const double ext = std::max(0.0, c->cfg.world_max[k] - c->cfg.world_min[k]);
        double cell = a.range > 0 ? a.range : std::max(ext, 1.0);
        if (ext / cell > 64.0) cell = ext / 64.0;
        a.grid.lo[k] = (float)c->cfg.world_min[k];
        // cells 1e-3 larger than the range: the cell index is computed in float32 with up to 64 cells per axis (rounding error up
        // to ~4e-6 cell units), so the margin has to be well above that for two agents within range never to land two cells apart
        // (the exact distance test filters afterwards: larger cells cost nothing in correctness)
        a.grid.inv_cell[k] = (float)(1.0 / (cell * (1.0 + 1e-3)));
        a.grid.n[k] = a.range > 0 ? std::max(1, (int)std::floor(ext / cell) + 1) : 1;
        a.n_cells *= a.grid.n[k];
    }
    // Few agents (every mission of the reference): the cells cost four launches to spare each agent a look at a handful of others.
    // The candidates are everybody, the test and the order of the list are the same: the same lists (tests/test_gpu_neighbours.py).
    const char *direct_env = std::getenv("DLSCQP_NBR_DIRECT_MAX");
    const int direct_max = direct_env ? std::atoi(direct_env) : NBR_DIRECT_MAX;
    a.direct = ns->n_total <= direct_max ? 1 : 0;
    a.cell_count = a.cell_start = a.agent_cell = a.agent_slot = a.by_cell = nullptr;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_neighbours_device(dlscqp_ctx *c, const dlscqp_neighbour_search *ns, void *stream) {
    g_last_error.clear();
    if (!c || !ns) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (ns->n_local <= 0 || ns->n_total <= 0) return DLSCQP_SUCCESS;
    NbrArgs a;
    {
        const int rc = nbr_build_args(c, ns, a);
        if (rc != DLSCQP_SUCCESS) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    if (a.direct) {
        a.cell_count = a.cell_start = a.agent_cell = a.agent_slot = a.by_cell = nullptr;
        hipLaunchKernelGGL(dlscqp_nbr_gather_kernel, dim3(ns->n_local), dim3(64), 0, st, a);
        HIP_TRY(hipGetLastError());
        return DLSCQP_SUCCESS;
    }
    const size_t need = (size_t)2 * (a.n_cells + 1) + (size_t)3 * ns->n_total + 16;
    if (need > c->nbr_scratch_ints) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) { g_last_error = "the neighbour search's scratch must grow: not inside a stream capture (search once eagerly first)"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        HIP_TRY(hipDeviceSynchronize());
        if (c->nbr_scratch) (void)hipFree(c->nbr_scratch);
        c->nbr_scratch = nullptr;
        c->nbr_scratch_ints = 0;
        HIP_TRY(hipMalloc((void **)&c->nbr_scratch, need * sizeof(int32_t)));
        HIP_TRY(hipMemset(c->nbr_scratch, 0, need * sizeof(int32_t)));  // (the cell counts start at zero; every search leaves them so)
        c->nbr_scratch_ints = need;
        c->nbr_cells_clean = a.n_cells;
    }
    // The counts are zero only where the last complete search of THIS grid left them so: another number of cells (dlscqp_update_param
    // with another communication range or world box) lays the scratch out differently -- what were cell starts or agents' cells are
    // counts now -- and a search that failed between its kernels left them dirty.  Then they are cleared here, on the stream.
    if (c->nbr_cells_clean != a.n_cells) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) { g_last_error = "the neighbour search's cell counts must be cleared for a new grid: not inside a stream capture (search once eagerly first)"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        HIP_TRY(hipMemsetAsync(c->nbr_scratch, 0, (size_t)(a.n_cells + 1) * sizeof(int32_t), st));
    }
    c->nbr_cells_clean = -1;  // (until all four kernels are queued)
    a.cell_count = c->nbr_scratch;
    a.cell_start = a.cell_count + a.n_cells + 1;
    a.agent_cell = a.cell_start + a.n_cells + 1;
    a.agent_slot = a.agent_cell + ns->n_total;
    a.by_cell = a.agent_slot + ns->n_total;
    const int nb = (ns->n_total + 255) / 256;  // (cell_count is zero: the gather kernel of the search before left it so)
    hipLaunchKernelGGL(dlscqp_nbr_bin_kernel, dim3(nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(dlscqp_nbr_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(dlscqp_nbr_scatter_kernel, dim3(nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(dlscqp_nbr_gather_kernel, dim3(ns->n_local), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    c->nbr_cells_clean = a.n_cells;  // (the gather kernel clears the counts of this grid behind itself)
    return DLSCQP_SUCCESS;
}

// the k nearest agents of an agent's own block (include/dlscqp.h: dlscqp_neighbours_knn_device): one wavefront per agent, k rounds
// of "the nearest candidate beyond the one taken last" -- the candidates of a block in trips of 64, keys (distance^2, id)
namespace {
__global__ __launch_bounds__(64) void dlscqp_nbr_knn_kernel(dlscqp_neighbour_knn a) {
    const int self = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (self >= a.n_total) return;
    const int g0 = (self / a.group) * a.group, g1 = g0 + a.group < a.n_total ? g0 + a.group : a.n_total;
    const float *ps = a.position + (size_t)self * a.position_stride;
    const double sx = ps[0], sy = ps[1], sz = ps[2];
    const int want = a.k < g1 - g0 - 1 ? a.k : g1 - g0 - 1;
    double last_d = -1.0;
    int last_id = -1;
    for (int r = 0; r < want; r++) {  // uniform
        double best_d = 1.7976931348623157e308;
        int best_id = 0x7fffffff;
        for (int c0 = g0; c0 < g1; c0 += 64) {
            const int id = c0 + lane;
            const int ic = id < g1 ? id : g1 - 1;
            const float *pq = a.position + (size_t)ic * a.position_stride;
            const double dx = (double)pq[0] - sx, dy = (double)pq[1] - sy, dz = (double)pq[2] - sz;
            const double d = dx * dx + dy * dy + dz * dz;
            const bool after_last = d > last_d || (d == last_d && ic > last_id);
            const bool cand = id < g1 && ic != self && after_last;
            const bool better = cand && (d < best_d || (d == best_d && ic < best_id));
            best_d = better ? d : best_d;
            best_id = better ? ic : best_id;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {  // wave minimum of (distance, id)
            const double od = __shfl_xor(best_d, s, 64);
            const int oi = __shfl_xor(best_id, s, 64);
            const bool take = od < best_d || (od == best_d && oi < best_id);
            best_d = take ? od : best_d;
            best_id = take ? oi : best_id;
        }
        last_d = best_d;
        last_id = best_id;
        if (lane == 0) a.out_nbr_ids[(size_t)self * a.pitch + r] = best_id;
    }
    if (lane == 0) a.out_nbr_counts[self] = want;
}
}  // namespace

extern "C" int dlscqp_neighbours_knn_device(dlscqp_ctx *c, const dlscqp_neighbour_knn *nk, void *stream) {
    g_last_error.clear();
    if (!c || !nk) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (nk->n_total <= 0) return DLSCQP_SUCCESS;
    if (!nk->position || !nk->out_nbr_ids || !nk->out_nbr_counts || nk->group < 1 || nk->k < 0 || nk->k > nk->pitch || nk->position_stride < 3) {
        g_last_error = "null buffer or invalid size";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    hipLaunchKernelGGL(dlscqp_nbr_knn_kernel, dim3(nk->n_total), dim3(64), 0, (hipStream_t)stream, *nk);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// launch order of the next step (include/dlscqp.h: dlscqp_launch_order_device)
// ---------------------------------------------------------------------------------------------
namespace {
#ifndef DQ_ORDER_THREADS
#define DQ_ORDER_THREADS 1024
#endif
constexpr int ORDER_THREADS = DQ_ORDER_THREADS, ORDER_BINS = 64;
static_assert(ORDER_THREADS > 2 * ORDER_BINS, "the scatter kernel's last workgroup zeroes the scratch block with one thread per word");

// Counting sort of the agents by the iteration count of the step just planned, descending (bin 0 = 63 iterations or
// more), one thread per agent in two passes.  A wavefront handles the bins present among its 64 agents one after the
// other (usually one or two): one atomic per bin and wavefront instead of one per agent.
template <class F>
__device__ __forceinline__ void order_bins_of_wave(int bin, bool valid, F f) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    while (todo) {  // uniform
        const int leader = __builtin_ctzll(todo);
        const int lb = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(valid && bin == lb);
        f(lb, same, leader, valid && bin == lb, __popcll(same & ((1ull << lane) - 1ull)));
        todo &= ~same;
    }
}

// The key of an agent: the iteration count of the step just planned (out_iters: iterations, adds, drops, active rows at
// exit) -- or, with the counts of up to three earlier steps as well, their maximum with decay, max_k (3/4)^k iterations_k: an
// agent that was hard two or three steps ago is likely to be hard again, and one long agent that starts late is what a
// launch waits for.
struct OrderHist {
    const int32_t *it[3];  // newest first; null = not given
};
__device__ __forceinline__ int order_bin(const OrderHist &h, int idx) {
    int it = h.it[0][(size_t)idx * 4];
    if (h.it[1]) { const int v = (3 * h.it[1][(size_t)idx * 4]) / 4; it = v > it ? v : it; }
    if (h.it[2]) { const int v = (9 * h.it[2][(size_t)idx * 4]) / 16; it = v > it ? v : it; }
    return ORDER_BINS - 1 - (it < 0 ? 0 : (it > ORDER_BINS - 1 ? ORDER_BINS - 1 : it));
}

// Pass 1: histogram.  Wavefronts add to the workgroup's LDS histogram, the workgroup adds its non-empty bins to the global
// one: a few dozen global atomics on the crowded bin (agents that never iterate) instead of one per wavefront -- contended
// device-scope atomics on one address cost ~100 ns each.  `scratch` = 64 counts + 64 cursors + a ticket, zero when a
// call starts: the scatter kernel's last workgroup zeroes them again (a replayed graph finds them as the capture did).
__global__ __launch_bounds__(ORDER_THREADS) void dlscqp_order_count_kernel(OrderHist iters, int n, int *scratch) {
    __shared__ int hist[ORDER_BINS];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < ORDER_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    const int idx = (int)(blockIdx.x * ORDER_THREADS + threadIdx.x);
    const bool valid = idx < n;
    order_bins_of_wave(valid ? order_bin(iters, idx) : -1, valid, [&](int lb, unsigned long long same, int leader, bool, int) {
        if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
    });
    __syncthreads();
    if (threadIdx.x < ORDER_BINS && hist[threadIdx.x] > 0) atomicAdd(&scratch[threadIdx.x], hist[threadIdx.x]);
}

// Pass 2: slot of an agent = first slot of its bin (exclusive prefix sum over the global histogram) + the range the
// workgroup claims from the bin's global cursor + the range the wavefront claimed inside the workgroup + the agent's rank
// among the wavefront's agents of that bin.
__global__ __launch_bounds__(ORDER_THREADS) void dlscqp_order_scatter_kernel(OrderHist iters, int n, int *scratch, int32_t *order) {
    __shared__ int local[ORDER_BINS], first[ORDER_BINS];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < ORDER_BINS) local[threadIdx.x] = 0;
    __syncthreads();
    const int idx = (int)(blockIdx.x * ORDER_THREADS + threadIdx.x);
    const bool valid = idx < n;
    const int bin = valid ? order_bin(iters, idx) : 0;
    int within = 0;  // the agent's position among the workgroup's agents of its bin
    order_bins_of_wave(valid ? bin : -1, valid, [&](int lb, unsigned long long same, int leader, bool mine, int rank) {
        int start = 0;
        if (lane == leader) start = atomicAdd(&local[lb], __popcll(same));
        start = __shfl(start, leader, 64);
        within = mine ? start + rank : within;
    });
    __syncthreads();
    if (threadIdx.x < 64) {
        const int h = scratch[lane];
        int incl = h;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int v = __shfl_up(incl, s, 64);
            incl += lane >= s ? v : 0;
        }
        const int mine = local[lane];
        first[lane] = incl - h + (mine > 0 ? atomicAdd(&scratch[ORDER_BINS + lane], mine) : 0);
    }
    __syncthreads();
    const int at = first[bin] + within;
    if (valid && at >= 0 && at < n) order[at] = idx;  // in range by construction; a corrupted scratch block must not write elsewhere
    // the last workgroup to get here leaves the scratch block zeroed for the next call (a memset on the stream before every
    // call was a third stream operation: with several groups of swarms sorting side by side each one waits for wave slots)
    __shared__ int last;
    __syncthreads();  // this workgroup has read the histogram and claimed its ranges
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(&scratch[2 * ORDER_BINS], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x <= 2 * ORDER_BINS) scratch[threadIdx.x] = 0;
}
}  // namespace

static int order_launch(dlscqp_ctx *c, const OrderHist &h, int32_t n_agents, int32_t *order_out, void *stream);

extern "C" int dlscqp_launch_order_device(dlscqp_ctx *c, const int32_t *iters, int32_t n_agents, int32_t *order_out, void *stream) {
    return order_launch(c, OrderHist{{iters, nullptr, nullptr}}, n_agents, order_out, stream);
}

extern "C" int dlscqp_launch_order_history_device(dlscqp_ctx *c, const int32_t *const *iters, int32_t n_steps, int32_t n_agents,
                                                  int32_t *order_out, void *stream) {
    if (!iters || n_steps < 1 || n_steps > 3) { g_last_error = "one to three steps of iteration counts"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    OrderHist h{{iters[0], n_steps > 1 ? iters[1] : nullptr, n_steps > 2 ? iters[2] : nullptr}};
    for (int k = 0; k < n_steps; k++)
        if (!iters[k] && n_agents > 0) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    return order_launch(c, h, n_agents, order_out, stream);
}

static int order_launch(dlscqp_ctx *c, const OrderHist &iters, int32_t n_agents, int32_t *order_out, void *stream) {
    g_last_error.clear();
    if (!c || n_agents < 0) { g_last_error = "invalid argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (n_agents == 0) return DLSCQP_SUCCESS;
    if (!iters.it[0] || !order_out) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->d_order_scratch) {  // 64 counts, 64 cursors, a ticket: zero now, and every call leaves them zero (dlscqp_order_scatter_kernel)
        HIP_TRY(hipMalloc((void **)&c->d_order_scratch, (2 * ORDER_BINS + 1) * sizeof(int)));
        HIP_TRY(hipMemset(c->d_order_scratch, 0, (2 * ORDER_BINS + 1) * sizeof(int)));
    }
    int *cur = c->d_order_scratch;  // one block, used by the calls of a context in stream order (they must not overlap)
    const int grid = (n_agents + ORDER_THREADS - 1) / ORDER_THREADS;
    hipLaunchKernelGGL(dlscqp_order_count_kernel, dim3(grid), dim3(ORDER_THREADS), 0, (hipStream_t)stream, iters, (int)n_agents, cur);
    hipLaunchKernelGGL(dlscqp_order_scatter_kernel, dim3(grid), dim3(ORDER_THREADS), 0, (hipStream_t)stream, iters, (int)n_agents, cur, order_out);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

// What hinted_agent() answers for every launch slot, written out: the permutation a launch with these hint_iters uses (tests,
// diagnostics).  order_out[64 * n_blocks]: the agent of slot s, or -1 for a slot beyond its block's agents.
namespace {
__global__ __launch_bounds__(64) void dlscqp_hint_preview_kernel(KernelArgs a, int32_t *order_out) {
    const int slot = (int)blockIdx.x;
    const int ag = hinted_agent(a, slot, (int)threadIdx.x);
    if (threadIdx.x == 0) order_out[slot] = ag;
}
}  // namespace

namespace {
__global__ __launch_bounds__(64) void dlscqp_mask_preview_kernel(KernelArgs a, int32_t *order_out) {
    const int slot = (int)blockIdx.x;
    int cls_in;
    const int ag = masked_agent(a.mask_in, a.n_blocks, a.blocks_magic, a.n_agents, slot, (int)threadIdx.x, cls_in);
    if (threadIdx.x == 0) order_out[slot] = ag;
}
}  // namespace

extern "C" int dlscqp_mask_order_preview_device(dlscqp_ctx *c, const uint64_t *masks, int32_t n_agents, int32_t *order_out, void *stream) {
    g_last_error.clear();
    if (!c || !masks || n_agents < 0 || (n_agents > 0 && !order_out)) { g_last_error = "invalid argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (n_agents == 0) return DLSCQP_SUCCESS;
    HIP_TRY(hipSetDevice(c->cfg.device));
    KernelArgs a;
    std::memset(&a, 0, sizeof a);
    a.n_agents = n_agents;
    a.mask_in = reinterpret_cast<const unsigned long long *>(masks);
    a.n_blocks = (n_agents + 63) / 64;
    const int slots = 64 * a.n_blocks;
    if (a.n_blocks > 1 && (unsigned long long)slots * (unsigned long long)a.n_blocks < (1ull << 32))
        a.blocks_magic = (unsigned int)(((1ull << 32) + (unsigned long long)a.n_blocks - 1) / (unsigned long long)a.n_blocks);
    hipLaunchKernelGGL(dlscqp_mask_preview_kernel, dim3(slots), dim3(64), 0, (hipStream_t)stream, a, order_out);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_hint_order_preview_device(dlscqp_ctx *c, const int32_t *const *iters, int32_t n_steps, int32_t n_agents, int32_t *order_out,
                                                void *stream) {
    g_last_error.clear();
    if (!c || !iters || n_steps < 1 || n_steps > 3 || n_agents < 0 || (n_agents > 0 && !order_out)) { g_last_error = "invalid argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (n_agents == 0) return DLSCQP_SUCCESS;
    HIP_TRY(hipSetDevice(c->cfg.device));
    KernelArgs a;
    std::memset(&a, 0, sizeof a);
    a.n_agents = n_agents;
    for (int k = 0; k < n_steps; k++) {
        if (!iters[k]) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        a.hint[k] = iters[k];
    }
    a.n_blocks = (n_agents + 63) / 64;
    const int slots = 64 * a.n_blocks;
    if (a.n_blocks > 1 && (unsigned long long)slots * (unsigned long long)a.n_blocks < (1ull << 32))
        a.blocks_magic = (unsigned int)(((1ull << 32) + (unsigned long long)a.n_blocks - 1) / (unsigned long long)a.n_blocks);
    hipLaunchKernelGGL(dlscqp_hint_preview_kernel, dim3(slots), dim3(64), 0, (hipStream_t)stream, a, order_out);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// between two launches (include/dlscqp.h: dlscqp_advance)
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void dlscqp_advance_kernel(AdvanceArgs g) {
    __shared__ float stage[512];  // advance_args: a plan has at most 512 floats
    advance_agent(g, (int)blockIdx.x, (int)threadIdx.x, stage);
}
}  // namespace

extern "C" int dlscqp_advance_device(dlscqp_ctx *c, const dlscqp_advance *adv, void *stream) {
    g_last_error.clear();
    if (!c || !adv) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (adv->n_agents <= 0) return DLSCQP_SUCCESS;
    AdvanceArgs g;
    const int rc = advance_args(c, adv, g);
    if (rc != DLSCQP_SUCCESS) return rc;
    (void)hipSetDevice(c->cfg.device);
    hipLaunchKernelGGL(dlscqp_advance_kernel, dim3(adv->n_agents), dim3(64), 0, (hipStream_t)stream, g);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

struct dlscqp_world {
    int device;
    SfcWorldView view;  // device pointers
    void *d_near;
    void *d_gaps;
    // The summed-volume tables (SfcWorldView::svt) are built when the first launch names its agents: which tables are needed
    // depends on the agents' radii, and those live in device memory.  Host copy of the lattice-point table to build them from;
    // svt_key = the (radius pointer, count) the tables were last checked against.  A launch whose agents need a table that is
    // not there (radii changed behind the same pointer) still runs: those agents test their boxes point by point.
    mutable SfcWorldHost host;       // only res, lo, n, gaps kept
    mutable void *d_svt_ptrs = nullptr;  // device [15] pointers
    mutable void *d_svt[15] = {};
    mutable const void *svt_key_ptr = nullptr;
    mutable int svt_key_n = 0;
    mutable bool prepared = false;  // dlscqp_world_prepare named the radii: launches no longer look at theirs
};

namespace {
// Build the tables the agents of this launch need and do not have.  Synchronises with `stream` (the radii may be the result of
// work queued on it) -- once per (radius pointer, count); never during a stream capture.
// the tables of the gap classes of `radius[0..n)` that are not there yet (host radii).  Bounded: a table of more than 1 GiB, or
// one that would take the world's tables beyond SVT_TOTAL_LIMIT bytes, is not built (its agents test their boxes point by point);
// a table whose upload fails is freed again and the pointer table on the device is left as it was.
constexpr size_t SVT_TOTAL_LIMIT = (size_t)4 << 30;
int world_build_svt(const dlscqp_world *w, const double *radius, int n_agents) {
    const size_t entries = ((size_t)w->host.n[0] + 2) * ((size_t)w->host.n[1] + 2) * ((size_t)w->host.n[2] + 2);
    size_t have = 0;
    for (void *t : w->d_svt)
        if (t) have += entries * 4;
    bool added = false;
    for (int i = 0; i < n_agents; i++) {
        const int g = sfc_gap_max(radius[(size_t)i], w->host.res);
        if (g < 0 || g > 14 || w->d_svt[g] || entries * 4 > ((size_t)1 << 30) || have + entries * 4 > SVT_TOTAL_LIMIT) continue;
        std::vector<int32_t> svt;
        build_sfc_svt(w->host, g, svt);
        void *t = nullptr;
        hipError_t e = hipMalloc(&t, entries * 4);
        if (e == hipSuccess && (e = hipMemcpy(t, svt.data(), entries * 4, hipMemcpyHostToDevice)) != hipSuccess) (void)hipFree(t);
        if (e != hipSuccess) {
            g_last_error = std::string("summed-volume table: ") + hipGetErrorString(e);
            return DLSCQP_ERR_HIP;
        }
        w->d_svt[g] = t;
        have += entries * 4;
        added = true;
    }
    if (added) HIP_TRY(hipMemcpy(w->d_svt_ptrs, w->d_svt, sizeof(w->d_svt), hipMemcpyHostToDevice));
    return DLSCQP_SUCCESS;
}

int world_prepare_svt(const dlscqp_world *w, const double *d_radius, int n_agents, hipStream_t stream) {
    if (!w->view.svt || w->prepared || (w->svt_key_ptr == (const void *)d_radius && w->svt_key_n == n_agents)) return DLSCQP_SUCCESS;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return DLSCQP_SUCCESS;
    std::vector<double> radius((size_t)n_agents);
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(radius.data(), d_radius, sizeof(double) * (size_t)n_agents, hipMemcpyDeviceToHost));
    const int rc = world_build_svt(w, radius.data(), n_agents);
    if (rc != DLSCQP_SUCCESS) return rc;
    w->svt_key_ptr = d_radius;
    w->svt_key_n = n_agents;
    return DLSCQP_SUCCESS;
}
}  // namespace

extern "C" int dlscqp_world_create(dlscqp_ctx *c, const double *boxes, int32_t n_boxes, double resolution,
                                   const float world_min[3], const float world_max[3], dlscqp_world **out) {
    g_last_error.clear();
    if (!c || !out || (n_boxes > 0 && !boxes) || !(resolution > 0) || !world_min || !world_max) {
        g_last_error = "null or invalid argument";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(c->cfg.device);
    SfcWorldHost hw;
    build_sfc_world(boxes, n_boxes, resolution, world_min, world_max, hw);
    dlscqp_world *w = new dlscqp_world();
    w->device = c->cfg.device;
    w->d_near = nullptr;
    w->d_gaps = nullptr;
    const size_t nc = hw.cells();
    auto fail = [&](hipError_t e, const char *what) {
        g_last_error = std::string(what) + ": " + hipGetErrorString(e);
        dlscqp_world_destroy(w);
        return DLSCQP_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&w->d_near, nc * 32)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemcpy(w->d_near, hw.near_packed.data(), nc * 32, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    w->view.res = hw.res;
    for (int k = 0; k < 3; k++) {
        w->view.wmin[k] = hw.wmin[k]; w->view.wmax[k] = hw.wmax[k];
        w->view.lo[k] = hw.lo[k]; w->view.n[k] = hw.n[k];
    }
    w->view.near = (const uint32_t *)w->d_near;
    // the lattice-point table (one word per cell): DLSCQP_SFC_RECORDS=1 keeps the kernel on the 32-byte records (tests compare the two)
    w->view.gaps = nullptr;
    if (!std::getenv("DLSCQP_SFC_RECORDS") && nc < (size_t)SFC_GAPS_MAX_CELLS) {
        if ((e = hipMalloc(&w->d_gaps, nc * 4)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMemcpy(w->d_gaps, hw.gaps.data(), nc * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
        w->view.gaps = (const uint32_t *)w->d_gaps;
    }
    // DLSCQP_SFC_SVT=0: every box test walks its lattice points (tests compare the two)
    w->view.svt = nullptr;
    const char *svt_env = std::getenv("DLSCQP_SFC_SVT");
    if (w->view.gaps && !(svt_env && svt_env[0] == '0')) {
        if ((e = hipMalloc(&w->d_svt_ptrs, sizeof(w->d_svt))) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMemset(w->d_svt_ptrs, 0, sizeof(w->d_svt))) != hipSuccess) return fail(e, "hipMemset");
        w->view.svt = (const int32_t *const *)w->d_svt_ptrs;
        w->host.res = hw.res;
        for (int k = 0; k < 3; k++) { w->host.lo[k] = hw.lo[k]; w->host.n[k] = hw.n[k]; }
        w->host.gaps.swap(hw.gaps);
    }
    *out = w;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_world_prepare(dlscqp_world *w, const double *radius, int32_t n_agents) {
    g_last_error.clear();
    if (!w || (n_agents > 0 && !radius) || n_agents < 0) { g_last_error = "null or invalid argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    (void)hipSetDevice(w->device);
    if (w->view.svt) {
        HIP_TRY(hipDeviceSynchronize());  // (a launch that reads the pointer table may be running)
        const int rc = world_build_svt(w, radius, n_agents);
        if (rc != DLSCQP_SUCCESS) return rc;
    }
    w->prepared = true;
    return DLSCQP_SUCCESS;
}

extern "C" void dlscqp_world_destroy(dlscqp_world *w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->d_near) (void)hipFree(w->d_near);
    if (w->d_gaps) (void)hipFree(w->d_gaps);
    if (w->d_svt_ptrs) (void)hipFree(w->d_svt_ptrs);
    for (void *t : w->d_svt)
        if (t) (void)hipFree(t);
    delete w;
}

// dlscqp_sfc_gen -> the corridor kernel's argument block (the world's summed-volume tables are made ready on `stream`)
static int sfc_build_args(dlscqp_ctx *c, const dlscqp_world *world, const dlscqp_sfc_gen *gen, void *stream, SfcGenArgs &a) {
    if (!gen->radius || !gen->max_vel || !gen->sfcs ||
        (gen->initialize ? !gen->position : (!gen->initial_traj || !gen->goals || !gen->waypoints))) {
        g_last_error = "null buffer";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(c->cfg.device);
    {
        const int rc = world_prepare_svt(world, gen->radius, gen->n_agents, (hipStream_t)stream);
        if (rc != DLSCQP_SUCCESS) return rc;
    }
    a.world = world->view;
    a.n_agents = gen->n_agents; a.initialize = gen->initialize;
    a.M = c->M; a.n1 = c->n + 1;
    a.grid_resolution = gen->grid_resolution; a.dt = c->dt;
    a.radius = gen->radius; a.max_vel = gen->max_vel;
    a.position = gen->position; a.initial_traj = gen->initial_traj; a.goals = gen->goals; a.waypoints = gen->waypoints;
    a.sfcs = gen->sfcs; a.out_code = gen->out_code;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_generate_sfc_device(dlscqp_ctx *c, const dlscqp_world *world, const dlscqp_sfc_gen *gen, void *stream) {
    g_last_error.clear();
    if (!c || !world || !gen) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (gen->n_agents <= 0) return DLSCQP_SUCCESS;
    if (!gen->radius || !gen->max_vel || !gen->sfcs ||
        (gen->initialize ? !gen->position : (!gen->initial_traj || !gen->goals || !gen->waypoints))) {
        g_last_error = "null buffer";
        return DLSCQP_ERR_INVALID_ARGUMENT;
    }
    SfcGenArgs a;
    {
        const int rc = sfc_build_args(c, world, gen, stream, a);
        if (rc != DLSCQP_SUCCESS) return rc;
    }
    static const int sfc_threads = std::getenv("DLSCQP_SFC_THREADS") ? std::atoi(std::getenv("DLSCQP_SFC_THREADS")) : SFC_THREADS;
    if (sfc_threads == 64) hipLaunchKernelGGL(dlscqp_sfc_kernel<64>, dim3(gen->n_agents), dim3(64), 0, (hipStream_t)stream, a);
    else if (sfc_threads == 128) hipLaunchKernelGGL(dlscqp_sfc_kernel<128>, dim3(gen->n_agents), dim3(128), 0, (hipStream_t)stream, a);
    else if (sfc_threads == 512) hipLaunchKernelGGL(dlscqp_sfc_kernel<512>, dim3(gen->n_agents), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dlscqp_sfc_kernel<SFC_THREADS>, dim3(gen->n_agents), dim3(SFC_THREADS), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

extern "C" void dlscqp_rows_to_soa(const dlscqp_lsc_row *rows, int64_t n_rows, float *nx, float *ny, float *nz,
                                   float *px, float *py, float *pz, double *d, uint32_t *key) {
    for (int64_t i = 0; i < n_rows; i++) {
        nx[i] = rows[i].normal[0]; ny[i] = rows[i].normal[1]; nz[i] = rows[i].normal[2];
        px[i] = rows[i].point[0]; py[i] = rows[i].point[1]; pz[i] = rows[i].point[2];
        d[i] = rows[i].d;
        key[i] = DLSCQP_KEY(rows[i].m, rows[i].i, rows[i].slack, rows[i].flags & DLSCQP_ROW_DYNAMIC);
    }
}

// ---------------------------------------------------------------------------------------------
// one launch per replanning step (include/dlscqp.h: dlscqp_fused_step_device)
// ---------------------------------------------------------------------------------------------
// A mission of the reference is ten agents: its replanning step was four or five launches of a few microseconds each, one behind
// the other (neighbour lists, corridor boxes, half-spaces, waypoint trap, goal LP + QP with the state update in its epilogue), and
// the launches' gaps were most of the step.  Here the stages are device functions of ONE kernel: a workgroup of 256 threads per
// agent runs them one after the other -- the agent's lists, boxes and rows go through memory as before (they stay in L2), the
// workgroup's own barrier orders them.  What crosses agents is read-only until the epilogue: every stage reads the OTHER agents'
// positions, previous plans and sub-goals of the step before (multi_sync_simulator.cpp:468-524: every agent plans on step k - 1's
// messages), and only the epilogue (dlscqp_advance) overwrites them.  So ONE grid-wide barrier, in front of the epilogue: a
// workgroup arrives when its last read of other agents' data is done -- before its solve -- and waits after its solve, i.e. for
// the slowest agent's stages, which the step waits for anyway.  All workgroups are resident (at most one per CU, checked at launch).
// The barrier is sense-reversing (arrivals, generation): nothing on the host has to count launches, a captured launch replays.
namespace {
struct FusedArgs {  // the stages in front of the plan stage (whose arguments are the plan kernel's three)
    NbrArgs nbr;
    SfcGenArgs sfc;
    LscGenArgs lsc;
    TrapArgs trap;
    int have_nbr, have_sfc, have_trap;
};
constexpr int FUSED_THREADS = 256;
constexpr int FUSED_WPA = FUSED_THREADS / 64;

template <int RC, int DIMS, int FS, int MS, int KU, int SU, int SLK>
__global__ __launch_bounds__(FUSED_THREADS, 1) void dlscqp_fused_kernel(KParams kp, MemLayout lay, KernelArgs a, FusedArgs f) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int k = (int)blockIdx.x, tid = (int)threadIdx.x;
    // (read before this workgroup arrives: the generation cannot change until it has)
    const unsigned int gen = a.grid ? __hip_atomic_load(a.grid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (f.have_nbr) {  // uniform
        nbr_gather_agent(f.nbr, k, tid);
        __syncthreads();
    }
    if (f.have_sfc) {
        sfc_agent<FUSED_THREADS>(f.sfc, k);
        __syncthreads();
    }
    lsc_agent<FUSED_THREADS>(f.lsc, k);
    __syncthreads();
    if (f.have_trap) {
        if (tid < 64) trap_agent(f.trap, k, tid);
        __syncthreads();
    }
    if (a.grid && tid == 0) {
        // every read this workgroup makes of OTHER agents' positions, plans and sub-goals has returned (the barriers above waited
        // for them): arrive.  The last one to arrive opens the next generation.
        __threadfence();
        const unsigned int old = __hip_atomic_fetch_add(a.grid, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned int)a.grid_n) {
            __hip_atomic_store(a.grid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.grid + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    WaveGpuT<false> w;
    w.lane = tid & 63;
    w.grid_gen = gen;
    plan_slot<RC, DIMS, FS, MS, KU, SU, SLK, WS_LDS, false, FUSED_WPA>(kp, lay, a, lds, w, k, k, 0);
}

typedef void (*fused_kernel_t)(KParams, MemLayout, KernelArgs, FusedArgs);
// the instantiations the reference's missions use (3-D M = 5, 2-D M = 10, 3-D M = 10; with and without slack obstacles)
fused_kernel_t fused_kernel_for(int id) {
    switch (id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID:                                       \
        if constexpr ((DIMS) > 0 && !((KU) == 2 && (SU) == 0)) return dlscqp_fused_kernel<RC, DIMS, FS, MS, KU, SU, SLK>; \
        else return nullptr;
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    default: return nullptr;
    }
}
}  // namespace

extern "C" int dlscqp_fused_step_device(dlscqp_ctx *c, const dlscqp_neighbour_search *ns, const dlscqp_world *world, const dlscqp_sfc_gen *sfc,
                                        const dlscqp_lsc_gen *gen, const dlscqp_waypoint_trap *trap, const dlscqp_device_batch *b, void *stream) {
    g_last_error.clear();
    if (!c || !gen || !b) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    const int N = b->n_agents;
    if (N <= 0) return DLSCQP_SUCCESS;
    (void)hipSetDevice(c->cfg.device);
    // one workgroup per agent, every workgroup resident: at most one per CU
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->cfg.device));
        n_cu = prop.multiProcessorCount;
    }
    if (N > DLSCQP_FUSED_MAX_AGENTS || N > n_cu) { g_last_error = "fused step: more agents than workgroups that are resident at once"; return DLSCQP_ERR_UNSUPPORTED; }
    if (gen->n_local != N || gen->local_ids || (ns && (ns->n_local != N || ns->n_total != N || ns->local_ids)) || (sfc && sfc->n_agents != N)) {
        g_last_error = "fused step: every stage over the same agents, all of them local";
        return DLSCQP_ERR_UNSUPPORTED;
    }
    if (b->agent_order || b->hint_steps != 0 || b->launch_count > 0 || !b->do_goal || !b->do_traj || !b->agents || !b->out_status ||
        (!b->row_offsets && b->row_pitch <= 0)) {
        g_last_error = "fused step: a plain launch of goal LP + QP over all agents";
        return DLSCQP_ERR_UNSUPPORTED;
    }
    if (sfc && (!world || sfc->initialize)) { g_last_error = "fused step: corridor boxes are shifted, not initialised, in a fused step"; return DLSCQP_ERR_UNSUPPORTED; }
    FusedArgs f;
    std::memset(&f.nbr, 0, sizeof f.nbr);
    std::memset(&f.sfc, 0, sizeof f.sfc);
    std::memset(&f.trap, 0, sizeof f.trap);
    f.have_nbr = f.have_sfc = f.have_trap = 0;
    int rc;
    if (ns) {
        rc = nbr_build_args(c, ns, f.nbr);
        if (rc != DLSCQP_SUCCESS) return rc;
        if (!f.nbr.direct) { g_last_error = "fused step: the neighbour search through cells is several launches"; return DLSCQP_ERR_UNSUPPORTED; }
        f.have_nbr = 1;
    }
    if (sfc) {
        rc = sfc_build_args(c, world, sfc, stream, f.sfc);
        if (rc != DLSCQP_SUCCESS) return rc;
        f.have_sfc = 1;
    }
    rc = lsc_gen_args(c, gen, &f.lsc);
    if (rc != DLSCQP_SUCCESS) return rc;
    if (trap && gen->n_dyn > 0) {
        if (!trap->agents || (c->kp.use_sfc && !trap->sfc_boxes)) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        f.trap.g = f.lsc;
        f.trap.agents = trap->agents;
        f.trap.sfc = c->kp.use_sfc ? trap->sfc_boxes : nullptr;
        f.trap.out_released = trap->out_released;
        f.trap.horizon = c->M * c->dt;
        f.have_trap = 1;
    }
    // the plan stage: prepared by launch() -- the build with helper wavefronts, a working set in what the other stages leave of the LDS
    FusedPlan fp;
    {
        const KernelChoice kc0 = select_kernel(c->kp, b->max_slack_obs, b->max_rows_per_agent, false);
        fused_kernel_t probe = kc0.fn ? fused_kernel_for(kc0.id) : nullptr;
        if (!probe) { g_last_error = "fused step: no fused build for this shape"; return DLSCQP_ERR_UNSUPPORTED; }
        hipFuncAttributes fa;
        HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(probe)));
        fp.static_lds = (int)fa.sharedSizeBytes + 1024;
    }
    dlscqp_device_batch bb = *b;
    bb.hint_mask_in = nullptr;  // (every agent starts at once: there is no order to find)
    bb.hint_mask_out = nullptr;
    bb.hint_mask_zero = nullptr;
    rc = launch(c, &bb, (hipStream_t)stream, &fp);
    if (rc != DLSCQP_SUCCESS) return rc;
    fused_kernel_t kfn = fused_kernel_for(fp.id);
    if (!kfn) { g_last_error = "fused step: no fused build for this shape"; return DLSCQP_ERR_UNSUPPORTED; }
    if (fp.lay.bytes + fp.static_lds > LDS_LIMIT) { g_last_error = "fused step: the working set and the stages' buffers exceed the LDS of a CU"; return DLSCQP_ERR_UNSUPPORTED; }
    fp.a.fresh_inputs = 1;
    if (fp.a.adv_on) {  // (without the epilogue nothing that other agents read is written: no barrier)
        fp.a.grid = c->fused_grid;
        fp.a.grid_n = N;
    }
    if (fp.lay.bytes > c->fused_dyn_lds[fp.id]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, fp.lay.bytes));
        c->fused_dyn_lds[fp.id] = fp.lay.bytes;
    }
    hipLaunchKernelGGL(kfn, dim3(N), dim3(FUSED_THREADS), fp.lay.bytes, (hipStream_t)stream, fp.kp, fp.lay, fp.a, f);
    HIP_TRY(hipGetLastError());
    return DLSCQP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// host-buffer entry points: marshal into pinned staging, H2D, launch, D2H
// ---------------------------------------------------------------------------------------------
namespace {
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}

static int plan_host(dlscqp_ctx *c, int32_t n_agents, const dlscqp_agent *agents, const int64_t *row_offsets,
                     const dlscqp_lsc_row *rows, const float *sfc_boxes, int32_t max_slack_obs, int do_goal,
                     int do_traj, float *out_goal, int32_t *out_goal_status, float *out_ctrl, double *out_cost,
                     double *out_slack_cost, int32_t *out_status, int declared_rows_per_agent = 0,
                     double *out_conflict = nullptr, int conflict_stride = 0) {
    g_last_error.clear();
    if (!c) { g_last_error = "null context"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (n_agents < 0) { g_last_error = "negative n_agents"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (n_agents == 0) return DLSCQP_SUCCESS;
    if (!agents || !row_offsets) { g_last_error = "null buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    const int64_t nrows = row_offsets[n_agents];
    if (nrows < 0 || nrows >= (int64_t)1 << 31 || (nrows > 0 && !rows)) { g_last_error = "bad row_offsets"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    (void)hipSetDevice(c->cfg.device);
    const int M = c->M, n1 = c->n + 1;
    const bool use_sfc = c->kp.use_sfc && sfc_boxes;
    if (c->kp.use_sfc && !sfc_boxes) { g_last_error = "world_use_octomap is set but sfc_boxes is NULL"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (max_slack_obs < 0) max_slack_obs = 0;
    // ---- staging layout (inputs then outputs), 256-byte aligned sections
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t rpad = (size_t)nrows + 64;
    const size_t o_agents = take(sizeof(dlscqp_agent) * n_agents);
    const size_t o_off = take(4 * ((size_t)n_agents + 1));
    const size_t o_nx = take(4 * rpad), o_ny = take(4 * rpad), o_nz = take(4 * rpad);
    const size_t o_px = take(4 * rpad), o_py = take(4 * rpad), o_pz = take(4 * rpad);
    const size_t o_d = take(8 * rpad), o_key = take(4 * rpad);
    const size_t o_sfc = take(use_sfc ? sizeof(float) * 6 * M * n_agents : 0);
    const size_t in_bytes = o;
    const size_t o_goal = take(sizeof(float) * 3 * n_agents);
    const size_t o_gst = take(4 * (size_t)n_agents);
    const size_t o_ctrl = take(sizeof(float) * 3 * M * n1 * n_agents);
    const size_t o_cost = take(8 * (size_t)n_agents);
    const size_t o_slk = take(8 * (size_t)n_agents * std::max(max_slack_obs, 1));
    const size_t o_st = take(4 * (size_t)n_agents);
    const size_t o_cf = take(out_conflict ? 8 * (size_t)n_agents * (size_t)conflict_stride : 0);  // dlscqp_device_batch.out_conflict
    const size_t total = o;
    if (total > c->stage_cap) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->d_stage = nullptr; c->h_stage = nullptr; c->stage_cap = 0;
        size_t cap = total + total / 2;
        HIP_TRY(hipMalloc((void **)&c->d_stage, cap));
        HIP_TRY(hipHostMalloc((void **)&c->h_stage, cap, hipHostMallocDefault));
        c->stage_cap = cap;
    }
    unsigned char *h = c->h_stage, *d = c->d_stage;
    std::memcpy(h + o_agents, agents, sizeof(dlscqp_agent) * n_agents);
    int32_t *off32 = reinterpret_cast<int32_t *>(h + o_off);
    int kmax = 0;
    for (int a = 0; a <= n_agents; a++) {
        off32[a] = (int32_t)row_offsets[a];
        if (a > 0) {
            int64_t k = row_offsets[a] - row_offsets[a - 1];
            if (k < 0) { g_last_error = "row_offsets not monotone"; return DLSCQP_ERR_INVALID_ARGUMENT; }
            kmax = std::max<int>(kmax, (int)k);
        }
    }
    dlscqp_rows_to_soa(rows, nrows, (float *)(h + o_nx), (float *)(h + o_ny), (float *)(h + o_nz), (float *)(h + o_px),
                       (float *)(h + o_py), (float *)(h + o_pz), (double *)(h + o_d), (uint32_t *)(h + o_key));
    if (use_sfc) std::memcpy(h + o_sfc, sfc_boxes, sizeof(float) * 6 * M * n_agents);
    HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d + in_bytes, 0, total - in_bytes, c->stream));
    // verdict defaults: an agent the kernel never reaches keeps a failure code (0x03030303 is not DLSCQP_OK), not "solved"
    HIP_TRY(hipMemsetAsync(d + o_st, DLSCQP_NUMERICAL, 4 * (size_t)n_agents, c->stream));
    HIP_TRY(hipMemsetAsync(d + o_gst, DLSCQP_NUMERICAL, 4 * (size_t)n_agents, c->stream));
    dlscqp_device_batch b;
    std::memset(&b, 0, sizeof b);
    b.n_agents = n_agents;
    b.max_slack_obs = max_slack_obs;
    b.do_goal = do_goal;
    b.do_traj = do_traj;
    b.max_rows_per_agent = std::max(kmax, declared_rows_per_agent);
    b.agents = (const dlscqp_agent *)(d + o_agents);
    b.row_offsets = (const int32_t *)(d + o_off);
    b.nx = (const float *)(d + o_nx); b.ny = (const float *)(d + o_ny); b.nz = (const float *)(d + o_nz);
    b.px = (const float *)(d + o_px); b.py = (const float *)(d + o_py); b.pz = (const float *)(d + o_pz);
    b.d = (const double *)(d + o_d);
    b.key = (const uint32_t *)(d + o_key);
    b.sfc_boxes = use_sfc ? (const float *)(d + o_sfc) : nullptr;
    b.out_goal = (float *)(d + o_goal);
    b.out_goal_status = (int32_t *)(d + o_gst);
    b.out_ctrl = (float *)(d + o_ctrl);
    b.out_cost = (double *)(d + o_cost);
    b.out_slack_cost = max_slack_obs > 0 ? (double *)(d + o_slk) : nullptr;
    b.out_status = (int32_t *)(d + o_st);
    if (out_conflict) {
        b.out_conflict = (double *)(d + o_cf);
        b.conflict_stride = conflict_stride;
    }
    int rc = launch(c, &b, c->stream);
    if (rc != DLSCQP_SUCCESS) return rc;
    HIP_TRY(hipMemcpyAsync(h + in_bytes, d + in_bytes, total - in_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t cs = (size_t)3 * M * n1;
    for (int a = 0; a < n_agents; a++) {
        const int32_t gst = reinterpret_cast<int32_t *>(h + o_gst)[a];
        const int32_t st = reinterpret_cast<int32_t *>(h + o_st)[a];
        if (do_goal) {
            if (out_goal_status) out_goal_status[a] = gst;
            if (out_goal && gst != DLSCQP_INFEASIBLE) std::memcpy(out_goal + (size_t)a * 3, h + o_goal + sizeof(float) * 3 * a, sizeof(float) * 3);
        }
        if (do_traj) {
            if (out_status) out_status[a] = st;
            if (st == DLSCQP_OK) {
                if (out_ctrl) std::memcpy(out_ctrl + a * cs, h + o_ctrl + sizeof(float) * cs * a, sizeof(float) * cs);
                if (out_cost) out_cost[a] = reinterpret_cast<double *>(h + o_cost)[a];
                if (out_slack_cost && max_slack_obs > 0)
                    std::memcpy(out_slack_cost + (size_t)a * max_slack_obs, h + o_slk + 8 * (size_t)a * max_slack_obs, 8 * (size_t)max_slack_obs);
            } else {
                if (out_cost) out_cost[a] = 0;
            }
        }
    }
    if (out_conflict) std::memcpy(out_conflict, h + o_cf, 8 * (size_t)n_agents * (size_t)conflict_stride);
    return DLSCQP_SUCCESS;
}


// ---------------------------------------------------------------------------------------------
// self-test (include/dlscqp.h: dlscqp_selftest)
// ---------------------------------------------------------------------------------------------
extern "C" int dlscqp_selftest(dlscqp_ctx *c) {
    g_last_error.clear();
    if (!c) { g_last_error = "null context"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    const int M = c->M, n = c->n, n1 = n + 1, dim = c->dim;
    const double dt = c->dt;
    const float z = dim == 2 ? (float)c->cfg.world_z_2d : (float)(0.5 * (c->cfg.world_min[2] + c->cfg.world_max[2]));
    const float x0 = (float)(0.5 * (c->cfg.world_min[0] + c->cfg.world_max[0])), y0 = (float)(0.5 * (c->cfg.world_min[1] + c->cfg.world_max[1]));
    const double horizon = M * dt;
    // how far agent 0 may go along x: less than it coasts when it brakes as gently as it can (v T / 2 = 0.1 horizon), eight
    // times its stopping distance at full deceleration -- the wall is active whatever the weights are, and can be met
    const float wall = (float)(0.08 * horizon);
    dlscqp_agent ag[2];
    std::memset(ag, 0, sizeof ag);
    // (inside any world and any communication range: agent 1 a quarter of the world's width aside at most, the waypoints within
    // half the communication range of the start -- the rows of traj_optimizer.cpp:490-513 would cut the problem off otherwise)
    const float aside = (float)std::min(0.5, 0.25 * std::max(0.0, c->cfg.world_max[1] - c->cfg.world_min[1]));
    for (int a = 0; a < 2; a++) {
        ag[a].position[0] = x0; ag[a].position[1] = y0 + (a ? aside : 0.0f); ag[a].position[2] = z;
        ag[a].velocity[0] = 0.2f;
        float reach = (float)(0.45 * horizon);
        if (c->cfg.communication_range > 0) reach = std::min(reach, (float)(0.8 * c->cfg.communication_range));  // waypoint at reach / 2
        ag[a].current_goal_point[0] = x0 + reach; ag[a].current_goal_point[1] = ag[a].position[1]; ag[a].current_goal_point[2] = z;
        ag[a].next_waypoint[0] = x0 + 0.5f * reach; ag[a].next_waypoint[1] = ag[a].position[1]; ag[a].next_waypoint[2] = z;
        ag[a].id = a; ag[a].n_obs = a == 0 ? 1 : 0; ag[a].n_slack_obs = 0;
        ag[a].max_vel = 1.0; ag[a].max_acc = 2.0; ag[a].radius = 0.15; ag[a].nominal_velocity = 1.0;
    }
    // agent 0: one "neighbour" whose half-spaces are the wall x <= x0 + wall at every control point of the model
    std::vector<dlscqp_lsc_row> rows;
    for (int m = 0; m < M; m++)
        for (int i = 0; i < n1; i++) {
            if (m == 0 && i < c->phi) continue;
            dlscqp_lsc_row r;
            std::memset(&r, 0, sizeof r);
            r.normal[0] = -1.0f;
            r.point[0] = x0 + wall; r.point[1] = y0; r.point[2] = z;
            r.d = 0.0; r.m = (int16_t)m; r.i = (int8_t)i; r.flags = 0; r.obs = 0; r.slack = -1;
            rows.push_back(r);
        }
    const int64_t ro[3] = {0, (int64_t)rows.size(), (int64_t)rows.size()};
    std::vector<float> sfc;
    if (c->kp.use_sfc) {
        sfc.resize((size_t)2 * M * 6);
        for (int a = 0; a < 2; a++)
            for (int m = 0; m < M; m++)
                for (int k = 0; k < 3; k++) {
                    sfc[((size_t)a * M + m) * 6 + k] = (float)c->cfg.world_min[k];
                    sfc[((size_t)a * M + m) * 6 + 3 + k] = (float)c->cfg.world_max[k];
                }
    }
    struct Variant { const char *name; int slack_obs, declared_rows, force; };
    const Variant variants[] = {{"register-cache instantiation", 0, 0, 0}, {"instantiation for long row lists", 0, 1000, 0},
                                {"instantiation with slack obstacles", 1, 0, 0}, {"runtime-shape instantiation", 1, 1000, 1},
                                {"working set in HBM", 1, 1000, 2}};
    const size_t cs = (size_t)3 * M * n1;
    std::vector<float> ref;
    // every BUILD of every instantiation: the standard one (what a launch of more than a thousand agents runs: two wavefronts per
    // SIMD, the tightest register budget), the WIDE one and the one with helper wavefronts (an instantiation that has no such
    // build runs its standard one again: a few microseconds)
    static const char *const builds[] = {"", "standard build", "WIDE build", "build with helper wavefronts"};
    for (const Variant &v : variants)
    for (int build = 1; build <= 3; build++) {
        if (v.force == 2 && build > 1) continue;  // the HBM working set has one build
        std::vector<float> ctrl(2 * cs, 0.0f);
        std::vector<double> cost(2, 0.0), slack(2, 0.0);
        int32_t st[2] = {-99, -99};
        c->force_instance = v.force;
        c->force_build = build;
        const int rc = plan_host(c, 2, ag, ro, rows.data(), sfc.empty() ? nullptr : sfc.data(), v.slack_obs, 0, 1, nullptr, nullptr, ctrl.data(),
                                 cost.data(), slack.data(), st, v.declared_rows);
        c->force_instance = 0;
        c->force_build = 0;
        if (rc == DLSCQP_ERR_UNSUPPORTED) continue;  // a shape this build has no such instantiation for
        if (rc != DLSCQP_SUCCESS) { g_last_error = std::string("selftest (") + v.name + "): " + g_last_error; return rc; }
        auto fail = [&](const char *what, int a, double val) {
            char buf[256];
            std::snprintf(buf, sizeof buf, "selftest failed (%s, %s): %s (agent %d, %.3g) -- this build of libdlscqp does not plan correctly on this "
                          "machine; rebuild with the flags of __graft_entry__.py", v.name, builds[build], what, a, val);
            g_last_error = buf;
            return DLSCQP_ERR_SELFTEST;
        };
        for (int a = 0; a < 2; a++) {
            if (st[a] != DLSCQP_OK) return fail("verdict is not DLSCQP_OK", a, st[a]);
            const float *p = ctrl.data() + a * cs;
            auto at = [&](int m, int i, int k) { return (double)p[((size_t)m * n1 + i) * 3 + k]; };
            for (int k = 0; k < dim; k++) {
                const double s1 = n / dt, s2 = n * (n - 1) / (dt * dt);
                if (std::fabs(at(0, 0, k) - ag[a].position[k]) > 1e-6) return fail("initial position", a, at(0, 0, k));
                if (std::fabs((at(0, 1, k) - at(0, 0, k)) * s1 - ag[a].velocity[k]) > 1e-3) return fail("initial velocity", a, (at(0, 1, k) - at(0, 0, k)) * s1);
                if (std::fabs((at(0, 2, k) - 2 * at(0, 1, k) + at(0, 0, k)) * s2) > 5e-2) return fail("initial acceleration", a, 0);
                for (int m = 1; m < M; m++) {
                    if (std::fabs(at(m, 0, k) - at(m - 1, n, k)) > 1e-6) return fail("position continuity", a, m);
                    if (std::fabs((at(m, 1, k) - at(m, 0, k)) - (at(m - 1, n, k) - at(m - 1, n - 1, k))) > 2e-6) return fail("velocity continuity", a, m);
                }
                if (c->kp.stop_mode && std::fabs(at(M - 1, n, k) - at(M - 1, n - 1, k)) > 2e-6) return fail("terminal stop", a, k);
                for (int m = 0; m < M; m++)
                    for (int i = 0; i < n; i++)
                        if (std::fabs((at(m, i + 1, k) - at(m, i, k)) * s1) > ag[a].max_vel + 1e-3) return fail("velocity limit", a, m);
            }
            double far = -1e30;
            for (int m = 0; m < M; m++)
                for (int i = 0; i < n1; i++) {
                    if (a == 0 && !(m == 0 && i < c->phi) && at(m, i, 0) > (double)x0 + (double)wall + 2e-6) return fail("the wall is crossed", a, at(m, i, 0));
                    if (std::fabs(at(m, i, 1) - ag[a].position[1]) > 1e-5) return fail("lateral drift", a, at(m, i, 1));
                    far = std::max(far, at(m, i, 0));
                }
            // the wall matters: without it the agent goes past it (agent 1 does), with it the plan touches it
            if (a == 0 && std::fabs(far - ((double)x0 + (double)wall)) > 1e-5) return fail("the plan does not touch the wall", a, far);
            if (a == 1 && !(far > (double)x0 + (double)wall + 1e-4)) return fail("the free agent stops short of where the wall would be", a, far);
        }
        if (ref.empty()) {
            ref = ctrl;
        } else {
            for (size_t j = 0; j < ctrl.size(); j++)
                if (std::fabs((double)ctrl[j] - (double)ref[j]) > 1e-6) return fail("differs from the first instantiation's plan", (int)(j / cs), (double)ctrl[j] - (double)ref[j]);
        }
    }
    return DLSCQP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// the exchange of the solved plans and one call per replanning step (include/dlscqp.h: dlscqp_exchange_*, dlscqp_step_device)
// ---------------------------------------------------------------------------------------------
// librccl is loaded when the first exchange is created, not linked: a process that never shards its agents needs no RCCL,
// and a process that already holds one (torch bundles its own) keeps exactly that one.
namespace {
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
bool load_rccl() {
    if (g_rccl.handle) return true;
    const char *names[] = {std::getenv("DLSCQP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)  // a copy that is already in the process first (RTLD_NOLOAD), then by name
        if (n && n[0] && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!h)
        for (const char *n : names)
            if (n && n[0] && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { g_last_error = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?"); return false; }
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<int (*)(void **, int, NcclId, int)>(dlsym(h, "ncclCommInitRank"));
    a.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(h, "ncclAllGather"));
    a.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(h, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) { g_last_error = "librccl lacks the ncclAllGather entry points"; return false; }
    g_rccl = a;
    return true;
}
bool rccl_ok(int rc, const char *what) {
    if (rc == 0) return true;
    g_last_error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return false;
}

// one record per agent: control points, sub-goal, verdict (the int32's bits) -- E + 4 floats
// (a few hundred workgroups that walk the records' words, not one workgroup per agent: beside the plan kernels of the next step a
// launch gets its wave slots one by one, and 32 768 two-wavefront workgroups took 0.12 ms to get theirs)
__global__ __launch_bounds__(256) void dlscqp_pack_kernel(const float *ctrl, const float *goal, const int32_t *status, float *rec, int n, int E) {
    const int W = E + 4;
    const size_t total = (size_t)n * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t a = e / W;
        const int j = (int)(e - a * W);
        float v;
        if (j < E) v = ctrl[a * E + j];
        else if (j < E + 3) v = goal[a * 3 + (j - E)];
        else v = __int_as_float(status[a]);
        rec[e] = v;
    }
}
}  // namespace

struct dlscqp_exchange {
    int device = 0, rank = 0, world = 1, n_local = 0, E = 0, W = 0, local_only = 0;
    void *comm = nullptr;                    // ncclComm_t
    hipStream_t comm_stream = nullptr, join_stream = nullptr;
    float *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};
    bool own_recv = true;
    hipEvent_t ready[2] = {nullptr, nullptr}, packed[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    bool posted[2] = {false, false};
    long long k = 0;                         // exchanges posted so far
    // "the communication stream has read what post k sent" for the last SRC_RING posts (post k: src_read[k % SRC_RING]): a caller that
    // rotates through B tables overwrites, at step k, the table post k - B read (dlscqp_exchange_before_overwrite_rotating)
    static constexpr int SRC_RING = 8;
    hipEvent_t src_read[SRC_RING] = {};
    int last = -1;
    std::vector<hipEvent_t> planned;         // dlscqp_step_device: one per group
    hipEvent_t table_ready = nullptr;        // serial mode of dlscqp_step_device
    bool have_table_ready = false;
    // timing (dlscqp_exchange_set_timing): per exchange t0 (communication stream reached the pack), packed, gathered; ready is the compute side
    std::vector<hipEvent_t> tev;             // 4 per exchange: ready, t0, packed, done
    int t_cap = 0, t_used = 0;
};

extern "C" int dlscqp_exchange_unique_id(void *id_out) {
    g_last_error.clear();
    if (!id_out) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (!load_rccl()) return DLSCQP_ERR_UNSUPPORTED;
    if (!rccl_ok(g_rccl.GetUniqueId(id_out), "ncclGetUniqueId")) return DLSCQP_ERR_HIP;
    return DLSCQP_SUCCESS;
}

extern "C" void dlscqp_exchange_destroy(dlscqp_exchange *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
    if (x->join_stream) (void)hipStreamSynchronize(x->join_stream);
    if (x->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(x->comm);
    for (int b = 0; b < 2; b++) {
        if (x->send[b]) (void)hipFree(x->send[b]);
        if (x->recv[b] && x->own_recv) (void)hipFree(x->recv[b]);
        if (x->ready[b]) (void)hipEventDestroy(x->ready[b]);
        if (x->packed[b]) (void)hipEventDestroy(x->packed[b]);
        if (x->done[b]) (void)hipEventDestroy(x->done[b]);
    }
    for (auto e : x->src_read)
        if (e) (void)hipEventDestroy(e);
    for (auto e : x->planned) (void)hipEventDestroy(e);
    for (auto e : x->tev) (void)hipEventDestroy(e);
    if (x->table_ready) (void)hipEventDestroy(x->table_ready);
    if (x->comm_stream) (void)hipStreamDestroy(x->comm_stream);
    if (x->join_stream) (void)hipStreamDestroy(x->join_stream);
    delete x;
}

extern "C" int dlscqp_exchange_create(dlscqp_ctx *c, const dlscqp_exchange_config *cfg, dlscqp_exchange **out) {
    g_last_error.clear();
    if (!c || !cfg || !out) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world || cfg->n_local < 1) { g_last_error = "rank / world / n_local out of range"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (cfg->local_only && cfg->world != 1) { g_last_error = "local_only needs world == 1"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (cfg->world > 1 && !cfg->unique_id) { g_last_error = "unique_id: the 128 bytes of rank 0's dlscqp_exchange_unique_id"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    dlscqp_exchange *x = new dlscqp_exchange();
    x->device = c->cfg.device;
    x->rank = cfg->rank; x->world = cfg->world; x->n_local = cfg->n_local; x->local_only = cfg->local_only;
    x->E = c->M * (c->n + 1) * 3;
    x->W = x->E + 4;
    auto fail = [&](int rc) { const std::string why = g_last_error; dlscqp_exchange_destroy(x); g_last_error = why; return rc; };
    if (!x->local_only) {
        if (!load_rccl()) return fail(DLSCQP_ERR_UNSUPPORTED);
        NcclId id;
        if (cfg->unique_id) std::memcpy(&id, cfg->unique_id, sizeof id);
        else if (!rccl_ok(g_rccl.GetUniqueId(&id), "ncclGetUniqueId")) return fail(DLSCQP_ERR_HIP);
        if (!rccl_ok(g_rccl.CommInitRank(&x->comm, cfg->world, id, cfg->rank), "ncclCommInitRank")) return fail(DLSCQP_ERR_HIP);
    }
    // the communication stream at the highest priority: the collective's few workgroups share the GPU with the plan kernel of
    // the next step and should get the first free slots
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&x->comm_stream, hipStreamNonBlocking, hi) != hipSuccess) { g_last_error = "hipStreamCreateWithPriority"; return fail(DLSCQP_ERR_HIP); }
    if (hipStreamCreateWithFlags(&x->join_stream, hipStreamNonBlocking) != hipSuccess) { g_last_error = "hipStreamCreate"; return fail(DLSCQP_ERR_HIP); }
    const size_t sb = (size_t)x->n_local * x->W * sizeof(float), rb = sb * (size_t)x->world;
    x->own_recv = !(cfg->recv_buffers[0] && cfg->recv_buffers[1]);
    for (int b = 0; b < 2; b++) {
        if (!x->own_recv) x->recv[b] = cfg->recv_buffers[b];
        if (hipMalloc((void **)&x->send[b], sb) != hipSuccess || (x->own_recv && hipMalloc((void **)&x->recv[b], rb) != hipSuccess)) { g_last_error = "hipMalloc(exchange buffers)"; return fail(DLSCQP_ERR_HIP); }
        if (hipEventCreateWithFlags(&x->ready[b], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x->packed[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&x->done[b], hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate"; return fail(DLSCQP_ERR_HIP); }
    }
    if (hipEventCreateWithFlags(&x->table_ready, hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate"; return fail(DLSCQP_ERR_HIP); }
    for (auto &e : x->src_read)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { g_last_error = "hipEventCreate"; return fail(DLSCQP_ERR_HIP); }
    *out = x;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_exchange_set_timing(dlscqp_exchange *x, int32_t capacity) {
    g_last_error.clear();
    if (!x || capacity < 0) return DLSCQP_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(x->device));
    HIP_TRY(hipStreamSynchronize(x->comm_stream));
    while ((int)x->tev.size() < 4 * capacity) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        x->tev.push_back(e);
    }
    x->t_cap = capacity;
    x->t_used = 0;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_exchange_get_times(dlscqp_exchange *x, float *pack_ms, float *all_gather_ms, float *lag_ms, int32_t capacity, int32_t *n_out) {
    g_last_error.clear();
    if (!x || !n_out) return DLSCQP_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(x->device));
    const int n = std::min<int>(x->t_used, capacity);
    for (int i = 0; i < n; i++) {
        hipEvent_t *e = &x->tev[4 * (size_t)i];
        HIP_TRY(hipEventSynchronize(e[3]));
        if (pack_ms) HIP_TRY(hipEventElapsedTime(&pack_ms[i], e[1], e[2]));
        if (all_gather_ms) HIP_TRY(hipEventElapsedTime(&all_gather_ms[i], e[2], e[3]));
        if (lag_ms) HIP_TRY(hipEventElapsedTime(&lag_ms[i], e[0], e[3]));
    }
    *n_out = n;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_exchange_before_overwrite(dlscqp_exchange *x, void *stream) {
    g_last_error.clear();
    if (!x) return DLSCQP_ERR_INVALID_ARGUMENT;
    const int b = (int)(x->k & 1);  // the buffers the NEXT post uses: their previous exchange (post k - 2) read the caller's tensors of then
    if (x->posted[b]) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, x->packed[b], 0));
    return DLSCQP_SUCCESS;
}

// The same for a caller that rotates through `tables` output tables (step k writes table k mod tables): the table the next step writes was
// read by post k - tables, not by post k - 2.  With two tables a launch of step k waits until the collective of step k - 2 has run -- and
// that collective waited for EVERY group's launch of step k - 2, then for room on a GPU that the launches of step k - 1 fill: measured on
// one MI355X (four groups of 16 384 agents, kernel trace in profiles/r06_force_dist_trace.md) the copy of step k - 2 starts ~250 us into
// step k - 1 and every group starts step k in lockstep behind it -- the groups' independence is gone, 0.338 -> 0.469 ms
// per step.  With the rotation's real length the wait refers to a collective that finished steps ago.
extern "C" int dlscqp_exchange_before_overwrite_rotating(dlscqp_exchange *x, void *stream, int32_t tables) {
    g_last_error.clear();
    if (!x) return DLSCQP_ERR_INVALID_ARGUMENT;
    if (tables < 2 || tables > dlscqp_exchange::SRC_RING) { g_last_error = "before_overwrite_rotating: 2 <= tables <= 8"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    const long long k = x->k - tables;  // the post that read the table the next step writes
    if (k >= 0) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, x->src_read[k % dlscqp_exchange::SRC_RING], 0));
    return DLSCQP_SUCCESS;
}

// `stream` waits until the communication stream has read the arrays of the post just made (a caller that rewrites them right away)
extern "C" int dlscqp_exchange_wait_packed(dlscqp_exchange *x, void *stream) {
    g_last_error.clear();
    if (!x || x->last < 0) { g_last_error = "no exchange has been posted"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, x->packed[x->last], 0));
    return DLSCQP_SUCCESS;
}

static int exchange_post(dlscqp_exchange *x, const float *ctrl, const float *goal, const int32_t *status, const float *records, void *ready_stream);

extern "C" int dlscqp_exchange_post(dlscqp_exchange *x, const float *ctrl, const float *goal, const int32_t *status, void *ready_stream) {
    g_last_error.clear();
    if (!x || !ctrl || !goal || !status) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    return exchange_post(x, ctrl, goal, status, nullptr, ready_stream);
}

extern "C" int dlscqp_exchange_post_records(dlscqp_exchange *x, const float *records, void *ready_stream) {
    g_last_error.clear();
    if (!x || !records) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    return exchange_post(x, nullptr, nullptr, nullptr, records, ready_stream);
}

static int exchange_post(dlscqp_exchange *x, const float *ctrl, const float *goal, const int32_t *status, const float *records, void *ready_stream) {
    HIP_TRY(hipSetDevice(x->device));
    const int b = (int)(x->k & 1);
    hipEvent_t src_read = x->src_read[x->k % dlscqp_exchange::SRC_RING];
    x->k++;
    const bool timed = x->t_used < x->t_cap;
    hipEvent_t *te = timed ? &x->tev[4 * (size_t)x->t_used] : nullptr;
    // the communication stream is in order behind exchange k - 2, which used the same buffers
    if ((hipStream_t)ready_stream != x->comm_stream) {  // (dlscqp_step_device makes the communication stream wait for the launches itself)
        HIP_TRY(hipEventRecord(x->ready[b], (hipStream_t)ready_stream));
        HIP_TRY(hipStreamWaitEvent(x->comm_stream, x->ready[b], 0));
    }
    if (timed) HIP_TRY(hipEventRecord(te[0], (hipStream_t)ready_stream));
    if (timed) HIP_TRY(hipEventRecord(te[1], x->comm_stream));
    const float *src = records ? records : x->send[b];
    if (!records) {
        const size_t words = (size_t)x->n_local * x->W;
        const int grid = (int)std::min<size_t>((words + 255) / 256, 512);
        hipLaunchKernelGGL(dlscqp_pack_kernel, dim3(grid), dim3(256), 0, x->comm_stream, ctrl, goal, status, x->send[b], x->n_local, x->E);
    }
    HIP_TRY(hipGetLastError());
    if (!records) {  // the caller's three arrays have been read
        HIP_TRY(hipEventRecord(x->packed[b], x->comm_stream));
        HIP_TRY(hipEventRecord(src_read, x->comm_stream));
    }
    if (timed) HIP_TRY(hipEventRecord(te[2], x->comm_stream));
    if (x->local_only) {
        HIP_TRY(hipMemcpyAsync(x->recv[b], src, (size_t)x->n_local * x->W * sizeof(float), hipMemcpyDeviceToDevice, x->comm_stream));
    } else {
        if (!rccl_ok(g_rccl.AllGather(src, x->recv[b], (size_t)x->n_local * x->W, /* ncclFloat32 */ 7, x->comm, x->comm_stream), "ncclAllGather")) return DLSCQP_ERR_HIP;
    }
    HIP_TRY(hipEventRecord(x->done[b], x->comm_stream));
    if (records) {  // sent as they lie: read when the collective is through
        HIP_TRY(hipEventRecord(x->packed[b], x->comm_stream));
        HIP_TRY(hipEventRecord(src_read, x->comm_stream));
    }
    if (timed) { HIP_TRY(hipEventRecord(te[3], x->comm_stream)); x->t_used++; }
    x->posted[b] = true;
    x->last = b;
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_exchange_wait(dlscqp_exchange *x, void *stream, const float **table_out) {
    g_last_error.clear();
    if (!x || x->last < 0) { g_last_error = "no exchange has been posted"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, x->done[x->last], 0));
    if (table_out) *table_out = x->recv[x->last];
    return DLSCQP_SUCCESS;
}

// MultiSyncSimulator::plan + broadcastMsgs (multi_sync_simulator.cpp:516-524, 468-514) for every group of swarms of this rank: ONE
// call queues the groups' launches of a replanning step on their streams and the exchange of the step's plans behind them.
extern "C" int dlscqp_step_device(const dlscqp_step_args *s) {
    g_last_error.clear();
    if (!s || s->n_groups < 1 || !s->groups) { g_last_error = "null argument"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    dlscqp_exchange *x = s->exchange;
    const int mode = x ? s->exchange_mode : DLSCQP_EXCHANGE_NONE;
    if (x && mode != DLSCQP_EXCHANGE_NONE && !s->records && (!s->ctrl || !s->goal || !s->status)) { g_last_error = "exchange: null output table"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    if (x) {
        HIP_TRY(hipSetDevice(x->device));
        while ((int)x->planned.size() < s->n_groups) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            x->planned.push_back(e);
        }
    }
    for (int g = 0; g < s->n_groups; g++) {
        const dlscqp_step_group &G = s->groups[g];
        if (!G.ctx || !G.batch) { g_last_error = "null group"; return DLSCQP_ERR_INVALID_ARGUMENT; }
        hipStream_t st = (hipStream_t)G.stream;
        if (mode == DLSCQP_EXCHANGE_PIPELINED) {
            // the exchange of step k (packing included) runs on the communication stream while step k + 1 writes another table of
            // the caller's rotation: a launch only waits until the table it is about to overwrite has been packed
            const int rc = s->rotation > 2 ? dlscqp_exchange_before_overwrite_rotating(x, st, s->rotation) : dlscqp_exchange_before_overwrite(x, st);
            if (rc != DLSCQP_SUCCESS) return rc;
        } else if (mode == DLSCQP_EXCHANGE_SERIAL && x->have_table_ready) {
            HIP_TRY(hipStreamWaitEvent(st, x->table_ready, 0));  // the next launch waits for the exchange
        }
        const int rc = dlscqp_plan_batch_device(G.ctx, G.batch, G.stream);
        if (rc != DLSCQP_SUCCESS) return rc;
        if (mode != DLSCQP_EXCHANGE_NONE) HIP_TRY(hipEventRecord(x->planned[g], st));
    }
    if (mode == DLSCQP_EXCHANGE_NONE) return DLSCQP_SUCCESS;
    // the communication stream itself waits for the groups' launches of the step (no stream in between: streams share a few
    // hardware queues, and a stream that only waits can hold up the queue of a group it shares one with)
    // (measured, round 6, one rank, four groups of 16 384 agents: these four waits are 0.07 ms of a 0.44 ms step -- without them, which
    // is wrong, 0.369; with one, 0.404; hipStreamWriteValue64 / hipStreamWaitValue64 on signal memory instead of events: 0.535.  A wait
    // across hardware queues costs the waited-for queue too on this runtime)
    for (int g = 0; g < s->n_groups; g++) HIP_TRY(hipStreamWaitEvent(x->comm_stream, x->planned[g], 0));
    int rc = s->records ? dlscqp_exchange_post_records(x, s->records, x->comm_stream) : dlscqp_exchange_post(x, s->ctrl, s->goal, s->status, x->comm_stream);
    if (rc != DLSCQP_SUCCESS) return rc;
    if (mode == DLSCQP_EXCHANGE_SERIAL) {
        HIP_TRY(hipEventRecord(x->table_ready, x->comm_stream));  // (in order behind the collective)
        x->have_table_ready = true;
    }
    return DLSCQP_SUCCESS;
}

extern "C" int dlscqp_solve_batch(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents,
                                  const int64_t *row_offsets, const dlscqp_lsc_row *rows, const float *sfc_boxes,
                                  int32_t max_slack_obs, float *out_ctrl, double *out_cost, double *out_slack_cost,
                                  int32_t *out_status) {
    return plan_host(ctx, n_agents, agents, row_offsets, rows, sfc_boxes, max_slack_obs, 0, 1, nullptr, nullptr,
                     out_ctrl, out_cost, out_slack_cost, out_status);
}

extern "C" int dlscqp_solve_batch_conflict(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents, const int64_t *row_offsets,
                                           const dlscqp_lsc_row *rows, const float *sfc_boxes, int32_t max_slack_obs, int32_t *out_status,
                                           double *out_conflict, int32_t conflict_stride) {
    if (!out_conflict || conflict_stride <= 0) { g_last_error = "null conflict buffer"; return DLSCQP_ERR_INVALID_ARGUMENT; }
    return plan_host(ctx, n_agents, agents, row_offsets, rows, sfc_boxes, max_slack_obs, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr,
                     out_status, 0, out_conflict, conflict_stride);
}

extern "C" int dlscqp_goal_batch(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents,
                                 const int64_t *row_offsets, const dlscqp_lsc_row *rows, const float *sfc_boxes,
                                 float *out_goal, int32_t *out_goal_status) {
    return plan_host(ctx, n_agents, agents, row_offsets, rows, sfc_boxes, 0, 1, 0, out_goal, out_goal_status, nullptr,
                     nullptr, nullptr, nullptr);
}

extern "C" int dlscqp_plan_batch(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents,
                                 const int64_t *row_offsets, const dlscqp_lsc_row *rows, const float *sfc_boxes,
                                 int32_t max_slack_obs, float *out_goal, int32_t *out_goal_status, float *out_ctrl,
                                 double *out_cost, double *out_slack_cost, int32_t *out_status) {
    return plan_host(ctx, n_agents, agents, row_offsets, rows, sfc_boxes, max_slack_obs, 1, 1, out_goal,
                     out_goal_status, out_ctrl, out_cost, out_slack_cost, out_status);
}
