// emul.cpp -- TEST INFRASTRUCTURE: lane-serial host build of the solver kernel's source.
//
// Compiles dlsc_gc_planner_amd/csrc/gi_core.hpp (the code the gfx950 kernel runs) with a "wave"
// whose lanes are executed one after another on the CPU, so that the CPU-only test tier
// (pytest -m "not gpu", no GPU in the build container) can check the algorithm and the host
// tables against the oracle.  The product library never loads, links or falls back to this.
// EMUL_REVERSE=1 runs every lane loop backwards: results must not change (no cross-lane hazard
// inside a lane loop).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#define DQ_FN inline
#include "../../dlsc_gc_planner_amd/csrc/gi_core.hpp"
#include "../../dlsc_gc_planner_amd/csrc/lsc_core.hpp"
#include "../../dlsc_gc_planner_amd/csrc/model_rows.hpp"
#include "../../dlsc_gc_planner_amd/csrc/qp_tables.hpp"
#include "../../dlsc_gc_planner_amd/csrc/sfc_core.hpp"
#include "../../dlsc_gc_planner_amd/csrc/sfc_world.hpp"

namespace {

std::vector<double> *g_trace = nullptr;
std::vector<unsigned char> g_last_lds;  // debug: LDS image of the last agent solved  // -DDLSCQP_TRACE builds: records of the agent being solved

static double *g_conflict = nullptr;  // emul_set_conflict
static int g_conflict_stride = 0;

struct WaveCpu {
    bool reverse;
    static constexpr int nw = 1;  // no helper wavefronts on the host
    void pass_begin(int) const {}
    double pass_end(double top) const { return top; }
    template <class T>
    struct Reg {
        T v[64];
#ifndef EMUL_NO_POISON
        Reg() { std::memset(v, 0xCD, sizeof v); }  // poison: a lane value nobody wrote must not be used
#endif
        T &at(int l) { return v[l]; }
        const T &at(int l) const { return v[l]; }
    };
    template <class F>
    void each(F f) const {
        if (!reverse)
            for (int l = 0; l < 64; l++) f(l);
        else
            for (int l = 63; l >= 0; l--) f(l);
    }
    void sync() const {}
    template <class F>
    void once(F f) const { f(); }  // wave-uniform work on the working set: here it runs once, on the device every lane does the same
    void stamp(int) const {}
    void trace(int tag, double a, double b, double c, double d) const {
        if (g_trace) { g_trace->push_back(tag); g_trace->push_back(a); g_trace->push_back(b); g_trace->push_back(c); g_trace->push_back(d); }
    }
    template <class T>
    T bcast(const Reg<T> &r, int lane) const { return r.v[lane]; }
    double max(const Reg<double> &r) const {
        double s = r.v[0];
        for (int l = 1; l < 64; l++) s = r.v[l] > s ? r.v[l] : s;
        return s;
    }
    double max_pos(const Reg<double> &r) const { return max(r); }  // the GPU's shortcut for values >= +0
    double min(const Reg<double> &r) const {
        double s = r.v[0];
        for (int l = 1; l < 64; l++) s = r.v[l] < s ? r.v[l] : s;
        return s;
    }
    void sum2(const Reg<double> &a, const Reg<double> &b, double &sa, double &sb) const {
        sa = 0; sb = 0;
        for (int l = 0; l < 64; l++) { sa += a.v[l]; sb += b.v[l]; }
    }
    double sum(const Reg<double> &a) const {
        double sa = 0;
        for (int l = 0; l < 64; l++) sa += a.v[l];
        return sa;
    }
    double uni(double v) const { return v; }  // a value every lane computed identically
    uint64_t ballot(const Reg<int> &r) const {
        uint64_t m = 0;
        for (int l = 0; l < 64; l++) if (r.v[l]) m |= 1ull << l;
        return m;
    }
    bool any(const Reg<int> &r) const {
        for (int l = 0; l < 64; l++) if (r.v[l]) return true;
        return false;
    }
    int first_lane(const Reg<int> &r) const {
        for (int l = 0; l < 64; l++) if (r.v[l]) return l;
        return -1;
    }
    // inclusive prefix sum over the elements of a vector spread over `nchunks` registers
    void prefix_sum(Reg<double> *arr, int nchunks) const {
        double run = 0;
        for (int c = 0; c < nchunks; c++)
            for (int l = 0; l < 64; l++) { run += arr[c].v[l]; arr[c].v[l] = run; }
    }
    // out[l] = element (l + 64 c) - 1 (0 for the very first element)
    template <class T>
    void shift_up(const Reg<T> *arr, int c, int nchunks, Reg<T> &out) const {
        (void)nchunks;
        for (int l = 63; l >= 1; l--) out.v[l] = arr[c].v[l - 1];
        out.v[0] = c > 0 ? arr[c - 1].v[63] : T(0);
    }
    // out[l] = element (l + 64 c) + 1 of the vector spread over `nchunks` registers
    template <class T>
    void shift_down(const Reg<T> *arr, int c, int nchunks, Reg<T> &out) const {
        for (int l = 0; l < 63; l++) out.v[l] = arr[c].v[l + 1];
        out.v[63] = c + 1 < nchunks ? arr[c + 1].v[0] : arr[c].v[63];
    }
};

}  // namespace

static void dlscqp_rows_to_soa_host(const dlscqp_lsc_row *rows, int64_t n, float *nx, float *ny, float *nz, float *px, float *py,
                                    float *pz, double *d, uint32_t *key) {
    for (int64_t i = 0; i < n; i++) {
        nx[i] = rows[i].normal[0]; ny[i] = rows[i].normal[1]; nz[i] = rows[i].normal[2];
        px[i] = rows[i].point[0]; py[i] = rows[i].point[1]; pz[i] = rows[i].point[2];
        d[i] = rows[i].d;
        key[i] = DLSCQP_KEY(rows[i].m, rows[i].i, rows[i].slack, rows[i].flags & DLSCQP_ROW_DYNAMIC);
    }
}

extern "C" int emul_plan_batch(const dlscqp_config *cfg, int n_agents, const dlscqp_agent *agents,
                               const int64_t *row_offsets, const dlscqp_lsc_row *rows, const float *sfc,
                               int max_slack_obs, int do_goal, int do_traj, float *out_goal,
                               int *out_goal_status, float *out_ctrl, double *out_x, double *out_cost,
                               double *out_slack_cost, int *out_status, int *out_iters) {
    using namespace dlscqp;
    HostTables tb;
    std::string err;
    if (!build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err)) return -1;
    KParams kp = tb.kp;
    kp.tabG = tb.tabG.data();
    kp.tabN = tb.tabN.data();
    kp.tabX0s = tb.tabX0s.data();
    kp.tabX0g = tb.tabX0g.data();
    kp.tabNN = tb.tabNN.data();
    kp.tabSIW = tb.tabSIW.data();
    kp.tabKIW = tb.tabKIW.data();
    kp.tabPhi = tb.tabPhi.data();
    kp.slots = tb.slots.data();
    const int64_t nrows = row_offsets[n_agents];
    std::vector<float> nx(nrows), ny(nrows), nz(nrows), px(nrows), py(nrows), pz(nrows);
    std::vector<double> d(nrows);
    std::vector<uint32_t> key(nrows);
    int kmax = 0;
    for (int a = 0; a < n_agents; a++) kmax = std::max<int>(kmax, (int)(row_offsets[a + 1] - row_offsets[a]));
    // test hook: a caller that understates dlscqp_device_batch.max_rows_per_agent
    if (const char *ov = std::getenv("EMUL_KMAX")) kmax = std::atoi(ov);
    for (int64_t i = 0; i < nrows; i++) {
        nx[i] = rows[i].normal[0]; ny[i] = rows[i].normal[1]; nz[i] = rows[i].normal[2];
        px[i] = rows[i].point[0]; py[i] = rows[i].point[1]; pz[i] = rows[i].point[2];
        d[i] = rows[i].d;
        key[i] = DLSCQP_KEY(rows[i].m, rows[i].i, rows[i].slack, rows[i].flags & DLSCQP_ROW_DYNAMIC);
    }
    RowsView rv{nx.data(), ny.data(), nz.data(), px.data(), py.data(), pz.data(), d.data(), key.data()};
    // same instantiation choice as the launcher in dlscqp.hip (select_instance in gi_core.hpp)
    const InstanceChoice ic = select_instance(kp, max_slack_obs, kmax, std::getenv("EMUL_GENERIC") != nullptr);
    if (ic.id < 0) return -2;
    const int KUsel = ic.KU, SUsel = ic.SU;
    typedef void (*fn_t)(const WaveCpu &, const KParams &, const MemLayout &, unsigned char *, double *, const dlscqp_agent *,
                         const RowsView &, int, int, const float *, int, int, int, float *, int *, float *, double *,
                         double *, double *, int *, int *, int *, double *);
    fn_t fn = nullptr;
    switch (ic.id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID: fn = plan_agent<RC, DIMS, FS, MS, KU, SU, SLK, WaveCpu>; break;
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    }
    // rows beyond the register cache in the working set, as the launcher sizes it for a build that runs four agents per CU
    // (EMUL_LDS_BUDGET=0: every such row from "HBM" in every pass)
    const char *bud = std::getenv("EMUL_LDS_BUDGET");
    // column records of the active rows: EMUL_QZ of them in the working set, the rest in a spill record (0: all in the working
    // set); the launcher's default for launches that fill the chip is 16
    const char *qzs = std::getenv("EMUL_QZ");
    MemLayout lay = make_layout(kp, max_slack_obs, kmax, KUsel, SUsel, ic.by_kind, bud ? std::atoi(bud) : 40448, qzs ? std::atoi(qzs) : 16);
    std::vector<unsigned char> mem(lay.bytes + 64 + 4096);
    std::vector<double> spill(lay.spill_doubles + 8);
    WaveCpu w;
    const char *rev = std::getenv("EMUL_REVERSE");
    w.reverse = rev && rev[0] == '1';
    const int M = cfg->M, n1 = cfg->n + 1, dim = cfg->world_dimension;
    const size_t xs = (size_t)dim * M * n1 + (size_t)M * max_slack_obs;
    for (int a = 0; a < n_agents; a++) {
        std::memset(mem.data(), 0xCD, mem.size());  // poison: nothing may rely on zero-initialised LDS
        std::memset(spill.data(), 0xCD, spill.size() * sizeof(double));
        fn(w, kp, lay, mem.data(), lay.spill_doubles ? spill.data() : nullptr, agents + a, rv, (int)row_offsets[a],
                   (int)(row_offsets[a + 1] - row_offsets[a]), sfc ? sfc + (size_t)a * M * 6 : nullptr,
                   max_slack_obs, do_goal, do_traj, out_goal ? out_goal + (size_t)a * 3 : nullptr,
                   out_goal_status ? out_goal_status + a : nullptr,
                   out_ctrl ? out_ctrl + (size_t)a * M * n1 * 3 : nullptr, out_x ? out_x + a * xs : nullptr,
                   out_cost ? out_cost + a : nullptr,
                   out_slack_cost ? out_slack_cost + (size_t)a * max_slack_obs : nullptr,
                   out_status ? out_status + a : nullptr, out_iters ? out_iters + (size_t)a * 4 : nullptr, nullptr,
                   g_conflict ? g_conflict + (size_t)a * g_conflict_stride : nullptr);
        g_last_lds = mem;
    }
    return 0;
}

// dlscqp_device_batch.out_conflict for the following emul_plan_batch calls ([n_agents][stride] doubles, zeroed by the caller; null: off)
extern "C" void emul_set_conflict(double *buf, int stride) {
    g_conflict = buf;
    g_conflict_stride = stride;
}

// the product's host functions of the reference's failure path (csrc/model_rows.hpp), without a device context
static bool emul_model(const dlscqp_config *cfg, dlscqp::HostTables &tb) {
    std::string err;
    return dlscqp::build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err);
}
extern "C" int emul_write_lp(const dlscqp_config *cfg, const dlscqp_agent *agent, const dlscqp_lsc_row *rows, int n_rows, const float *sfc,
                             int max_slack_obs, const float *goal, const char *path) {
    using namespace dlscqp;
    HostTables tb;
    if (!emul_model(cfg, tb)) return -1;
    HostModel md;
    std::string err;
    if (!build_host_model(*cfg, tb.kp, *agent, rows, n_rows, sfc, max_slack_obs, goal, md, err) || !write_lp(md, path, err)) return -2;
    return 0;
}
extern "C" int emul_conflict_decode(const dlscqp_config *cfg, const dlscqp_agent *agent, const dlscqp_lsc_row *rows, int n_rows, const float *sfc,
                                    int max_slack_obs, const float *goal, const double *record, dlscqp_conflict_member *out, int capacity,
                                    int *n_out, int *n_con_out, double *gap_out) {
    using namespace dlscqp;
    HostTables tb;
    if (!emul_model(cfg, tb)) return -1;
    HostModel md;
    std::string err;
    if (!build_host_model(*cfg, tb.kp, *agent, rows, n_rows, sfc, max_slack_obs, goal, md, err)) return -2;
    *n_con_out = (int)md.con.size();
    const int n = (int)record[0];
    *n_out = 0;
    *gap_out = n > 0 ? record[1] : 0.0;
    if (n > capacity) return -3;
    for (int t = 0; t < n; t++) {
        double scale = 1.0;
        if (!conflict_member(*cfg, tb.kp, tb.slots.data(), md, sfc, (uint32_t)record[4 + 2 * t], out[t], scale, err)) return -4;
        out[t].lambda = record[5 + 2 * t] / scale;
    }
    *n_out = n;
    return 0;
}

// Host tables for inspection by the tests: returns f (free dofs per axis), fills sizes.
// debug: trace of ONE agent (the last of the last emul_plan_batch call made with tracing on)
extern "C" int emul_trace(int enable, double *out, int cap) {
    static std::vector<double> buf;
    if (enable == 1) { buf.clear(); g_trace = &buf; return 0; }
    if (enable == 0) { g_trace = nullptr; return 0; }
    int n = (int)std::min<size_t>(buf.size(), (size_t)cap);
    for (int i = 0; i < n; i++) out[i] = buf[i];
    return n;
}

extern "C" int emul_last_lds(unsigned char *out, int cap) {
    int n = (int)std::min<size_t>(g_last_lds.size(), (size_t)cap);
    std::memcpy(out, g_last_lds.data(), n);
    return n;
}

extern "C" int emul_table_info(const dlscqp_config *cfg, int *f, int *n_slots, int *nb, int *lds_bytes,
                               int max_slack_obs, int kmax) {
    using namespace dlscqp;
    HostTables tb;
    std::string err;
    if (!build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err)) return -1;
    *f = tb.kp.f;
    *n_slots = tb.kp.n_slots;
    *nb = tb.kp.nb;
    const InstanceChoice ic = select_instance(tb.kp, max_slack_obs, kmax, false);
    *lds_bytes = ic.id < 0 ? -1 : make_layout(tb.kp, max_slack_obs, kmax, ic.KU, ic.SU, ic.by_kind).bytes;
    return 0;
}

// condition numbers of the reduced Hessians (one per number of terminal segments; study tool of the dependency thresholds)
extern "C" int emul_table_cond(const dlscqp_config *cfg, double *out /*[4][M]: condition, largest, smallest eigenvalue, slack whitening sigma*/) {
    using namespace dlscqp;
    HostTables tb;
    std::string err;
    if (!build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err)) return -1;
    for (int m = 0; m < cfg->M; m++) { out[m] = tb.hr_cond[(size_t)m]; out[cfg->M + m] = tb.hr_lmax[(size_t)m]; out[2 * cfg->M + m] = tb.hr_lmin[(size_t)m]; out[3 * cfg->M + m] = tb.kp.sigma[m]; }
    return 0;
}

// Row N1: the device code of lsc_core.hpp run item by item on the host (same source as dlscqp_lsc_kernel).
static dlscqp::LscGenArgs emul_lsc_args(const dlscqp_config *cfg, const dlscqp_lsc_gen *gen) {
    using namespace dlscqp;
    LscGenArgs g;
    g.n_local = gen->n_local; g.first_plan = gen->first_plan;
    g.M = cfg->M; g.n1 = cfg->n + 1; g.phi = cfg->phi; g.dt = cfg->dt;
    g.local_ids = gen->local_ids;
    g.plans = gen->plans; g.position = gen->position; g.velocity = gen->velocity; g.goals = gen->goals;
    g.radius = gen->radius; g.downwash = gen->downwash;
    g.nbr_offsets = gen->nbr_offsets; g.nbr_ids = gen->nbr_ids; g.nbr_counts = gen->nbr_counts;
    g.nx = gen->nx; g.ny = gen->ny; g.nz = gen->nz; g.px = gen->px; g.py = gen->py; g.pz = gen->pz;
    g.d = gen->d; g.key = gen->key;
    g.n_dyn = gen->n_dyn > 0 ? gen->n_dyn : 0;
    g.dyn_size_prediction = gen->dyn_size_prediction;
    g.dyn_uncertainty_horizon = gen->dyn_uncertainty_horizon;
    g.dyn_position = gen->dyn_position; g.dyn_velocity = gen->dyn_velocity;
    g.dyn_radius = gen->dyn_radius; g.dyn_max_acc = gen->dyn_max_acc; g.dyn_downwash = gen->dyn_downwash;
    g.observed_position = gen->observed_position;
    g.reset_threshold = gen->reset_threshold;
    g.comm_range = cfg->communication_range;
    g.use_sfc = cfg->world_use_octomap ? 1 : 0;
    g.prune_sfc = gen->prune ? (g.use_sfc ? gen->prune_sfc_boxes : nullptr) : nullptr;
    g.row_counts = gen->prune ? gen->out_row_counts : nullptr;
    g.tail_from = gen->tail_from;
    g.tail_segments = gen->prune && gen->tail_segments > 0 ? gen->tail_segments : 0;
    g.dim = cfg->world_dimension;
    for (int k = 0; k < 3; k++) { g.world_min[k] = cfg->world_min[k]; g.world_max[k] = cfg->world_max[k]; }
    return g;
}

// TrajPlanner::checkWaypointTrap with the device helpers of lsc_core.hpp, agents and rows one after another
extern "C" int emul_waypoint_trap(const dlscqp_config *cfg, const dlscqp_lsc_gen *gen, const dlscqp_waypoint_trap *trap) {
    using namespace dlscqp;
    const LscGenArgs g = emul_lsc_args(cfg, gen);
    const int per = g.M * g.n1 - g.phi;
    for (int k = 0; k < g.n_local && g.n_dyn > 0; k++) {
        const int p0 = g.nbr_offsets[k], p1 = g.nbr_counts ? p0 + g.nbr_counts[k] : g.nbr_offsets[k + 1];
        const size_t row0 = (size_t)per * ((size_t)p0 + (size_t)k * g.n_dyn);
        const dlscqp_agent *ag = trap->agents + k;
        const float *cg = ag->current_goal_point, *wp = ag->next_waypoint;
        bool trapped = false;
        const int n_rows = g.row_counts ? g.row_counts[k] : (p1 - p0 + g.n_dyn) * per;
        for (int j = 0; j < n_rows; j++) trapped |= trap_row_excludes(g, row0 + j, cg, wp);
        const float *box = (g.use_sfc && trap->sfc_boxes) ? trap->sfc_boxes + ((size_t)k * g.M + (g.M - 1)) * 6 : nullptr;
        trapped = trapped || !point_in_boxes(g, box, wp, cg) || !point_in_boxes(g, box, wp, wp);
        int released = 0;
        for (int o = 0; trapped && o < g.n_dyn; o++) {
            if (!obstacle_reaches(g.dyn_position + 3 * o, g.dyn_velocity + 3 * o, g.dyn_radius[o], g.dyn_max_acc[o], wp, ag->radius,
                                  cfg->M * cfg->dt, g.dyn_uncertainty_horizon))
                continue;
            released++;
            for (int j = 0; j < n_rows; j++) {
                const uint32_t key = g.key[row0 + j];
                if (((key >> 12) & 1) && (int)(key >> 16) == o) trap_empty_row(g, row0 + j);
            }
        }
        if (trap->out_released) trap->out_released[k] = released;
    }
    return 0;
}

// broadcastMsgs' communication-range filter (multi_sync_simulator.cpp:482-503) with the output layout of
// dlscqp_neighbours_device (lists of fixed pitch, ascending ids, cut to the pitch smallest): the CPU tier's stand-in for the
// device search -- a plain loop over all pairs, no grid.
extern "C" int emul_neighbours(const dlscqp_config *cfg, const dlscqp_neighbour_search *ns) {
    const double range = cfg->communication_range;
    const int per = cfg->M * (cfg->n + 1) - cfg->phi;
    for (int k = 0; k < ns->n_local; k++) {
        const int self = ns->local_ids ? ns->local_ids[k] : k;
        const float *ps = ns->position + (size_t)self * ns->position_stride;
        int count = 0, total = 0;
        for (int q = 0; q < ns->n_total; q++) {
            if (q == self) continue;
            const float *pq = ns->position + (size_t)q * ns->position_stride;
            double dist = 0;  // util.hpp:131-140
            for (int c = 0; c < 3; c++) {
                const double dc = std::abs(ps[c] - pq[c]);
                if (dist < dc) dist = dc;
            }
            if (range > 0 && dist > range) continue;
            total++;
            if (count < ns->pitch) ns->out_nbr_ids[(size_t)k * ns->pitch + count++] = q;
        }
        ns->out_nbr_counts[k] = count;
        if (ns->out_row_counts) ns->out_row_counts[k] = (count + ns->n_dyn) * per;
        if (ns->out_overflow && total > ns->pitch) *ns->out_overflow = 1;
    }
    return 0;
}

extern "C" int emul_generate_lsc(const dlscqp_config *cfg, const dlscqp_lsc_gen *gen) {
    using namespace dlscqp;
    const LscGenArgs g = emul_lsc_args(cfg, gen);
    const int per = g.M * g.n1 - g.phi;
    float hull_pts[18];
    const HullStore hull = {hull_pts, 1};
    for (int k = 0; k < g.n_local; k++) {  // same item -> row mapping as dlscqp_lsc_kernel
        const int self = g.local_ids ? g.local_ids[k] : k;
        const int p0 = g.nbr_offsets[k], p1 = g.nbr_counts ? p0 + g.nbr_counts[k] : g.nbr_offsets[k + 1];
        const size_t pair0 = (size_t)p0 + (size_t)k * g.n_dyn;
        if (g.row_counts) {  // presolve: the rows that are not implied by the agent's box rows, contiguously, in order
            size_t at = (size_t)per * pair0;
            const size_t first = at;
            const int items = (p1 - p0) * g.M, all = items + g.n_dyn * g.M;
            for (int it = 0; it < all; it++) {
                LscRows r;
                int m;
                if (it < items) {
                    const int j = it / g.M;
                    m = it - j * g.M;
                    if (g.tail_segments > 0 && j >= g.tail_from && m < g.M - g.tail_segments) continue;  // dlscqp_lsc_gen.tail_from
                    lsc_item_rows(g, self, g.nbr_ids[p0 + j], m, r, hull);
                } else { const int q = it - items, o = q / g.M; m = q - o * g.M; lsc_dyn_item_rows(g, self, o, m, r); }
                const float *box = g.prune_sfc ? g.prune_sfc + ((size_t)k * g.M + m) * 6 : nullptr;
                bool keep[6];
                const int cnt = lsc_rows_kept(g, box, r, keep);
                lsc_store_rows(g, r, at, keep);
                at += cnt;
            }
            g.row_counts[k] = (int)(at - first);
            continue;
        }
        for (int p = p0; p < p1; p++)
            for (int m = 0; m < g.M; m++) lsc_item(g, self, g.nbr_ids[p], m, (size_t)per * (pair0 + (p - p0)), hull);
        for (int o = 0; o < g.n_dyn; o++)
            for (int m = 0; m < g.M; m++) lsc_dyn_item(g, self, o, m, (size_t)per * (pair0 + (p1 - p0) + o));
    }
    return 0;
}


// Row N2: the device code of sfc_core.hpp with the lanes of a wave run one after another, on the world table that
// the product's host code (sfc_world.cpp) builds.
namespace {
int64_t g_sfc_stats[5];  // box tests point by point, lattice points of those boxes, tests that found an obstacle, agents, passes of 64 trials (developer aid, emul_sfc_stats)
struct WaveSfcCpu {
    template <class F>
    bool any_of(int n, F f) const {
        g_sfc_stats[0]++;
        g_sfc_stats[1] += n;
        for (int i = 0; i < n; i++)
            if (f(i)) { g_sfc_stats[2]++; return true; }
        return false;
    }
    void load6(const float *src, float *dst) const { for (int k = 0; k < 6; k++) dst[k] = src[k]; }
    void store6(float *dst, const float *src) const { for (int k = 0; k < 6; k++) dst[k] = src[k]; }
    void fence() const {}
    int uniform_max8(int v) const { return v; }  // one point at a time on the host
    mutable uint32_t axis[3][dlscqp::SFC_AXIS_CAP];
    template <class F>
    void lanes(F f, uint64_t &m0, uint64_t &m1) const {
        m0 = m1 = 0;
        for (int l = 0; l < 64; l++) {
            const int fl = f(l);
            m0 |= (uint64_t)(fl & 1) << l;
            m1 |= (uint64_t)((fl >> 1) & 1) << l;
        }
        g_sfc_stats[4]++;
    }
    mutable float chainv[6][dlscqp::SFC_CHAIN_CAP];
    mutable int chain_n = 1;
    void chain_fill(const float *init, double res, int n) const {
        for (int d = 0; d < 6; d++) {
            float v = init[d];
            for (int c = 0; c < n; c++) {
                chainv[d][c] = v;
                v = (float)((double)v + (d < 3 ? -res : res));
            }
        }
        chain_n = n;
    }
    float chain(int d, int c) const { return chainv[d][c < 0 ? 0 : (c >= chain_n ? chain_n - 1 : c)]; }
    mutable float boxes[2][dlscqp::SFC_BOXES_CAP][6];
    void boxes_load(const float *sfcs, int M) const { std::memcpy(boxes[0], sfcs, sizeof(float) * 6 * M); }
    void box(int m, float *out) const { for (int k = 0; k < 6; k++) out[k] = boxes[0][m][k]; }
    void box_keep(int m, const float *in) const { for (int k = 0; k < 6; k++) boxes[1][m][k] = in[k]; }
    void boxes_store(float *sfcs, int M) const { std::memcpy(sfcs, boxes[1], sizeof(float) * 6 * M); }
    float rcp(float x) const { return 1.0f / x; }
    int uniform(int v) const { return v; }
    float uniform_f(float v) const { return v; }
    void axis_fill(const dlscqp::SfcWorldView &g, double inv_res, const float *b) const {
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < dlscqp::SFC_AXIS_CAP; l++) axis[k][l] = dlscqp::sfc_axis_entry(g, inv_res, k, b[k], l);
    }
    uint32_t axis_entry(int k, int it) const { return axis[k][it]; }
    bool corner_blocked(const dlscqp::SfcWorldView &g, int ic, int jc, int kc, int corner, int gap_max) const {
        const uint32_t word = g.gaps[((size_t)ic * g.n[1] + jc) * g.n[2] + kc];
        return (int)((word >> (4 * corner)) & 15u) <= gap_max;
    }
};
}  // namespace

extern "C" void emul_sfc_stats(int64_t *out, int reset) {
    for (int k = 0; k < 5; k++) { out[k] = g_sfc_stats[k]; if (reset) g_sfc_stats[k] = 0; }
}

extern "C" int emul_generate_sfc(const dlscqp_config *cfg, const double *boxes, int n_boxes, double resolution,
                                 const float *world_min, const float *world_max, const dlscqp_sfc_gen *gen) {
    using namespace dlscqp;
    SfcWorldHost hw;
    build_sfc_world(boxes, n_boxes, resolution, world_min, world_max, hw);
    SfcWorldView v;
    v.res = hw.res;
    for (int k = 0; k < 3; k++) { v.wmin[k] = hw.wmin[k]; v.wmax[k] = hw.wmax[k]; v.lo[k] = hw.lo[k]; v.n[k] = hw.n[k]; }
    v.near = hw.near_packed.data();
    v.gaps = std::getenv("EMUL_SFC_RECORDS") || hw.cells() >= (size_t)SFC_GAPS_MAX_CELLS ? nullptr : hw.gaps.data();  // the lattice-point table, or the 32-byte records as in round 3
    // the summed-volume tables the agents' radii ask for (the product builds them the same way on the first launch); EMUL_SFC_SVT=0: none
    std::vector<int32_t> svt[15];
    const int32_t *svt_ptrs[15] = {};
    v.svt = nullptr;
    const char *svt_env = std::getenv("EMUL_SFC_SVT");
    if (v.gaps && !(svt_env && svt_env[0] == '0')) {
        for (int a = 0; a < gen->n_agents; a++) {
            const int gm = sfc_gap_max(gen->radius[a], hw.res);
            if (gm < 0 || gm > 14 || svt_ptrs[gm]) continue;
            build_sfc_svt(hw, gm, svt[gm]);
            svt_ptrs[gm] = svt[gm].data();
        }
        v.svt = svt_ptrs;
    }
    WaveSfcCpu w;
    const int M = cfg->M, n1 = cfg->n + 1;
    for (int a = 0; a < gen->n_agents; a++) {
        SfcParams p{gen->radius[a], gen->grid_resolution, gen->max_vel[a], cfg->dt, 0.0, 0, nullptr};
        sfc_prepare(v, p);
        float *sfcs = gen->sfcs + (size_t)a * M * 6;
        int code;
        if (gen->initialize) code = sfc_initialize(w, v, p, gen->position + 3 * (size_t)a, M, sfcs) ? 0 : -1;
        else code = sfc_construct(w, v, p, gen->initial_traj + (size_t)a * M * n1 * 3, M, n1, gen->goals + 3 * (size_t)a,
                                  gen->waypoints + 3 * (size_t)a, sfcs);
        if (gen->out_code) gen->out_code[a] = code;
        g_sfc_stats[3]++;
    }
    return 0;
}


// CPU baseline of bench.py (cpu_baseline.kind = "port"): the kernel's own algorithm -- this lane-serial build of gi_core.hpp,
// compiled -O3 -march=native without the test build's poisoning (libdlscqp_cpu.so, see Makefile) -- on `n_threads` native
// threads, each walking its own contiguous range of the batch again and again until `seconds` have passed.  Returns the
// number of agent-QPs (goal LP + trajectory QP) planned and the elapsed time.
extern "C" int emul_throughput(const dlscqp_config *cfg, int n_agents, const dlscqp_agent *agents, const int64_t *row_offsets,
                               const dlscqp_lsc_row *rows, const float *sfc, int max_slack_obs, int n_threads, double seconds,
                               int64_t *n_done, double *elapsed_s) {
    using namespace dlscqp;
    HostTables tb;
    std::string err;
    if (!build_tables(*cfg, cfg->world_dimension, cfg->M, cfg->n, cfg->phi, cfg->dt, tb, err)) return -1;
    KParams kp = tb.kp;
    kp.tabN = tb.tabN.data();
    kp.tabG = tb.tabG.data(); kp.tabX0s = tb.tabX0s.data(); kp.tabX0g = tb.tabX0g.data();
    kp.tabNN = tb.tabNN.data(); kp.tabSIW = tb.tabSIW.data(); kp.tabKIW = tb.tabKIW.data(); kp.tabPhi = tb.tabPhi.data(); kp.slots = tb.slots.data();
    const int64_t nrows = row_offsets[n_agents];
    std::vector<float> nx(nrows), ny(nrows), nz(nrows), px(nrows), py(nrows), pz(nrows);
    std::vector<double> d(nrows);
    std::vector<uint32_t> key(nrows);
    int kmax = 0;
    for (int a = 0; a < n_agents; a++) kmax = std::max<int>(kmax, (int)(row_offsets[a + 1] - row_offsets[a]));
    dlscqp_rows_to_soa_host(rows, nrows, nx.data(), ny.data(), nz.data(), px.data(), py.data(), pz.data(), d.data(), key.data());
    RowsView rv{nx.data(), ny.data(), nz.data(), px.data(), py.data(), pz.data(), d.data(), key.data()};
    const InstanceChoice ic = select_instance(kp, max_slack_obs, kmax, false);
    if (ic.id < 0) return -2;
    typedef void (*fn_t)(const WaveCpu &, const KParams &, const MemLayout &, unsigned char *, double *, const dlscqp_agent *,
                         const RowsView &, int, int, const float *, int, int, int, float *, int *, float *, double *,
                         double *, double *, int *, int *, int *, double *);
    fn_t fn = nullptr;
    switch (ic.id) {
#define DQ_CASE(ID, RC, DIMS, FS, MS, KU, SU, SLK) \
    case ID: fn = plan_agent<RC, DIMS, FS, MS, KU, SU, SLK, WaveCpu>; break;
        DQ_INSTANCES(DQ_CASE)
#undef DQ_CASE
    }
    const MemLayout lay = make_layout(kp, max_slack_obs, kmax, ic.KU, ic.SU, ic.by_kind);
    const int M = cfg->M, n1 = cfg->n + 1;
    if (n_threads < 1) n_threads = 1;
    std::vector<int64_t> done(n_threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int t) {
        std::vector<unsigned char> mem(lay.bytes + 64 + 4096);
        std::vector<float> ctrl((size_t)M * n1 * 3);
        std::vector<double> slack(std::max(max_slack_obs, 1));
        float goal[3];
        int gst, st, it[4];
        double cost;
        WaveCpu w;
        w.reverse = false;
        const int lo = (int)((int64_t)n_agents * t / n_threads), hi = (int)((int64_t)n_agents * (t + 1) / n_threads);
        if (hi <= lo) return;
        int a = lo;
        for (;;) {
            fn(w, kp, lay, mem.data(), nullptr, agents + a, rv, (int)row_offsets[a], (int)(row_offsets[a + 1] - row_offsets[a]),
               sfc ? sfc + (size_t)a * M * 6 : nullptr, max_slack_obs, 1, 1, goal, &gst, ctrl.data(), nullptr, &cost,
               max_slack_obs ? slack.data() : nullptr, &st, it, nullptr, nullptr);
            done[t]++;
            if (++a == hi) a = lo;
            if ((done[t] & 15) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= seconds) break;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t tot = 0;
    for (auto v : done) tot += v;
    *n_done = tot;
    return 0;
}
