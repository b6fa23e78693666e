// matp.hpp -- host-side C++ mirror of the reference's optimizer interface over the C ABI.
//
// The reference's "plugin API" for the hot path is a set of C++ classes in namespace MATP:
//   MATP::TrajOptimizer        /root/reference/include/traj_optimizer.hpp:24-59
//   MATP::GoalOptimizer        include/goal_optimizer.hpp:17-34
//   MATP::CollisionConstraints include/collision_constraints.hpp:116-171 (the accessors the
//                              optimizers read: getLSC/getSFC/getObsSize/isDynamicObstacle/
//                              getObsPosition, and the setters that fill them)
// whose headers pull in ROS, octomap, Eigen and CPLEX.  This header re-declares exactly the types
// that cross that boundary with the same names, members and error behaviour, ROS/octomap/Eigen
// free, and implements solve() by marshalling into include/dlscqp.h (the HIP library).
// A maintainer of the reference swaps `#include <traj_optimizer.hpp>` / `<goal_optimizer.hpp>`
// for this header and drops the CPLEX link line (INTEGRATION.md shows the diff); with the real
// octomap::point3d available, define MATP_HAVE_OCTOMAP before including to reuse it.
//
// Added entry point (not in the reference): TrajOptimizer::solveBatch plans every agent of a
// replanning step with ONE kernel launch -- the restructured MultiSyncSimulator::plan loop
// (multi_sync_simulator.cpp:516-524).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

// the C ABI: -I<repo>/include (or wherever dlscqp.h is installed); the relative path is the in-tree fallback
#if defined(__has_include) && __has_include(<dlscqp.h>)
#include <dlscqp.h>
#else
#include "../../../include/dlscqp.h"
#endif

#define SP_EPSILON 1e-9        // sp_const.hpp:3
#define SP_EPSILON_FLOAT 1e-5  // sp_const.hpp:4
#define SP_INFINITY 1e+9       // sp_const.hpp:5

namespace MATP {

#ifdef MATP_HAVE_OCTOMAP
typedef octomap::point3d point3d;
#else
// octomath::Vector3: three float32 with float arithmetic (sp_const.hpp:15)
class point3d {
public:
    point3d() : data{0, 0, 0} {}
    point3d(float x, float y, float z) : data{x, y, z} {}
    float &x() { return data[0]; }
    float &y() { return data[1]; }
    float &z() { return data[2]; }
    const float &x() const { return data[0]; }
    const float &y() const { return data[1]; }
    const float &z() const { return data[2]; }
    float &operator()(unsigned i) { return data[i]; }
    const float &operator()(unsigned i) const { return data[i]; }
    point3d operator+(const point3d &o) const { return point3d(data[0] + o.data[0], data[1] + o.data[1], data[2] + o.data[2]); }
    point3d operator-(const point3d &o) const { return point3d(data[0] - o.data[0], data[1] - o.data[1], data[2] - o.data[2]); }
    point3d operator*(float s) const { return point3d(data[0] * s, data[1] * s, data[2] * s); }
    double dot(const point3d &o) const { return data[0] * o.data[0] + data[1] * o.data[1] + data[2] * o.data[2]; }
    double norm_sq() const { return data[0] * data[0] + data[1] * data[1] + data[2] * data[2]; }
    double norm() const { return std::sqrt(norm_sq()); }
    double distance(const point3d &o) const { return (*this - o).norm(); }

private:
    float data[3];
};
#endif
typedef std::vector<point3d> point3ds;

// sp_const.hpp:18-79
enum class PlannerMode { DLSCGC, DLSC, LSC, BVC, ORCA, RECIPROCALRSFC, CIRCLETEST };
enum class SlackMode { NONE, CONTINUITY, COLLISIONCONSTRAINT };
enum PlanningReport { Initialized, INITTRAJGENERATIONFAILED, CONSTRAINTGENERATIONFAILED, QPFAILED, WAITFORROSMSG, SUCCESS };
enum ObstacleType { DEFAULT, AGENT, DYN_SPIN, DYN_STRAIGHT, DYN_PATROL, DYN_CHASING, DYN_GAUSSIAN, DYN_REAL };
typedef std::vector<ObstacleType> ObstacleTypes;

struct State {  // sp_const.hpp:141-145
    point3d position, velocity, acceleration;
};

struct Agent {  // sp_const.hpp:147-160
    int id = 0;
    int cid = 0;
    State current_state;
    point3d start_point, desired_goal_point, current_goal_point, next_waypoint;
    double max_vel = 0, max_acc = 0, radius = 0, downwash = 0, nominal_velocity = 0;
};
typedef std::vector<Agent> Agents;

// The members of MATP::Param (param.hpp) that the optimizers read.
struct Param {
    bool log_solver = false;
    std::string package_path;
    int world_dimension = 3;
    bool world_use_octomap = false;
    double world_z_2d = 1.0;
    PlannerMode planner_mode = PlannerMode::DLSCGC;
    SlackMode slack_mode = SlackMode::NONE;
    double dt = 0.2;
    int M = 5, n = 5, phi = 3, phi_n = 1;
    double control_input_weight = 1, terminal_weight = 1, slack_collision_weight = 1;
    double slack_threshold = 0.001;
    double communication_range = 3.0;
    double goal_threshold = 0.1;
    int device = 0;  // added: HIP device ordinal
};

// The members of MATP::Mission (mission.hpp:23-36) that the optimizers read.
struct Mission {
    point3d world_min, world_max;
};

// trajectory.hpp:9-68 (control-point container; evaluation lives in the caller)
template <typename T>
class Segment {
public:
    std::vector<T> control_points;
    double segment_time = 0;
    T operator[](int idx) const { return control_points[idx]; }
    T &operator[](int idx) { return control_points[idx]; }
};

template <typename T>
class Trajectory {
public:
    Trajectory() : M(0), n(0) {}
    Trajectory(size_t _M, size_t _n, double dt) : M(_M), n(_n), segments(_M) {
        for (auto &s : segments) {
            s.control_points.resize(n + 1);
            s.segment_time = dt;
        }
    }
    int size() const { return (int)segments.size(); }
    bool empty() const { return segments.empty(); }
    T startPoint() const { return segments[0][0]; }
    T lastPoint() const { return segments[M - 1][n]; }
    Segment<T> operator[](int idx) const { return segments[idx]; }
    Segment<T> &operator[](int idx) { return segments[idx]; }

private:
    size_t M, n;
    std::vector<Segment<T>> segments;
};
typedef Trajectory<point3d> traj_t;

struct Obstacle {  // obstacle.hpp:12-38 (members used on this path)
    ObstacleType type = ObstacleType::DEFAULT;
    int id = -1;
    double radius = 0, downwash = 0, max_acc = 0;
    point3d position, velocity, goal_point;
    traj_t prev_traj;
};
typedef std::vector<Obstacle> Obstacles;

struct CollisionAlert {  // obstacle.hpp:40-45
    point3d agent_position;
    Obstacles obstacles;
    void initialize() { obstacles.clear(); }
    bool activated() const { return !obstacles.empty(); }
};

// collision_constraints.hpp:19-35
class LSC {
public:
    LSC() = default;
    LSC(const point3d &_obs_control_point, const point3d &_normal_vector, double _d)
        : obs_control_point(_obs_control_point), normal_vector(_normal_vector), d(_d) {}
    bool isPointInLSC(const point3d &point) const {  // collision_constraints.cpp:35-37
        return (point - obs_control_point).dot(normal_vector) - d > -SP_EPSILON;
    }
    point3d obs_control_point;
    point3d normal_vector;
    double d = 0;
};
typedef std::vector<LSC> LSCs;

// collision_constraints.hpp:53-111 (the parts used on this path)
class Box {
public:
    point3d box_min, box_max;
    Box() = default;
    Box(const point3d &_box_min, const point3d &_box_max) : box_min(_box_min), box_max(_box_max) {}
    LSCs convertToLSCs(int dim) const {  // collision_constraints.cpp:66-87
        LSCs lscs(2 * dim);
        for (int i = 0; i < dim; i++) {
            point3d nmin(0, 0, 0), nmax(0, 0, 0);
            nmin(i) = 1;
            nmax(i) = -1;
            lscs[2 * i] = LSC(point3d(0, 0, 0), nmin, box_min(i));
            lscs[2 * i + 1] = LSC(point3d(0, 0, 0), nmax, -box_max(i));
        }
        return lscs;
    }
    bool isPointInBox(const point3d &p) const {  // collision_constraints.cpp:109-116
        return p.x() > box_min.x() - SP_EPSILON_FLOAT && p.y() > box_min.y() - SP_EPSILON_FLOAT &&
               p.z() > box_min.z() - SP_EPSILON_FLOAT && p.x() < box_max.x() + SP_EPSILON_FLOAT &&
               p.y() < box_max.y() + SP_EPSILON_FLOAT && p.z() < box_max.z() + SP_EPSILON_FLOAT;
    }
};
typedef std::vector<Box> SFCs;

// collision_constraints.hpp:116-171: the container the optimizers read.  Producers of its content
// (SFC growing on the octomap, DLSC-GC half-space generation) stay in the caller.
class CollisionConstraints {
public:
    CollisionConstraints(const Param &_param, const Mission &_mission, double _agent_radius = 0, double _agent_max_vel = 0)
        : mission(_mission), param(_param), agent_radius(_agent_radius), agent_max_vel(_agent_max_vel) {
        sfcs.resize(param.M);
    }
    void initializeLSC(const Obstacles &obstacles) {  // collision_constraints.cpp:454-461
        N_obs = (int)obstacles.size();
        types.assign(N_obs, ObstacleType::DEFAULT);
        obs_positions.assign(N_obs, point3d());
        constraints.assign(N_obs, std::vector<std::vector<LSC>>(param.M, std::vector<LSC>(param.n + 1)));
        for (int oi = 0; oi < N_obs; oi++) {
            types[oi] = obstacles[oi].type;
            obs_positions[oi] = obstacles[oi].position;
        }
    }
    bool isDynamicObstacle(int oi) const { return types[oi] != ObstacleType::AGENT; }  // :582-584
    LSC getLSC(int oi, int m, int i) const { return constraints[oi][m][i]; }           // :600-602
    Box getSFC(int m) const { return sfcs[m]; }                                        // :604-606
    size_t getObsSize() const { return (size_t)N_obs; }                                // :608-610
    point3d getObsPosition(int oi) const { return obs_positions[oi]; }                 // :612-614
    void setLSC(int oi, int m, int i, const LSC &lsc) { constraints[oi][m][i] = lsc; } // :624-626
    void setSFC(int m, const Box &sfc) { sfcs[m] = sfc; }                              // :628-630

private:
    Mission mission;
    Param param;
    double agent_radius, agent_max_vel;
    int N_obs = 0;
    ObstacleTypes types;
    point3ds obs_positions;
    std::vector<std::vector<std::vector<LSC>>> constraints;  // [obs_idx][segment_idx][control_pts_idx]
    SFCs sfcs;
};

struct TrajOptResult {  // traj_optimizer.hpp:18-22
    traj_t desired_traj;
    double total_qp_cost = 0;
    CollisionAlert collision_alert;
};

namespace detail {

inline dlscqp_config make_config(const Param &p, const Mission &m) {
    dlscqp_config c{};
    c.abi_version = DLSCQP_ABI_VERSION;
    c.world_dimension = p.world_dimension;
    c.M = p.M; c.n = p.n; c.phi = p.phi; c.phi_n = p.phi_n;
    c.planner_mode = (int)p.planner_mode;
    c.slack_mode = (int)p.slack_mode;
    c.world_use_octomap = p.world_use_octomap ? 1 : 0;
    c.device = p.device;
    c.dt = p.dt;
    c.control_input_weight = p.control_input_weight;
    c.terminal_weight = p.terminal_weight;
    c.slack_collision_weight = p.slack_collision_weight;
    c.communication_range = p.communication_range;
    c.world_z_2d = p.world_z_2d;
    c.slack_threshold = p.slack_threshold;
    for (int k = 0; k < 3; k++) {
        c.world_min[k] = m.world_min(k);  // float -> double, like mission.world_min(k) in the model
        c.world_max[k] = m.world_max(k);
    }
    return c;
}

inline void put(float dst[3], const point3d &p) { dst[0] = p.x(); dst[1] = p.y(); dst[2] = p.z(); }

// One device context per distinct configuration, shared by every optimizer object of the process.  The reference
// constructs a TEMPORARY GoalOptimizer per agent per replanning step (traj_planner.cpp:493-495) and one TrajOptimizer per
// agent (:29); a context (host tables, null-space QR, M Cholesky factors, device uploads, a stream) costs milliseconds to
// build, so constructing an optimizer must not build one: it looks its configuration up here (a memcmp of 144 bytes under
// a mutex).  Contexts live until the process ends (never destroyed: no ordering problem with the HIP runtime's own
// teardown).  The library's entry points are synchronous on the context's stream, so sharing is safe for the reference's
// single caller thread; concurrent callers are serialised by `use`.
struct SharedContext {
    dlscqp_config cfg;
    dlscqp_ctx *ctx = nullptr;
    std::mutex use;
};
inline std::shared_ptr<SharedContext> acquire_context(const dlscqp_config &c) {
    static std::mutex mu;
    static std::vector<std::shared_ptr<SharedContext>> *cache = new std::vector<std::shared_ptr<SharedContext>>();
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : *cache)
        if (std::memcmp(&e->cfg, &c, sizeof c) == 0) return e;
    auto e = std::make_shared<SharedContext>();
    std::memset(&e->cfg, 0, sizeof e->cfg);
    e->cfg = c;
    const int rc = dlscqp_create(&c, &e->ctx);
    if (rc == DLSCQP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(dlscqp_last_error());  // traj_optimizer.cpp:209, :248
    if (rc != DLSCQP_SUCCESS) throw std::runtime_error(std::string("[dlscqp] ") + dlscqp_last_error());
    cache->push_back(e);
    return e;
}

// Flatten one agent's constraints exactly in the order populatebyrow visits them
// (traj_optimizer.cpp:413-450): obstacle, segment, control point.
inline void marshal(const Param &param, const Agent &agent, const CollisionConstraints &cons, dlscqp_agent &ag,
                    std::vector<dlscqp_lsc_row> &rows, std::vector<float> &sfc) {
    put(ag.position, agent.current_state.position);
    put(ag.velocity, agent.current_state.velocity);
    put(ag.acceleration, agent.current_state.acceleration);
    put(ag.current_goal_point, agent.current_goal_point);
    put(ag.next_waypoint, agent.next_waypoint);
    ag.id = agent.id;
    ag.max_vel = agent.max_vel; ag.max_acc = agent.max_acc; ag.radius = agent.radius;
    ag.nominal_velocity = agent.nominal_velocity;
    const int N_obs = (int)cons.getObsSize();
    ag.n_obs = N_obs;
    int obs_slack_idx = 0;
    for (int oi = 0; oi < N_obs; oi++) {
        const bool dyn = cons.isDynamicObstacle(oi);
        const bool has_slack = param.slack_mode == SlackMode::COLLISIONCONSTRAINT || dyn;  // :274, :436
        for (int m = 0; m < param.M; m++)
            for (int i = 0; i < param.n + 1; i++) {
                if (m == 0 && i < param.phi) continue;  // :417-419, never part of the model
                LSC lsc = cons.getLSC(oi, m, i);
                dlscqp_lsc_row r{};
                put(r.normal, lsc.normal_vector);
                put(r.point, lsc.obs_control_point);
                r.d = lsc.d;
                r.m = (int16_t)m; r.i = (int8_t)i; r.obs = (int16_t)oi;
                r.slack = has_slack ? (int16_t)obs_slack_idx : (int16_t)-1;
                r.flags = dyn ? DLSCQP_ROW_DYNAMIC : 0;
                rows.push_back(r);
            }
        if (has_slack) obs_slack_idx++;
    }
    ag.n_slack_obs = obs_slack_idx;
    if (param.world_use_octomap)
        for (int m = 0; m < param.M; m++) {
            Box b = cons.getSFC(m);
            for (int k = 0; k < 3; k++) sfc.push_back(b.box_min(k));
            for (int k = 0; k < 3; k++) sfc.push_back(b.box_max(k));
        }
}

}  // namespace detail

class TrajOptimizer {
public:
    // B (Bernstein basis, polynomial.hpp:280-293) is accepted for signature compatibility; it is
    // a function of n only and is rebuilt inside the library.
    template <class MatrixB>
    TrajOptimizer(const Param &_param, const Mission &_mission, const MatrixB &) : param(_param), mission(_mission) { create(); }
    TrajOptimizer(const Param &_param, const Mission &_mission) : param(_param), mission(_mission) { create(); }

    // traj_optimizer.cpp:18-165.  use_primal_algorithm selects a CPLEX algorithm in the reference; the
    // optimum is unique, so it has no effect here.  Throws PlanningReport::QPFAILED on failure (:152).
    TrajOptResult solve(const Agent &agent, const CollisionConstraints &constraints, const traj_t &initial_traj,
                        bool use_primal_algorithm) {
        (void)initial_traj; (void)use_primal_algorithm;
        std::vector<const Agent *> a{&agent};
        std::vector<const CollisionConstraints *> c{&constraints};
        std::vector<int> status;
        std::vector<TrajOptResult> r = solveBatch(a, c, status);
        if (status[0] != DLSCQP_OK) throw PlanningReport::QPFAILED;
        return r[0];
    }

    // One launch for all agents of a replanning step; status[a] != 0 <=> the reference would have thrown
    // QPFAILED for agent a (the caller applies its fail-safe, traj_planner.cpp:749-777).
    std::vector<TrajOptResult> solveBatch(const std::vector<const Agent *> &agents,
                                          const std::vector<const CollisionConstraints *> &constraints,
                                          std::vector<int> &status) {
        const int N = (int)agents.size();
        std::vector<dlscqp_agent> ag(N);
        std::vector<dlscqp_lsc_row> rows;
        std::vector<int64_t> offs(N + 1, 0);
        std::vector<float> sfc;
        int max_slack = 0;
        for (int a = 0; a < N; a++) {
            detail::marshal(param, *agents[a], *constraints[a], ag[a], rows, sfc);
            offs[a + 1] = (int64_t)rows.size();
            if (ag[a].n_slack_obs > max_slack) max_slack = ag[a].n_slack_obs;
        }
        const int M = M_cached, n1 = n_cached + 1;
        std::vector<float> ctrl((size_t)N * M * n1 * 3);
        std::vector<double> cost(N), slack((size_t)N * (max_slack > 0 ? max_slack : 1));
        std::vector<int32_t> st(N, DLSCQP_NUMERICAL);
        int rc;
        {
            std::lock_guard<std::mutex> lock(shared->use);
            rc = dlscqp_solve_batch(shared->ctx, N, ag.data(), offs.data(), rows.data(), sfc.empty() ? nullptr : sfc.data(),
                                    max_slack, ctrl.data(), cost.data(), max_slack > 0 ? slack.data() : nullptr, st.data());
        }
        if (rc != DLSCQP_SUCCESS) throw std::runtime_error(std::string("[TrajOptimizer] ") + dlscqp_last_error());
        std::vector<TrajOptResult> out(N);
        status.assign(st.begin(), st.end());
        // the reference's catch block (traj_optimizer.cpp:112-146): the model of a failed agent as an .lp file and the rows that prove it
        // infeasible ("Conflict :" ... "Proved  : c<i> ...") -- here from the solver's own certificate, not from a second search
        for (int a = 0; a < N; a++)
            if (st[a] != DLSCQP_OK)
                reportFailure(*agents[a], ag[a], rows.data() + offs[a], (int)(offs[a + 1] - offs[a]),
                              sfc.empty() ? nullptr : sfc.data() + (size_t)a * M * 6, max_slack, st[a]);
        for (int a = 0; a < N; a++) {
            TrajOptResult &r = out[a];
            r.desired_traj = traj_t(M, n_cached, dt_cached);
            if (st[a] != DLSCQP_OK) continue;
            for (int m = 0; m < M; m++)
                for (int i = 0; i < n1; i++) {
                    const float *p = &ctrl[(((size_t)a * M + m) * n1 + i) * 3];
                    r.desired_traj[m][i] = point3d(p[0], p[1], p[2]);
                }
            r.total_qp_cost = cost[a];
            // collision alert, traj_optimizer.cpp:85-106
            int obs_slack_idx = 0;
            const CollisionConstraints &c = *constraints[a];
            for (size_t oi = 0; oi < c.getObsSize(); oi++) {
                if (!(param.slack_mode == SlackMode::COLLISIONCONSTRAINT || c.isDynamicObstacle((int)oi))) continue;
                double slack_cost = slack[(size_t)a * max_slack + obs_slack_idx];
                obs_slack_idx++;
                if (slack_cost > param.slack_threshold) {
                    r.collision_alert.agent_position = agents[a]->current_state.position;
                    Obstacle o;
                    o.id = (int)oi;
                    o.position = c.getObsPosition((int)oi);
                    r.collision_alert.obstacles.emplace_back(o);
                }
            }
        }
        return out;
    }

    // traj_optimizer.cpp:167-169: param is replaced, the cached M, n, dt are NOT refreshed (:7-11)
    void updateParam(const Param &_param) {
        param = _param;
        dlscqp_config c = detail::make_config(param, mission);
        c.M = M_cached; c.n = n_cached; c.dt = dt_cached; c.world_dimension = dim_cached; c.phi = phi_cached;
        shared = detail::acquire_context(c);
    }

    // Where the last failed agent's conflict went (tests): "Proved  : c<i> ..." lines as the reference's refiner prints them.
    const std::string &lastConflict() const { return last_conflict; }

private:
    void reportFailure(const Agent &agent, const dlscqp_agent &ag, const dlscqp_lsc_row *rows, int n_rows, const float *sfc, int max_slack, int st) {
        const std::string model_path = param.package_path + "/log/QPmodel_trajOpt.lp";  // traj_optimizer.cpp:45-46
        last_conflict.clear();
        {
            std::lock_guard<std::mutex> lock(shared->use);
            (void)dlscqp_write_lp(shared->ctx, &ag, rows, n_rows, sfc, max_slack, nullptr, model_path.c_str());  // cplex.exportModel, :113
        }
        if (st != DLSCQP_INFEASIBLE) {
            std::cerr << "[TrajOptimizer] solver failed at mav " << agent.id << " (status " << st << ")" << std::endl;
            return;
        }
        std::cerr << "[TrajOptimizer] No solution at mav " << agent.id << ", starting Conflict refinement" << std::endl;  // :116-117
        int32_t st1 = DLSCQP_NUMERICAL, n = 0, n_con = 0;
        double gap = 0;
        const int64_t offs[2] = {0, n_rows};
        std::vector<double> rec;
        std::vector<dlscqp_conflict_member> mem;
        int rc;
        {
            std::lock_guard<std::mutex> lock(shared->use);
            const int w = dlscqp_conflict_doubles(shared->ctx, max_slack);
            rec.assign((size_t)(w > 0 ? w : 4), 0.0);
            rc = dlscqp_solve_batch_conflict(shared->ctx, 1, &ag, offs, rows, sfc, max_slack, &st1, rec.data(), (int)rec.size());
            if (rc == DLSCQP_SUCCESS) {
                mem.resize((size_t)(rec[0] > 0 ? rec[0] : 1));
                rc = dlscqp_conflict_decode(shared->ctx, &ag, rows, n_rows, sfc, max_slack, nullptr, rec.data(), mem.data(), (int)mem.size(), &n, &n_con, &gap);
            }
        }
        if (rc != DLSCQP_SUCCESS || n <= 0) {
            std::cerr << "[TrajOptimizer] Conflict could not be refined" << std::endl;  // :141
            return;
        }
        // "Proved  : c<i> <row>" for the members (:133-136), in the reference's variable names; the multipliers beside them
        static const char axis[3] = {'x', 'y', 'z'};
        const int N1 = n_cached + 1, od = M_cached * N1, osl = dim_cached * od;
        auto var_name = [&](int v) {
            std::ostringstream o;
            if (v < osl) { const int k = v / od, p = v - k * od; o << axis[k] << "_" << p / N1 << "_" << p % N1; }
            else o << "epsilon_slack_col_" << (v - osl) / M_cached << "_" << (v - osl) % M_cached;
            return o.str();
        };
        std::ostringstream out;
        out << "Conflict :" << std::endl;
        for (int t = 0; t < n; t++) {
            const dlscqp_conflict_member &m = mem[(size_t)t];
            out << "Proved  : c" << m.ref_index << " ";
            if (m.lo > -1e299 && m.hi < 1e299 && m.kind < DLSCQP_MODEL_VAR_LB) out << m.lo << " <= ";
            for (int e = 0; e < m.n; e++) out << (m.coef[e] >= 0 && e > 0 ? "+ " : "") << m.coef[e] << " * " << var_name(m.var[e]) << " ";
            if (m.kind == DLSCQP_MODEL_VAR_LB || (m.lo > -1e299 && m.hi >= 1e299)) out << ">= " << m.lo;
            else out << "<= " << m.hi;
            out << "   (multiplier " << m.lambda << ")" << std::endl;
        }
        out << "violated by " << gap << " whatever the other variables do" << std::endl;
        last_conflict = out.str();
        std::cout << last_conflict;
        const std::string conflict_path = param.package_path + "/log/conflict_trajOpt.lp";  // cplex.writeConflict, :139
        std::ofstream cf(conflict_path);
        if (cf) cf << "\\ dlsc_gc_planner_amd: the rows that prove mav " << agent.id << " infeasible (numbers: rows of QPmodel_trajOpt.lp)\n" << last_conflict;
    }
    void create() {
        shared = detail::acquire_context(detail::make_config(param, mission));
        M_cached = param.M; n_cached = param.n; dt_cached = param.dt;  // traj_optimizer.cpp:7-11
        dim_cached = param.world_dimension; phi_cached = param.phi;
    }
    Param param;
    Mission mission;
    std::shared_ptr<detail::SharedContext> shared;
    int M_cached = 0, n_cached = 0, dim_cached = 0, phi_cached = 0;
    double dt_cached = 0;
    std::string last_conflict;
};

class GoalOptimizer {
public:
    // cheap: the context of this configuration is shared (detail::acquire_context); the reference constructs one of these per
    // agent per replanning step (traj_planner.cpp:493-495)
    GoalOptimizer(const Param &_param, const Mission &_mission)
        : param(_param), mission(_mission), shared(detail::acquire_context(detail::make_config(_param, _mission))) {}

    // goal_optimizer.cpp:7-136; throws PlanningReport::QPFAILED when the LP is really infeasible (:122).
    point3d solve(const Agent &agent, const CollisionConstraints &constraints, const point3d &current_goal_point,
                  const point3d &next_waypoint) {
        Agent a = agent;
        a.current_goal_point = current_goal_point;
        a.next_waypoint = next_waypoint;
        dlscqp_agent ag{};
        std::vector<dlscqp_lsc_row> rows;
        std::vector<float> sfc;
        detail::marshal(param, a, constraints, ag, rows, sfc);
        int64_t offs[2] = {0, (int64_t)rows.size()};
        float goal[3] = {0, 0, 0};
        int32_t st = DLSCQP_NUMERICAL;
        int rc;
        {
            std::lock_guard<std::mutex> lock(shared->use);
            rc = dlscqp_goal_batch(shared->ctx, 1, &ag, offs, rows.data(), sfc.empty() ? nullptr : sfc.data(), goal, &st);
        }
        if (rc != DLSCQP_SUCCESS) throw std::runtime_error(std::string("[GoalOptimizer] ") + dlscqp_last_error());
        if (st == DLSCQP_INFEASIBLE) throw PlanningReport::QPFAILED;
        return point3d(goal[0], goal[1], goal[2]);
    }

private:
    Param param;
    Mission mission;
    std::shared_ptr<detail::SharedContext> shared;
};

}  // namespace MATP
