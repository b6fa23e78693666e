// grid_mapf.cpp -- SURVEY.md row N4: the reference's decentralised waypoint planner, on the host.
//
// What MultiSyncSimulator::decentralizedMAPP (multi_sync_simulator.cpp:308-466) does once per replanning step, before
// the trajectory optimisation: communication groups, per group GridBasedPlanner::planMAPF (grid_based_planner.cpp:64-92:
// grid map with static and "warning" nodes :94-171, obstacles of interest :196-250, escape goals :253-305, PIBT
// src/mapf/pibt.cpp:13-203, plan bookkeeping :307-353, :383-422) and the rules that decide which agents may move their
// next_waypoint to the next node of their path (:378-453).  It is graph search on a few hundred nodes -- 35 us per group
// in the reference's own log -- and stays on the host; the device path reads its result (next_waypoint) from the agent
// records.  Written from the reference's behaviour, in one translation unit, with flat arrays instead of its Node graph.
//
// Reproducibility: PIBT shuffles an agent's candidate nodes with std::shuffle(..., std::mt19937(0)) (pibt.cpp:159,
// problem.cpp:84).  std::shuffle is not specified bit for bit; the reference was built with the libstdc++ of ROS
// melodic / noetic (GCC 7-9), whose uniform_int_distribution scales and rejects, while GCC 11 draws with Lemire's
// method.  shuffle_gcc9() below restates the old algorithm so that the waypoints are those of the reference's run
// (checked against the waypoints mined from its committed log, tests/test_mapf.py).
#include "grid_mapf.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <unordered_map>
#include <cstring>
#include <functional>
#include <queue>
#include <random>
#include <set>

#include "sfc_world.hpp"

namespace dlscqp {

namespace {

constexpr double EPS_F = 1e-5;  // SP_EPSILON_FLOAT
constexpr double EPS = 1e-9;    // SP_EPSILON
constexpr double INF = 1e9;     // SP_INFINITY
constexpr int MAX_TIMESTEP = 5000;  // mapf/default_params.hpp

struct P3 {  // octomap::point3d: float32 components
    float x = 0, y = 0, z = 0;
    float operator()(int k) const { return k == 0 ? x : (k == 1 ? y : z); }
    float &at(int k) { return k == 0 ? x : (k == 1 ? y : z); }
};
P3 sub(const P3 &a, const P3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
double norm(const P3 &a) { return std::sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z)); }  // Vector3::norm: float sum, double sqrt
double dist(const P3 &a, const P3 &b) { return norm(sub(a, b)); }
double linf(const P3 &a, const P3 &b) {  // util.hpp:131-140
    double d = 0;
    for (int k = 0; k < 3; k++) {
        const double c = std::fabs((double)(a(k) - b(k)));
        if (d < c) d = c;
    }
    return d;
}

// libstdc++ (GCC 7-10) uniform_int_distribution<unsigned long>{0, b}(mt19937): downscaling by rejection
unsigned long uniform_gcc9(std::mt19937 &g, unsigned long b) {
    const unsigned long urngrange = 0xFFFFFFFFul;
    const unsigned long uerange = b + 1;
    const unsigned long scaling = urngrange / uerange;
    const unsigned long past = uerange * scaling;
    unsigned long ret;
    do ret = (unsigned long)g();
    while (ret >= past);
    return ret / scaling;
}

// libstdc++ (GCC 7-10) std::shuffle with a 32-bit generator: two swap positions per draw
template <class T>
void shuffle_gcc9(T *v, size_t n, std::mt19937 &g) {
    if (n == 0) return;
    size_t i = 1;
    if (n % 2 == 0) {
        std::swap(v[i], v[uniform_gcc9(g, 1)]);
        i++;
    }
    while (i < n) {
        const unsigned long swap_range = (unsigned long)i + 1;
        const unsigned long x = uniform_gcc9(g, swap_range * (swap_range + 1) - 1);
        const unsigned long p0 = x / (swap_range + 1), p1 = x % (swap_range + 1);
        std::swap(v[i], v[p0]);
        i++;
        std::swap(v[i], v[p1]);
        i++;
    }
}

}  // namespace

struct MapfObstacle {
    int id;
    P3 position, velocity;
    double radius, max_acc;
    // obstacle.hpp:26-36
    bool is_collided(const P3 &point, double agent_radius, double horizon, double uncertainty_horizon) const {
        const double dt = std::min(0.1 * horizon, 0.1);
        for (double t = 0; t <= horizon; t += dt) {
            const float tf = (float)t;
            const P3 op{position.x + velocity.x * tf, position.y + velocity.y * tf, position.z + velocity.z * tf};
            const double t_min = std::min(t, uncertainty_horizon);
            if (dist(op, point) < agent_radius + radius + 0.5 * max_acc * t_min * t_min) return true;
        }
        return false;
    }
};

struct Doi {  // grid_based_planner.hpp: DOI
    P3 agent_position;
    int closest_obs_id = -1;
    P3 closest_obs_point;
    float closest_obs_dist = (float)INF;
    std::set<int> cand_ids;
    bool exist() const { return !cand_ids.empty(); }
};

struct MapfAgent {
    int id;
    P3 position, start, waypoint, goal;
    std::vector<int> alert;  // indices (into the obstacle list) of the collision-alert obstacles
    P3 alert_position;
    Doi doi;
};

struct PlanResult {
    std::set<int> agent_ids;
    std::vector<std::vector<P3>> paths;
    int makespan() const {  // grid_based_planner.cpp:11-26, quirk included: size of the first longer path, minus one
        if (paths.empty()) return -1;
        int mx = 0;
        for (const auto &p : paths) {
            if (p.empty()) return -1;
            if (mx < (int)p.size()) mx = (int)p.size() - 1;
        }
        return mx;
    }
};

// DLSCQP_MAPF_PROFILE=1 in the environment: where a step's time goes, printed when the planner is destroyed
struct MapfProfile {
    bool on = std::getenv("DLSCQP_MAPF_PROFILE") != nullptr;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long steps = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    struct Scope {
        MapfProfile &p; int k; double t0;
        Scope(MapfProfile &pp, int kk) : p(pp), k(kk), t0(pp.on ? now() : 0.0) {}
        ~Scope() { if (p.on) p.t[k] += now() - t0; }
    };
    ~MapfProfile() {
        if (!on || !steps) return;
        const char *names[8] = {"grid map", "obstacle tables", "obstacles of interest", "escape goals", "PIBT", "goal tables (inside PIBT)", "waypoint rules", "whole step"};
        for (int k = 0; k < 8; k++) std::fprintf(stderr, "[mapf profile] %-28s %8.1f us per step\n", names[k], 1e6 * t[k] / steps);
    }
};

struct GridMapf {
    MapfProfile prof;
    dlscqp_mapf_config cfg;
    SfcWorldHost world;
    bool have_world = false;
    double gmin[3], gmax[3];
    int dim[3];
    // current grid map.  Two of them are kept per replanning step -- with the step's dynamic obstacles and without any (the
    // fallback of planMAPF, and all there is in a world without obstacles) -- and every communication group of the step
    // reuses them: the map depends on the world, the agents' radius and the obstacles, not on the group (the reference
    // rebuilds it per agent; the result is the same).  `exist`, `warning`, `nbr` are the active one.
    std::vector<char> exist, warning;
    // adjacency, flat: node v has nbr_n[v] neighbours at nbr_flat[6 v ...], in the order of Grid::Grid (graph.cpp:371-431)
    std::vector<int> nbr_flat;
    std::vector<unsigned char> nbr_n;
    struct NbrRange {
        const int *b, *e;
        const int *begin() const { return b; }
        const int *end() const { return e; }
    };
    NbrRange nbr_of(int v) const { return {nbr_flat.data() + 6 * (size_t)v, nbr_flat.data() + 6 * (size_t)v + nbr_n[v]}; }
    // A breadth-first distance table that is filled as far as it is asked: at(v) expands the search until v has its distance
    // (or the component of the source is exhausted: unreachable).  The values are those of the full search -- a node's first
    // visit is its distance -- so a planner that only asks about the nodes between its agents and their goals never pays
    // for the rest of the map.
    struct LazyBfs {
        std::vector<int> t, q;
        int head = 0, tail = 0;
        void start(int n_nodes, int src) {
            t.assign(n_nodes, (int)INF);
            q.resize(n_nodes + 1);  // (+1: lazy_at writes a candidate before it knows whether it is new)
            head = tail = 0;
            q[tail++] = src;
            t[src] = 0;
        }
    };
    int lazy_at(LazyBfs &b, int v) const {
        const int inf = (int)INF;
        while (b.t[v] == inf && b.head < b.tail) {
            const int n = b.q[b.head++];
            const int d = b.t[n] + 1;
            for (int m : nbr_of(n)) {  // without a branch on "seen before": it is taken half the time, in no pattern
                const int tm = b.t[m];
                const bool fresh = tm == inf;
                b.t[m] = fresh ? d : tm;
                b.q[b.tail] = m;
                b.tail += fresh ? 1 : 0;
            }
        }
        return b.t[v];
    }
    struct GridState {
        std::vector<char> exist, warning;
        std::vector<int> nbr_flat;
        std::vector<unsigned char> nbr_n;
        bool valid = false;
        double radius = -1.0;
    } stored[2];
    int active = -1;
    std::vector<char> static_free;  // node is clear of the static world (depends on the agents' radius only)
    double static_radius = -1.0;
    std::vector<int> static_adj;    // adjacency of the map without dynamic obstacles (update_grid_map)
    std::vector<unsigned char> static_adj_n;
    double static_adj_radius = -1.0;
    void use_state(int which) {
        if (active == which) return;
        auto swap_with = [&](GridState &g) { exist.swap(g.exist); warning.swap(g.warning); nbr_flat.swap(g.nbr_flat); nbr_n.swap(g.nbr_n); };
        if (active >= 0) swap_with(stored[active]);
        swap_with(stored[which]);
        active = which;
    }
    void new_step() {  // the obstacles have moved; the map without them stays
        stored[1].valid = false;
        tables_valid[0] = tables_valid[1] = false;
        goal_tables[1].clear();
    }
    // breadth-first distance tables from the obstacles' nodes, per grid state and step (every group of a step reads the same)
    mutable std::vector<LazyBfs> tables[2];  // by obstacle, started when the step's first group is planned, filled as far as asked
    bool tables_valid[2] = {false, false};
    std::vector<LazyBfs> *obs_tables_p = nullptr;
    mutable std::unordered_map<int, LazyBfs> goal_tables[2];  // PIBT's distance tables by goal node, per grid state
    PlanResult plan_result;
    int n_agents = 0;

    double gres(int k) const { return k < 2 ? cfg.grid_resolution : cfg.grid_resolution * cfg.agent_downwash; }
    int n_nodes() const { return dim[0] * dim[1] * dim[2]; }
    int node_id(int x, int y, int z) const { return dim[0] * dim[1] * z + dim[0] * y + x; }
    bool exist_at(int x, int y, int z) const {
        return x >= 0 && y >= 0 && z >= 0 && x < dim[0] && y < dim[1] && z < dim[2] && exist[node_id(x, y, z)];
    }
    void node_xyz(int id, int &x, int &y, int &z) const {
        x = id % dim[0];
        y = (id / dim[0]) % dim[1];
        z = id / (dim[0] * dim[1]);
    }
    P3 pos_to_point(int x, int y, int z) const {  // grid_based_planner.cpp:499-510
        P3 p;
        p.x = (float)(gmin[0] + x * gres(0));
        p.y = (float)(gmin[1] + y * gres(1));
        p.z = cfg.world_dimension == 2 ? (float)cfg.world_z_2d : (float)(gmin[2] + z * gres(2));
        return p;
    }
    P3 node_point(int id) const {
        int x, y, z;
        node_xyz(id, x, y, z);
        return pos_to_point(x, y, z);
    }
    void point_to_pos(const P3 &p, int g[3]) const {  // :516-532
        g[0] = g[1] = g[2] = 0;
        for (int i = 0; i < cfg.world_dimension; i++) {
            g[i] = (int)std::round(((double)p(i) - gmin[i]) / gres(i));
            if (g[i] < 0) g[i] = 0;
            else if (g[i] >= dim[i]) g[i] = dim[i] - 1;
        }
    }
    int point_to_id(const P3 &p) const {
        int g[3];
        point_to_pos(p, g);
        return node_id(g[0], g[1], g[2]);
    }
    int point_to_node(const P3 &p) const {  // -1: no such node
        const int id = point_to_id(p);
        return exist[id] ? id : -1;
    }
    int point_to_closest_node(const P3 &p) const {  // :441-497
        int node = point_to_node(p);
        if (node >= 0) return node;
        int g[3];
        point_to_pos(p, g);
        const int cand[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
        double min_dist = INF;
        for (const auto &c : cand) {
            const int x = g[0] + c[0], y = g[1] + c[1], z = g[2] + c[2];
            if (!exist_at(x, y, z)) continue;
            const double d = dist(p, pos_to_point(x, y, z));
            if (d < min_dist) { min_dist = d; node = node_id(x, y, z); }
        }
        if (node >= 0) return node;
        for (int id = 0; id < n_nodes(); id++) {
            if (!exist[id]) continue;
            const double d = dist(p, node_point(id));
            if (d < min_dist) { min_dist = d; node = id; }
        }
        return node;
    }
    bool is_valid(const P3 &p) const {
        for (int i = 0; i < cfg.world_dimension; i++)
            if ((double)p(i) < gmin[i] || (double)p(i) > gmax[i]) return false;
        return true;
    }
    bool is_occupied(const P3 &p) const { return !is_valid(p) || !exist[point_to_id(p)]; }

    // DynamicEDTOctomap::getDistanceAndClosestObstacle + grid_based_planner.cpp:111-123 for one node: L-infinity distance of
    // the node to the nearest occupied cell (as a box); the origin when no cell is within the map's 1 m horizon, as the
    // reference's default-constructed closest_point
    double static_clearance(const P3 &sp) const {
        const float delta = (float)(0.5 * cfg.world_resolution);
        auto box_linf = [&](const float centre[3]) {
            double d = 0;
            for (int k = 0; k < 3; k++) {
                const float lo = centre[k] - delta, hi = centre[k] + delta;
                const float cp = sp(k) < lo ? lo : (sp(k) > hi ? hi : sp(k));
                const double c = std::fabs((double)(sp(k) - cp));
                if (d < c) d = c;
            }
            return d;
        };
        const float origin[3] = {0, 0, 0};
        if (!have_world) return box_linf(origin);
        int c[3];
        for (int k = 0; k < 3; k++) c[k] = (int)std::floor((1.0 / world.res) * (double)sp(k));
        const int i = c[0] - world.lo[0], j = c[1] - world.lo[1], kk = c[2] - world.lo[2];
        if (i < 0 || j < 0 || kk < 0 || i >= world.n[0] || j >= world.n[1] || kk >= world.n[2]) return box_linf(origin);
        const size_t q = ((size_t)i * world.n[1] + j) * world.n[2] + kk;
        const int d2 = world.near_d2[q];
        if (d2 < 0 || !(std::sqrt((double)d2) * world.res < 1.0)) return box_linf(origin);
        double best = 1e30;
        for (int t = 0; t < world.near_n[q]; t++) {
            float centre[3];
            for (int k = 0; k < 3; k++) centre[k] = (float)((c[k] + world.near_off[q * 24 + 3 * t + k] + 0.5) * world.res);
            best = std::min(best, box_linf(centre));
        }
        return best;
    }

    // updateGridMap, :94-171
    void update_grid_map(double agent_radius, const std::vector<MapfObstacle> &obstacles) {
        const int which = obstacles.empty() ? 0 : 1;
        use_state(which);
        GridState &st = stored[which];
        if (st.valid && st.radius == agent_radius) return;  // this step's map (or the static one) is there already
        st.valid = true;
        st.radius = agent_radius;
        goal_tables[which].clear();  // a new map: its distance tables go with the old one
        const int nn = n_nodes();
        if (static_radius != agent_radius) {  // clearance of the static world: once
            static_free.assign(nn, 1);
            if (have_world)
                for (int x = 0; x < dim[0]; x++)
                    for (int y = 0; y < dim[1]; y++)
                        for (int z = 0; z < dim[2]; z++)
                            if (static_clearance(pos_to_point(x, y, z)) < agent_radius - EPS_F) static_free[node_id(x, y, z)] = 0;
            static_radius = agent_radius;
        }
        // The map without dynamic obstacles -- nodes clear of the static world, six-neighbour adjacency in the order of Grid::Grid
        // (graph.cpp:371-431: left, right, up, down, top, bottom) -- is built once per radius; a step's map is that one with the
        // warning nodes marked and the edges from safe nodes into warning nodes taken out.
        if (static_adj_radius != agent_radius) {
            static_adj.assign((size_t)6 * nn, -1);
            static_adj_n.assign(nn, 0);
            const int d6[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
            for (int z = 0; z < dim[2]; z++)
                for (int y = 0; y < dim[1]; y++)
                    for (int x = 0; x < dim[0]; x++) {
                        const int v = node_id(x, y, z);
                        if (!static_free[v]) continue;
                        for (const auto &d : d6) {
                            const int x1 = x + d[0], y1 = y + d[1], z1 = z + d[2];
                            if (x1 < 0 || y1 < 0 || z1 < 0 || x1 >= dim[0] || y1 >= dim[1] || z1 >= dim[2]) continue;
                            const int w = node_id(x1, y1, z1);
                            if (static_free[w]) static_adj[6 * (size_t)v + static_adj_n[v]++] = w;
                        }
                    }
            static_adj_radius = agent_radius;
        }
        exist = static_free;
        warning.assign(nn, 0);
        nbr_flat = static_adj;
        nbr_n = static_adj_n;
        if (obstacles.empty()) return;
        const double horizon = cfg.M * cfg.dt;
        // warning nodes, updateGridMap :125-160: a node the obstacle can reach within the horizon (Obstacle::isCollided samples
        // its straight path, obstacle.hpp:26-36).  Only the nodes of the box around that path, grown by what the obstacle can
        // ever reach (+ 1 mm against the float32 arithmetic of the samples), are looked at, and of those only the ones within
        // that distance of the path.
        std::vector<int> warned;
        for (const auto &o : obstacles) {
            const double tu = std::min(horizon, cfg.obs_uncertainty_horizon);
            const double reach = agent_radius + o.radius + 0.5 * o.max_acc * tu * tu + 1e-3;
            const double a3[3] = {o.position.x, o.position.y, o.position.z};
            const double d3[3] = {o.velocity.x * horizon, o.velocity.y * horizon, o.velocity.z * horizon};
            int lo[3], hi[3];
            for (int k = 0; k < 3; k++) {
                if (k == 2 && cfg.world_dimension == 2) { lo[k] = hi[k] = 0; continue; }
                const double p0 = std::min(a3[k], a3[k] + d3[k]) - reach, p1 = std::max(a3[k], a3[k] + d3[k]) + reach;
                lo[k] = std::max(0, (int)std::floor((p0 - gmin[k]) / gres(k)) - 1);
                hi[k] = std::min(dim[k] - 1, (int)std::ceil((p1 - gmin[k]) / gres(k)) + 1);
            }
            const double dd = d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2];
            for (int x = lo[0]; x <= hi[0]; x++)
                for (int y = lo[1]; y <= hi[1]; y++)
                    for (int z = lo[2]; z <= hi[2]; z++) {
                        const int id = node_id(x, y, z);
                        if (!static_free[id] || warning[id]) continue;
                        const P3 sp = pos_to_point(x, y, z);
                        const double px = sp.x - a3[0], py = sp.y - a3[1], pz = sp.z - a3[2];
                        double t = dd > 0 ? (px * d3[0] + py * d3[1] + pz * d3[2]) / dd : 0.0;
                        t = t < 0 ? 0 : (t > 1 ? 1 : t);
                        const double qx = px - t * d3[0], qy = py - t * d3[1], qz = pz - t * d3[2];
                        if (std::sqrt(qx * qx + qy * qy + qz * qz) > reach) continue;
                        if (o.is_collided(sp, agent_radius, horizon, cfg.obs_uncertainty_horizon)) {
                            warning[id] = 1;
                            warned.push_back(id);
                        }
                    }
        }
        // an edge v -> w exists unless it leads from a safe node into a warning node: take w out of its safe neighbours' lists
        // (the grid is symmetric: w's own list names them), keeping the order of the others
        for (int w : warned)
            for (int u : NbrRange{static_adj.data() + 6 * (size_t)w, static_adj.data() + 6 * (size_t)w + static_adj_n[w]}) {
                if (warning[u]) continue;
                int *lu = nbr_flat.data() + 6 * (size_t)u;
                int n = nbr_n[u], k = 0;
                for (int j = 0; j < n; j++)
                    if (lu[j] != w) lu[k++] = lu[j];
                nbr_n[u] = (unsigned char)k;
            }
    }

    double obs_cost(const std::set<int> &obs_ids, int node) const {  // :686-698
        double cost = 0;
        for (int oid : obs_ids) {
            const int d = lazy_at((*obs_tables_p)[oid], node);
            cost += d == 0 ? INF : 1.0 / ((double)d * d);
        }
        return cost;
    }

    // updateDOI, :196-250.  Obstacle ids are their indices in `obstacles` (obstacle_generator assigns them that way).
    void update_doi(std::vector<MapfAgent> &agents, const std::vector<MapfObstacle> &obstacles, double agent_radius) {
        for (auto &a : agents) {
            Doi d;
            d.agent_position = a.position;
            std::vector<int> cands;
            if (a.alert.empty()) {
                for (int oi = 0; oi < (int)obstacles.size(); oi++)
                    if (obstacles[oi].is_collided(a.waypoint, agent_radius, cfg.M * cfg.dt, cfg.obs_uncertainty_horizon)) cands.push_back(oi);
            } else {
                cands = a.alert;
            }
            double min_dist = INF;
            for (int oi : cands) {
                d.cand_ids.insert(obstacles[oi].id);
                const double dd = dist(obstacles[oi].position, a.alert_position);
                if (dd < min_dist) {
                    min_dist = dd;
                    d.closest_obs_point = obstacles[oi].position;
                    d.closest_obs_dist = (float)min_dist;
                    d.closest_obs_id = obstacles[oi].id;
                }
            }
            a.doi = d;
        }
    }

    // updateGoal, :253-305: an agent with an obstacle of interest heads for the node that is farthest (in the obstacle
    // cost) from those obstacles instead of its goal
    void update_goal(std::vector<MapfAgent> &agents) {
        for (auto &a : agents) {
            if (!a.doi.exist()) continue;
            int n = point_to_closest_node(a.doi.agent_position);
            const int g = point_to_closest_node(a.waypoint);
            P3 new_goal = node_point(n);
            double min_cost = INF;
            bool restart = false;
            std::queue<int> open;
            open.push(n);
            while (!open.empty()) {
                n = open.front();
                open.pop();
                if (!restart && n == g) {
                    std::queue<int> empty;
                    std::swap(open, empty);
                    open.push(g);
                    min_cost = INF;
                    new_goal = node_point(g);
                    restart = true;
                    continue;
                }
                const double c_n = obs_cost(a.doi.cand_ids, n);
                for (int m : nbr_of(n)) {
                    const double c_m = obs_cost(a.doi.cand_ids, m);
                    if (c_n < c_m + EPS_F) continue;
                    if (c_m < min_cost) {
                        min_cost = c_m;
                        new_goal = node_point(m);
                    }
                    open.push(m);
                }
                if (min_cost < 0.01) break;
            }
            a.goal = new_goal;
        }
    }

    // ---- PIBT (src/mapf/pibt.cpp) ---------------------------------------------------------------------------------
    struct PAgent {
        int id, v_now, v_next, g, o;
        int elapsed, init_d;
        float tie_breaker, obs_d;
    };

    // returns the plan as configurations [t][agent] of node ids; solved or not, like Solver::getSolution
    std::vector<std::vector<int>> run_pibt(const std::vector<int> &start, const std::vector<int> &cur, const std::vector<int> &goal,
                                           const std::vector<int> &obs, const std::vector<float> &obs_dist) {
        const int n = (int)cur.size();
        std::mt19937 mt(0);  // DEFAULT_SEED, a fresh generator for every Problem (problem.cpp:84)
        // Solver::createDistanceTable: BFS from every goal.  A table depends on the grid map and the goal node only: kept per
        // grid state (the map without dynamic obstacles never changes; the one with them is new every step)
        std::vector<LazyBfs *> table(n);
        // (bounded: escape goals change from step to step, and a table is two vectors of n_nodes -- a long mission on a large grid
        // would otherwise grow the cache towards n_nodes^2; a cleared cache is refilled as far as it is asked)
        if (goal_tables[active].size() > (size_t)(8 * n + 64)) goal_tables[active].clear();
        for (int i = 0; i < n; i++) {
            auto &cache = goal_tables[active];
            auto it = cache.find(goal[i]);
            if (it == cache.end()) {
                it = cache.emplace(goal[i], LazyBfs()).first;
                it->second.start(n_nodes(), goal[i]);
            }
            table[i] = &it->second;
        }
        auto goal_steps = [&](int i, int v) {  // Solver::pathDist: the table's entry, MAX_TIMESTEP where the goal cannot be reached
            MapfProfile::Scope sc(prof, 5);
            const int d = lazy_at(*table[i], v);
            return d >= (int)INF ? MAX_TIMESTEP : d;
        };
        std::vector<PAgent> A(n);
        std::vector<int> occupied_now(n_nodes(), -1), occupied_next(n_nodes(), -1);
        auto compare = [&](int ia, int ib) {  // pibt.cpp:15-21
            const PAgent &a = A[ia], &b = A[ib];
            if (a.obs_d != b.obs_d) return a.obs_d > b.obs_d;
            if (a.elapsed != b.elapsed) return a.elapsed < b.elapsed;
            if (a.init_d != b.init_d) return a.init_d < b.init_d;
            return a.tie_breaker < b.tie_breaker;
        };
        std::priority_queue<int, std::vector<int>, decltype(compare)> undecided(compare);
        std::vector<int> decided;
        for (int i = 0; i < n; i++) {
            A[i] = PAgent{i, cur[i], -1, goal[i], obs_dist[i] > 1000 ? -1 : obs[i], 0, goal_steps(i, start[i]), (float)i / (float)n, obs_dist[i]};
            undecided.push(i);
            occupied_now[cur[i]] = i;
        }
        std::vector<std::vector<int>> solution;
        solution.push_back(cur);

        auto obs_dist_to = [&](const PAgent &a, int node) -> float {  // pibt.cpp:228-235
            const float infinity = 10000;
            if (a.obs_d > infinity) return infinity;
            int x0, y0, z0, x1, y1, z1;
            node_xyz(a.o, x0, y0, z0);
            node_xyz(node, x1, y1, z1);
            const float dx = (float)(x0 - x1), dy = (float)(y0 - y1), dz = (float)(z0 - z1);
            return std::sqrt(dx * dx + dy * dy + dz * dz);
        };
        auto goal_dist = [&](const PAgent &a, int node) -> float {
            int x0, y0, z0, x1, y1, z1;
            node_xyz(a.g, x0, y0, z0);
            node_xyz(node, x1, y1, z1);
            const float dx = (float)(x0 - x1), dy = (float)(y0 - y1), dz = (float)(z0 - z1);
            return std::sqrt(dx * dx + dy * dy + dz * dz);
        };
        auto choose_node = [&](PAgent &a) -> int {  // pibt.cpp:149-203
            int C[7], nc = 0;  // at most six neighbours and the node itself
            for (int u : nbr_of(a.v_now)) C[nc++] = u;
            C[nc++] = a.v_now;
            shuffle_gcc9(C, (size_t)nc, mt);
            int v = -1;
            for (int ci = 0; ci < nc; ci++) {
                const int u = C[ci];
                if (occupied_next[u] != -1) continue;
                const int aj = occupied_now[u];
                if (aj != -1 && A[aj].v_next == a.v_now) continue;
                if (u == a.g) return u;
                if (v == -1) {
                    v = u;
                } else {
                    const int c_v = goal_steps(a.id, v), c_u = goal_steps(a.id, u);
                    const float o_v = obs_dist_to(a, v), o_u = obs_dist_to(a, u);
                    const float d_v = goal_dist(a, v), d_u = goal_dist(a, u);
                    if ((c_u < c_v) || (c_u == c_v && occupied_now[v] != -1 && occupied_now[u] == -1) || (c_u == c_v && o_u > o_v) ||
                        (c_u == c_v && occupied_now[v] == -1 && occupied_now[u] == -1 && d_u < d_v))
                        v = u;
                }
            }
            return v;
        };
        auto plan_one_step = [&](PAgent &a) -> int {
            const int v = choose_node(a);
            if (v != -1) {
                occupied_next[v] = a.id;
                a.v_next = v;
            }
            return v;
        };
        // funcPIBT (pibt.cpp:113-135), recursion made explicit through std::function-free lambda recursion
        struct Rec {
            static bool run(int ai, std::vector<PAgent> &A, std::vector<int> &occupied_now, std::vector<int> &occupied_next,
                            const std::function<int(PAgent &)> &plan_one_step) {
                int v = plan_one_step(A[ai]);
                while (v != -1) {
                    const int aj = occupied_now[v];
                    if (aj != -1) {
                        if (aj != ai && A[aj].v_next == -1) {
                            if (!run(aj, A, occupied_now, occupied_next, plan_one_step)) {
                                v = plan_one_step(A[ai]);
                                continue;
                            }
                        }
                    }
                    return true;
                }
                occupied_next[A[ai].v_now] = ai;
                A[ai].v_next = A[ai].v_now;
                return false;
            }
        };
        const std::function<int(PAgent &)> pos_fn = plan_one_step;
        int timestep = 0;
        while (true) {
            while (!undecided.empty()) {
                const int ai = undecided.top();
                undecided.pop();
                if (A[ai].v_next == -1) Rec::run(ai, A, occupied_now, occupied_next, pos_fn);
                decided.push_back(ai);
            }
            bool at_goal = true;
            std::vector<int> config(n, -1);
            for (int ai : decided) {
                PAgent &a = A[ai];
                if (occupied_now[a.v_now] == ai) occupied_now[a.v_now] = -1;
                occupied_next[a.v_next] = -1;
                config[a.id] = a.v_next;
                occupied_now[a.v_next] = ai;
                at_goal &= (a.v_next == a.g);
                a.elapsed = (a.v_next == a.g) ? 0 : a.elapsed + 1;
                a.v_now = a.v_next;
                a.v_next = -1;
                undecided.push(ai);
            }
            decided.clear();
            solution.push_back(config);
            ++timestep;
            if (at_goal) break;
            if (timestep >= MAX_TIMESTEP) break;
        }
        return solution;
    }

    bool is_solution_valid(const PlanResult &pr, const std::vector<MapfAgent> &agents) const {  // :367-381
        if (pr.paths.empty()) return false;
        if ((int)pr.paths.size() < n_agents) return false;  // the reference reads past the end here; its result is unused then
        for (int qi = 0; qi < n_agents; qi++) {
            if (pr.paths[qi].empty()) return false;
            if (norm(sub(pr.paths[qi].back(), agents[qi].goal)) > EPS_F) return false;
        }
        return true;
    }

    PlanResult plan_initial_path(const PlanResult &prev, const std::vector<MapfAgent> &agents) const {  // :383-422
        std::set<int> ids;
        for (const auto &a : agents) ids.insert(a.id);
        if (prev.agent_ids.empty() || (int)prev.agent_ids.size() != n_agents || ids != prev.agent_ids) return prev;
        std::set<int> updated;
        for (int qi = 0; qi < n_agents; qi++)
            if (prev.paths[qi].size() < 2 || norm(sub(prev.paths[qi][1], agents[qi].waypoint)) < EPS_F) updated.insert(qi);
        PlanResult init = prev;
        if ((int)updated.size() == n_agents) {
            for (auto &p : init.paths)
                if (p.size() > 1) p.erase(p.begin());
        } else {
            for (int qi = 0; qi < n_agents; qi++)
                if (prev.paths[qi].size() > 1 && updated.count(qi)) init.paths[qi][0] = prev.paths[qi][1];
        }
        return init;
    }

    void update_plan_result(const std::vector<std::vector<int>> &plan, const std::vector<MapfAgent> &agents) {  // :307-353
        const PlanResult prev = plan_initial_path(plan_result, agents);
        plan_result = PlanResult();
        int repeat_start = 0;
        for (int i = 1; i < (int)plan.size(); i++) {
            bool repeat = true;
            for (int qi = 0; qi < n_agents; qi++)
                if (plan[0][qi] != plan[i][qi]) { repeat = false; break; }
            if (repeat) repeat_start = i;
        }
        plan_result.paths.resize(n_agents);
        for (int qi = 0; qi < n_agents; qi++) {
            plan_result.agent_ids.insert(agents[qi].id);
            for (int i = repeat_start; i < (int)plan.size(); i++) plan_result.paths[qi].push_back(node_point(plan[i][qi]));
        }
        const bool valid = is_solution_valid(plan_result, agents);
        const bool valid_prev = is_solution_valid(prev, agents);
        bool doi_exist = false;
        for (const auto &a : agents)
            if (a.doi.exist()) { doi_exist = true; break; }
        const bool new_agent = plan_result.agent_ids != prev.agent_ids;
        PlanResult tmp = plan_result;
        const bool better = tmp.makespan() < PlanResult(prev).makespan();
        if (!doi_exist && !new_agent && (!valid || (!better && valid_prev))) plan_result = prev;
    }

    bool run_mapf(const std::vector<MapfAgent> &agents) {  // :424-453
        std::vector<int> start, cur, goal, obs;
        std::vector<float> od;
        for (const auto &a : agents) {
            const int s = point_to_node(a.start), c = point_to_node(a.waypoint), g = point_to_node(a.goal);
            const int o = point_to_closest_node(a.doi.closest_obs_point);
            if (s < 0 || c < 0 || g < 0 || o < 0) return false;  // MAPF::Problem halts on a missing node
            start.push_back(s); cur.push_back(c); goal.push_back(g); obs.push_back(o);
            od.push_back(a.doi.closest_obs_dist);
        }
        const auto plan = run_pibt(start, cur, goal, obs, od);
        update_plan_result(plan, agents);
        return !plan.empty();
    }

    // planMAPF, :64-92
    bool plan_mapf(std::vector<MapfAgent> agents, double agent_radius, const std::vector<MapfObstacle> &obstacles) {
        n_agents = (int)agents.size();
        { MapfProfile::Scope sc(prof, 0); update_grid_map(agent_radius, obstacles); }
        bool valid = true;
        for (const auto &a : agents)
            if (is_occupied(a.waypoint) || is_occupied(a.goal)) { valid = false; break; }
        if (!valid) update_grid_map(agent_radius, {});
        bool any_node = false;
        for (int id = 0; id < n_nodes() && !any_node; id++) any_node = exist[id];
        if (!any_node) return false;  // every node is occupied: point_to_closest_node has nothing to return (the planner "fails")
        if (!tables_valid[active]) {
            MapfProfile::Scope sc(prof, 1);
            tables[active].resize(obstacles.size());
            for (size_t oi = 0; oi < obstacles.size(); oi++) tables[active][oi].start(n_nodes(), point_to_closest_node(obstacles[oi].position));
            tables_valid[active] = true;
        }
        obs_tables_p = &tables[active];
        { MapfProfile::Scope sc(prof, 2); update_doi(agents, obstacles, agent_radius); }
        { MapfProfile::Scope sc(prof, 3); update_goal(agents); }
        bool ok;
        { MapfProfile::Scope sc(prof, 4); ok = run_mapf(agents); }
        if (!ok && !obstacles.empty()) {
            update_grid_map(agent_radius, {});
            ok = run_mapf(agents);
        }
        return ok;
    }
};

}  // namespace dlscqp

using namespace dlscqp;

struct dlscqp_mapf {
    GridMapf g;
};

extern "C" int dlscqp_mapf_create(const dlscqp_mapf_config *cfg, const double *boxes, int32_t n_boxes, dlscqp_mapf **out) {
    if (!cfg || !out || (n_boxes > 0 && !boxes)) return DLSCQP_ERR_INVALID_ARGUMENT;
    if (cfg->world_dimension < 2 || cfg->world_dimension > 3 || !(cfg->grid_resolution > 0)) return DLSCQP_ERR_INVALID_ARGUMENT;
    dlscqp_mapf *m = new dlscqp_mapf();
    GridMapf &g = m->g;
    g.cfg = *cfg;
    if (n_boxes > 0 || cfg->world_use_octomap) {
        float wmin[3], wmax[3];
        for (int k = 0; k < 3; k++) { wmin[k] = (float)cfg->world_min[k]; wmax[k] = (float)cfg->world_max[k]; }
        build_sfc_world(boxes, n_boxes, cfg->world_resolution, wmin, wmax, g.world);
        g.have_world = true;
    }
    // GridBasedPlanner::GridBasedPlanner, grid_based_planner.cpp:28-50
    for (int i = 0; i < 3; i++) {
        const double r = g.gres(i);
        g.gmin[i] = -std::floor((-cfg->world_min[i] + EPS) / r) * r;
        g.gmax[i] = std::floor((cfg->world_max[i] + EPS) / r) * r;
    }
    if (cfg->world_dimension == 2) g.gmin[2] = g.gmax[2] = cfg->world_z_2d;
    for (int i = 0; i < cfg->world_dimension; i++) g.dim[i] = (int)std::round((g.gmax[i] - g.gmin[i]) / g.gres(i)) + 1;
    if (cfg->world_dimension == 2) g.dim[2] = 1;
    *out = m;
    return DLSCQP_SUCCESS;
}

extern "C" void dlscqp_mapf_destroy(dlscqp_mapf *m) { delete m; }

// MultiSyncSimulator::decentralizedMAPP, multi_sync_simulator.cpp:308-466
extern "C" int dlscqp_mapf_step(dlscqp_mapf *m, const dlscqp_mapf_step_args *s) {
    if (!m || !s || s->n_agents < 0) return DLSCQP_ERR_INVALID_ARGUMENT;
    if (s->n_agents == 0) return DLSCQP_SUCCESS;
    if (!s->position || !s->start_point || !s->desired_goal || !s->current_goal || !s->next_waypoint) return DLSCQP_ERR_INVALID_ARGUMENT;
    if (s->n_obs < 0 || (s->n_obs > 0 && (!s->obs_position || !s->obs_velocity || !s->obs_radius || !s->obs_max_acc))) return DLSCQP_ERR_INVALID_ARGUMENT;
    if (s->plans && s->n1 <= 0) return DLSCQP_ERR_INVALID_ARGUMENT;
    GridMapf &g = m->g;
    MapfProfile::Scope whole(g.prof, 7);
    g.prof.steps++;
    if (g.n_nodes() <= 0) return DLSCQP_ERR_INVALID_ARGUMENT;  // a world without a grid: there is nothing to plan on
    g.new_step();
    const dlscqp_mapf_config &cfg = g.cfg;
    const int N = s->n_agents;
    auto p3 = [](const float *a, int i) { return P3{a[3 * i], a[3 * i + 1], a[3 * i + 2]}; };
    // communication groups (:311-341)
    std::vector<std::set<int>> groups;
    groups.push_back({0});
    for (int qi = 1; qi < N; qi++) {
        int cand = -1;
        int gi = 0;
        while (gi < (int)groups.size()) {
            for (int qj : groups[gi]) {
                const double d = linf(p3(s->position, qi), p3(s->position, qj));
                if (cfg.communication_range < 0 || d < cfg.communication_range) {
                    if (cand == -1) {
                        groups[gi].insert(qi);
                        cand = gi;
                    } else {
                        groups[cand].insert(groups[gi].begin(), groups[gi].end());
                        groups.erase(groups.begin() + gi);
                        gi--;
                    }
                    break;
                }
            }
            gi++;
        }
        if (cand == -1) groups.push_back({qi});
    }
    std::vector<MapfObstacle> obstacles;
    for (int o = 0; o < s->n_obs; o++)
        obstacles.push_back(MapfObstacle{o, p3(s->obs_position, o), p3(s->obs_velocity, o), s->obs_radius[o], s->obs_max_acc[o]});
    if (s->out_group)
        for (int gi = 0; gi < (int)groups.size(); gi++)
            for (int qi : groups[gi]) s->out_group[qi] = gi;
    int failures = 0;
    const int M = cfg.M, n1 = s->n1;
    for (const auto &group : groups) {
        const std::vector<int> gv(group.begin(), group.end());
        const int gs = (int)gv.size();
        std::vector<MapfAgent> agents;
        for (int qi : gv) {
            MapfAgent a;
            a.id = qi;
            a.position = p3(s->position, qi);
            a.start = p3(s->start_point, qi);
            a.waypoint = p3(s->next_waypoint, qi);
            a.goal = p3(s->desired_goal, qi);
            a.alert_position = a.position;  // updateCollisionAlert refreshes it every step (:283-306)
            if (s->alert && s->n_obs > 0)
                for (int o = 0; o < s->n_obs; o++)
                    if (s->alert[(size_t)qi * s->n_obs + o]) a.alert.push_back(o);
            agents.push_back(a);
        }
        const bool ok = g.plan_mapf(agents, cfg.agent_radius, obstacles);
        if (!ok) { failures++; continue; }
        std::vector<P3> desired(gs);
        for (int q = 0; q < gs; q++) {
            const auto &path = g.plan_result.paths[q];
            desired[q] = path[std::min<int>(1, (int)path.size() - 1)];
            if (s->out_desired) {
                float *o = s->out_desired + 3 * gv[q];
                o[0] = desired[q].x; o[1] = desired[q].y; o[2] = desired[q].z;
            }
        }
        std::set<int> cand;
        for (int q = 0; q < gs; q++) {
            const int qi = gv[q];
            bool in_range = true;
            if (cfg.communication_range > 0) {
                for (int mm = 0; mm < M + 1; mm++) {
                    double d;
                    if (!s->plans) d = linf(desired[q], p3(s->position, qi));
                    else {
                        const float *pl = s->plans + (size_t)qi * M * n1 * 3;
                        const float *pt = mm < M ? pl + (size_t)mm * n1 * 3 : pl + ((size_t)(M - 1) * n1 + (n1 - 1)) * 3;
                        d = linf(desired[q], P3{pt[0], pt[1], pt[2]});
                    }
                    if (d > 0.5 * cfg.communication_range - EPS_F) { in_range = false; break; }
                }
            }
            const P3 wp = p3(s->next_waypoint, qi), cg = p3(s->current_goal, qi);
            // Line(next_waypoint, desired).includePoint(current_goal_point), geometry.hpp:28-30
            const bool on_line = std::fabs(dist(cg, wp) + dist(cg, desired[q]) - dist(wp, desired[q])) < EPS_F;
            if (in_range && norm(sub(desired[q], wp)) > EPS_F && on_line) cand.insert(qi);
        }
        // find a valid update (:418-447): no two agents end up with the same waypoint
        bool update = false;
        while (!update && !cand.empty() && gs > 1) {
            for (int qi : cand) {
                const int q = (int)(std::find(gv.begin(), gv.end(), qi) - gv.begin());
                for (int qj : gv) {
                    const int q2 = (int)(std::find(gv.begin(), gv.end(), qj) - gv.begin());
                    if (qi == qj) continue;
                    const P3 wj = cand.count(qj) ? desired[q2] : p3(s->next_waypoint, qj);
                    if (dist(desired[q], wj) < EPS_F) {
                        cand.erase(qi);
                        update = false;
                        break;
                    } else {
                        update = true;
                    }
                }
                if (!update) break;
            }
        }
        for (int qi : cand) {
            const int q = (int)(std::find(gv.begin(), gv.end(), qi) - gv.begin());
            float *o = s->next_waypoint + 3 * qi;
            o[0] = desired[q].x; o[1] = desired[q].y; o[2] = desired[q].z;
        }
    }
    return failures ? 1 : DLSCQP_SUCCESS;
}
