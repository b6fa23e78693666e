// shim_test.cpp -- drives the HIP library through the reference's own C++ interface
// (MATP::TrajOptimizer / GoalOptimizer / CollisionConstraints as mirrored in matp.hpp), the way
// TrajPlanner does (/root/reference/src/traj_planner.cpp:29, 491-496, 737-748).  Run on the GPU box
// by tests/test_gpu_shim.py.  Exit code 0 = all checks passed.
#include <cstdio>
#include <chrono>
#include <cmath>

#ifdef SHIM_TEST_OCTOMAP  // the branch a maintainer of the reference compiles: octomap's own point3d (here: tests/cpp/octomap_double)
#include <octomap/octomap_types.h>
#define MATP_HAVE_OCTOMAP
#endif
#include "../../dlsc_gc_planner_amd/include/dlsc_gc_planner_amd/matp.hpp"

using namespace MATP;

static int failures = 0;
#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static double bern(int n, int i, double t) {
    double c = 1;
    for (int k = 0; k < i; k++) c = c * (n - k) / (k + 1);
    return c * std::pow(t, i) * std::pow(1 - t, n - i);
}

int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    // ---- known answer from the reference's committed log (first replanning step, agent 0) -----------
    Param param;
    param.world_dimension = 2; param.M = 10; param.n = 5; param.phi = 3; param.dt = 0.2;
    param.control_input_weight = 0.01; param.terminal_weight = 1; param.communication_range = 3.0;
    param.world_use_octomap = false; param.world_z_2d = 1.0; param.slack_threshold = 0.1;
    Mission mission;
    mission.world_min = point3d(-2.0f, -0.3f, 0.0f);
    mission.world_max = point3d(6.0f, 4.3f, 2.5f);
    Agent agent;
    agent.id = 0;
    agent.current_state.position = point3d(-1, 3, 1);
    agent.current_goal_point = point3d(-0.5f, 3, 1);
    agent.next_waypoint = point3d(-0.5f, 3, 1);
    agent.max_vel = 1; agent.max_acc = 2; agent.radius = 0.15; agent.nominal_velocity = 1; agent.downwash = 2;
    {
        TrajOptimizer opt(param, mission, 0 /* B placeholder */);
        CollisionConstraints cons(param, mission, agent.radius, agent.max_vel);
        cons.initializeLSC(Obstacles());
        TrajOptResult r = opt.solve(agent, cons, traj_t(), true);
        double px = 0;
        for (int i = 0; i <= 5; i++) px += r.desired_traj[0][i].x() * bern(5, i, 0.5);
        std::printf("KAT pos(0.1s) = %.7f (log: -0.998916)  cost = %.9f\n", px, r.total_qp_cost);
        CHECK(std::fabs(px - (-0.998916)) < 2e-6);
        CHECK(std::fabs(r.desired_traj[0][5].x() - (-0.992206)) < 2e-6);
        CHECK(r.desired_traj[3][2].z() == 1.0f);  // world_z_2d in 2-D
        CHECK(std::fabs(r.total_qp_cost - 0.2024006) < 1e-6);
        CHECK(!r.collision_alert.activated());
        GoalOptimizer gopt(param, mission);
        point3d g = gopt.solve(agent, cons, point3d(-1, 3, 1), point3d(-0.5f, 3, 1));
        CHECK(g.x() == -0.5f && g.y() == 3.0f);  // unconstrained LP slides the goal onto the waypoint
    }
    // ---- 3-D, one neighbour agent + one dynamic obstacle, then an infeasible set ----------------------
    Param p3 = param;
    p3.world_dimension = 3; p3.M = 5; p3.communication_range = -1; p3.slack_collision_weight = 100;
    Mission m3;
    m3.world_min = point3d(-5, -5, 0);
    m3.world_max = point3d(5, 5, 2.5f);
    Agent a3 = agent;
    a3.current_state.position = point3d(0, 0, 1);
    a3.current_goal_point = point3d(1.5f, 0, 1);
    a3.next_waypoint = point3d(2.0f, 0, 1);
    TrajOptimizer opt3(p3, m3);
    CollisionConstraints cons(p3, m3, 0.15, 1.0);
    Obstacles obs(2);
    obs[0].type = ObstacleType::AGENT; obs[0].position = point3d(1, 0.1f, 1);
    obs[1].type = ObstacleType::DYN_SPIN; obs[1].position = point3d(1, -1, 1);
    cons.initializeLSC(obs);
    // rows only on the last two segments (default-constructed LSCs elsewhere are ignored, traj_optimizer.cpp:422)
    for (int m = 3; m < p3.M; m++)
        for (int i = 0; i <= p3.n; i++) {
            cons.setLSC(0, m, i, LSC(point3d(1, 0.1f, 1), point3d(0, -1, 0), 0.2));   // hard: y <= -0.1
            cons.setLSC(1, m, i, LSC(point3d(1, -1, 1), point3d(0, 1, 0), 1.05));     // soft: y >= +0.05 (conflicts)
        }
    {
        TrajOptResult r = opt3.solve(a3, cons, traj_t(), true);
        double ymax = -1e9;
        for (int m = 3; m < p3.M; m++)
            for (int i = 0; i <= p3.n; i++) ymax = std::fmax(ymax, r.desired_traj[m][i].y());
        std::printf("3-D: max y on segments 3.. = %.6f (hard LSC: <= -0.1)  alert obstacles = %zu\n", ymax, r.collision_alert.obstacles.size());
        CHECK(ymax <= -0.1 + 1e-6);
        CHECK(r.collision_alert.activated() && r.collision_alert.obstacles[0].id == 1);  // slack > threshold
        GoalOptimizer g3(p3, m3);
        bool thrown = false;
        try {
            // both the current goal and the waypoint violate the static LSC by 0.1 m: real infeasibility
            g3.solve(a3, cons, a3.current_goal_point, a3.next_waypoint);
        } catch (PlanningReport rep) {
            thrown = rep == PlanningReport::QPFAILED;
        }
        CHECK(thrown);
    }
    obs[1].type = ObstacleType::AGENT;  // now both are hard: y <= -0.1 and y >= 0.3 on the same points
    cons.initializeLSC(obs);
    for (int m = 3; m < p3.M; m++)
        for (int i = 0; i <= p3.n; i++) {
            cons.setLSC(0, m, i, LSC(point3d(1, 0.1f, 1), point3d(0, -1, 0), 0.2));
            cons.setLSC(1, m, i, LSC(point3d(1, 0.1f, 1), point3d(0, 1, 0), 0.2));
        }
    {
        bool thrown = false;
        try {
            opt3.solve(a3, cons, traj_t(), true);
        } catch (PlanningReport rep) {
            thrown = rep == PlanningReport::QPFAILED;  // traj_optimizer.cpp:152
        }
        CHECK(thrown);
        // ... and, like the reference's catch block (traj_optimizer.cpp:112-146), leaves the conflict behind: the two opposite
        // half-spaces y <= -0.1 and y >= 0.3 on one control point prove it, each with the multiplier 1 (violated by 0.4)
        {
            const std::string &cf = opt3.lastConflict();
            std::printf("%s", cf.c_str());
            CHECK(cf.find("Conflict :") == 0);
            size_t proved = 0, at = 0;
            while ((at = cf.find("Proved  : c", at)) != std::string::npos) { proved++; at++; }
            CHECK(proved == 2);
            CHECK(cf.find("violated by 0.4") != std::string::npos);
            CHECK(cf.find("(multiplier 1)") != std::string::npos);
        }
        // the batch entry reports the same verdict without throwing
        std::vector<const Agent *> as{&a3, &a3};
        CollisionConstraints free_cons(p3, m3);
        free_cons.initializeLSC(Obstacles());
        std::vector<const CollisionConstraints *> cs{&cons, &free_cons};
        std::vector<int> st;
        std::vector<TrajOptResult> rs = opt3.solveBatch(as, cs, st);
        CHECK(st[0] == DLSCQP_INFEASIBLE && st[1] == DLSCQP_OK);
        CHECK(rs[1].total_qp_cost > 0);
    }
    // ---- construction cost: the reference builds a temporary GoalOptimizer per agent per step (traj_planner.cpp:493-495) and
    // hands TrajOptimizer an Eigen matrix B (:29): both must be cheap / compile with any matrix-like argument -----------------
    {
        struct FakeMatrixXd { double v[36]; int rows() const { return 6; } };  // stands in for Eigen::MatrixXd (absent here)
        FakeMatrixXd B{};
        TrajOptimizer with_b(p3, m3, B);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 1000; k++) {
            GoalOptimizer g(p3, m3);
            (void)g;
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000;
        std::printf("GoalOptimizer construction: %.2f us each (shared context)\n", us);
        CHECK(us < 50.0);
        // updateParam keeps the cached dimensions (traj_optimizer.cpp:167-169 vs :7-11) and changes the weights
        Param p4 = p3;
        p4.M = 7;                       // ignored: M stays as constructed
        p4.control_input_weight = 0.05;
        CollisionConstraints none(p3, m3);
        none.initializeLSC(Obstacles());
        TrajOptResult before = with_b.solve(a3, none, traj_t(), true);
        with_b.updateParam(p4);
        TrajOptResult after = with_b.solve(a3, none, traj_t(), true);
        CHECK((int)after.desired_traj.size() == p3.M);
        CHECK(std::fabs(after.total_qp_cost - before.total_qp_cost) > 1e-9);
    }
    // ---- unsupported degree: std::invalid_argument like traj_optimizer.cpp:209 ------------------------
    {
        Param bad = p3;
        bad.n = 4;
        bool thrown = false;
        try {
            TrajOptimizer o(bad, m3);
        } catch (const std::invalid_argument &e) {
            thrown = true;
            std::printf("invalid_argument: %s\n", e.what());
        }
        CHECK(thrown);
    }
    std::printf(failures ? "shim_test: %d FAILURES\n" : "shim_test: ok\n", failures);
    return failures ? 1 : 0;
}
