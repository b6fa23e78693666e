/*
 * dlsc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's trajectory-QP / goal-LP path
 * (/root/reference/src/traj_optimizer.cpp, src/goal_optimizer.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * library (dlsc_gc_planner_amd/csrc -> libdlscqp.so) never links, loads or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"): the arithmetic of this path lives in IBM ILOG CPLEX
 * 22.1.1 (CMakeLists.txt:33 of the reference), which is proprietary and absent, and the
 * reference has no tests.  The oracle is pinned against (1) the committed run log
 * log/result_1742185870.978562_DLSCGC_10agents.csv of the reference (known-answer rows copied
 * as numbers into tests/golden/kat_log.json) and (2) cross-solver vectors produced in the build
 * container with scipy's bundled HiGHS QP solver (tests/golden/make_golden.py).
 *
 * The model is built dense and in the reference's own variable / row order
 * (populatebyrow, traj_optimizer.cpp:225-527) and solved with a generic method
 * (Householder null-space elimination of the equalities + Goldfarb-Idnani 1983 dual active-set
 * with Givens updates), deliberately unlike the structured GPU solver.
 */
#ifndef DLSC_ORACLE_H
#define DLSC_ORACLE_H

#include "../include/dlscqp.h" /* boundary types only (config, agent, lsc_row, status codes) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_model {
    int nv, neq, nin;
    int n_ctrl;  /* dim*M*(n+1) */
    int n_slack; /* M*n_slack_obs */
    int terminal_segments;
    double *Q;   /* nv*nv, objective x'Qx + c'x + c0 (no 1/2, traj_optimizer.cpp:286-331) */
    double *c;   /* nv */
    double c0;
    double *Aeq; /* neq*nv row-major */
    double *beq; /* neq */
    double *Ain; /* nin*nv row-major, rows in the reference's order */
    double *lo;  /* nin  lo <= Ain x <= hi, +-ORACLE_INF when one-sided */
    double *hi;  /* nin */
    double *lb;  /* nv variable bounds */
    double *ub;  /* nv */
} oracle_model;

#define ORACLE_INF 1e300

/* traj_optimizer.cpp:172-187 with B from polynomial.hpp:280-293.  Qb is (n+1)*(n+1). */
void oracle_build_q_base(int n, int phi, int phi_n, double dt, double *Qb);
/* traj_optimizer.cpp:189-223.  Aeq_base is ((M-2)*phi) x (M*(n+1)) row-major; returns rows. */
int oracle_build_aeq_base(int M, int n, int phi, double dt, double *Aeq_base);
/* traj_optimizer.cpp:543-551 with octomap float32 vector semantics. */
int oracle_terminal_segments(const dlscqp_config *cfg, const dlscqp_agent *ag);

/* populatebyrow, traj_optimizer.cpp:225-527.  Returns NULL on invalid argument. */
oracle_model *oracle_model_build(const dlscqp_config *cfg, const dlscqp_agent *ag, int n_rows,
                                 const dlscqp_lsc_row *rows, const float *sfc /*[M][6]|NULL*/);
void oracle_model_free(oracle_model *m);

/* Generic dense convex QP solve of a model.  x: nv.  Returns a DLSCQP_* verdict. */
int oracle_qp_solve(const oracle_model *m, double *x, double *obj, int *iters);

/* TrajOptimizer::solve (traj_optimizer.cpp:18-165).  x_f64 (nv) optional. */
int oracle_traj_solve(const dlscqp_config *cfg, const dlscqp_agent *ag, int n_rows,
                      const dlscqp_lsc_row *rows, const float *sfc, float *ctrl /*[M][n+1][3]*/,
                      double *x_f64, double *cost, double *slack_cost /*[n_slack_obs]*/,
                      int *iters);

/* GoalOptimizer::solve (goal_optimizer.cpp:7-198).  Returns DLSCQP_OK, DLSCQP_GOAL_FALLBACK or
 * DLSCQP_INFEASIBLE (the reference throws QPFAILED). */
int oracle_goal_solve(const dlscqp_config *cfg, const dlscqp_agent *ag, int n_rows,
                      const dlscqp_lsc_row *rows, const float *sfc, float goal_out[3],
                      double *t_out);

/* Batch drivers used for the CPU baseline timing (agents solved one after another, like
 * MultiSyncSimulator::plan, multi_sync_simulator.cpp:516-524). */
int oracle_plan_batch(const dlscqp_config *cfg, int n_agents, const dlscqp_agent *agents,
                      const int64_t *row_offsets, const dlscqp_lsc_row *rows, const float *sfc,
                      int max_slack_obs, float *out_goal, int *out_goal_status, float *out_ctrl,
                      double *out_x, double *out_cost, double *out_slack_cost, int *out_status,
                      int do_goal);

/* Trajectory<point3d>::getStateAt (trajectory.cpp:110-199) with float32 point3d semantics.
 * ctrl [M][n+1][3]; state_out = {pos[3], vel[3], acc[3]}. */
void oracle_state_at(const float *ctrl, int M, int n, double dt, double time, float state_out[9]);


/* ---------------------------------------------------------------------------------------------
 * Row N1 of SURVEY.md section 8(f): DLSC-GC half-space generation for neighbouring agents,
 * TrajPlanner::generateDLSCGC (traj_planner.cpp:603-666) with its helpers
 *   initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol   (:290-327, :409-421)
 *   planConstVelTraj                                            (trajectory.cpp:78-91)
 *   downwashBetween, coordinateTransform                        (:1150-1161, :1183-1187)
 *   normalVectorBetweenPolys                                    (:1102-1127)
 *   closestPointsBetweenPointAndConvexHull -> openGJK           (geometry.hpp:276-306)
 *   closestPointsBetweenLineSegments / Lines / PointAndLineSegment (geometry.hpp:76-274)
 * restated with octomath::Vector3 float32 semantics.  The point-to-hull distance sub-problem is
 * solved by exact enumeration of the hull's vertices, edges and triangles unless a GJK callback is
 * installed with oracle_set_gjk(); tests install the reference's own openGJK, compiled from
 * /root/reference into oracle/_ref/libref_gjk.so, to pin the enumeration against it. */
typedef double (*oracle_gjk_fn)(const double *pts /*[n][3]*/, int n, double v[3]);
void oracle_set_gjk(oracle_gjk_fn fn);
double oracle_hull_closest_point(const double *pts, int n, double v[3]);

/* plan [M][n+1][3] float32 -> shifted plan (initial trajectory of the agent = prediction of it by others) */
void oracle_shift_plan(const float *prev, int M, int n, float *out);
void oracle_const_vel_plan(const float pos[3], const float vel[3], int M, int n, double dt, float *out);

/* All DLSC-GC rows of one agent against `n_nbr` neighbouring agents.  init_* are the (already shifted or
 * constant-velocity) trajectories.  rows_out gets n_nbr * (M*(n+1) - phi) rows in populatebyrow's order
 * (obstacle, segment, control point; rows with m = 0, i < phi are not emitted).  Returns the row count. */
int oracle_generate_dlscgc(const dlscqp_config *cfg, const float *init_self, const float goal_self[3],
                           double radius_self, double downwash_self, int n_nbr, const float *const *init_nbr,
                           const float *goal_nbr /*[n_nbr][3]*/, const double *radius_nbr,
                           const double *downwash_nbr, dlscqp_lsc_row *rows_out);

/* The same for `n_dyn` dynamic (non-agent) obstacles: constant-velocity prediction (:293-303), size prediction with
 * constant acceleration (:338-368), normalVectorDynamicObs (:1129-1148), rows at :616-627.  Obstacle o gets slack
 * index o, obs index obs_base + o and the DYNAMIC flag.  Returns the row count, n_dyn * (M*(n+1) - phi). */
int oracle_generate_dlscgc_dynamic(const dlscqp_config *cfg, const float *init_self, double radius_self, int n_dyn,
                                   const float *dyn_position, const float *dyn_velocity, const double *dyn_radius,
                                   const double *dyn_max_acc, const double *dyn_downwash, int size_prediction,
                                   double uncertainty_horizon, int obs_base, dlscqp_lsc_row *rows_out);

/* ---------------------------------------------------------------------------------------------
 * Row N2: safe-flight-corridor boxes on a dense occupancy grid (oracle/dlsc_sfc_oracle.c; PARITY UNPINNED, see the
 * header of that file).  Boxes are {min x,y,z, max x,y,z} float32. */
typedef struct oracle_world oracle_world;
oracle_world *oracle_world_from_boxes(const double *boxes /*[n][6] centre, size*/, int n_boxes, double resolution,
                                      const float world_min[3], const float world_max[3]);
void oracle_world_free(oracle_world *w);
int oracle_sfc_obstacle_in_box(const oracle_world *w, const float bmin[3], const float bmax[3], double margin);
int oracle_sfc_expand(const oracle_world *w, const float init[6], double margin, double grid_resolution, double agent_max_vel,
                      double dt, float out[6]);
int oracle_sfc_from_hull(const oracle_world *w, const float *pts, int n_pts, const float *prev /*NULL or [6]*/, double margin,
                         double grid_resolution, double agent_max_vel, double dt, float out[6]);
int oracle_sfc_initialize(const oracle_world *w, const float pos[3], int M, double margin, double grid_resolution,
                          double agent_max_vel, double dt, float *sfcs /*[M][6]*/);
int oracle_sfc_construct(const oracle_world *w, const float *initial_traj /*[M][n+1][3]*/, int M, int n, const float goal[3],
                         const float waypoint[3], double margin, double grid_resolution, double agent_max_vel, double dt,
                         float *sfcs /*[M][6] in/out*/);

#ifdef __cplusplus
}
#endif
#endif
