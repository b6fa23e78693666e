/*
 * dlscqp.h -- C ABI of the MI355X-native batched DLSC-GC trajectory optimizer.
 *
 * This is the drop-in boundary for the ONE hot path of qwerty35/dlsc_gc_planner that this
 * project replaces: the per-agent Bernstein trajectory QP (TrajOptimizer::solve) and its
 * 1-variable goal LP (GoalOptimizer::solve), which the reference solves with CPLEX one agent
 * at a time.  All citations are file:line under the reference tree.
 *
 *   reference interface                                   replaced by
 *   ---------------------------------------------------   ---------------------------------
 *   TrajOptimizer::TrajOptimizer  (traj_optimizer.cpp:4)   dlscqp_create
 *   TrajOptimizer::updateParam    (traj_optimizer.cpp:167) dlscqp_update_param
 *   TrajOptimizer::solve          (traj_optimizer.cpp:18)  dlscqp_solve_batch[_device]
 *   GoalOptimizer::solve          (goal_optimizer.cpp:7)   dlscqp_goal_batch[_device]
 *   TrajPlanner::goalPlanning + trajOptimization
 *          (traj_planner.cpp:491-496, 737-744)             dlscqp_plan_batch[_device] (fused)
 *   throw PlanningReport::QPFAILED (traj_optimizer.cpp:152) status != DLSCQP_OK per agent
 *
 * Plain pointers and sizes only; no exceptions cross this boundary; buffers are caller-owned.
 * The C++ classes in dlsc_gc_planner_amd/include/dlsc_gc_planner_amd/ (MATP::TrajOptimizer,
 * MATP::GoalOptimizer, MATP::CollisionConstraints) are thin marshalling shims over these calls.
 */
#ifndef DLSCQP_H
#define DLSCQP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLSCQP_ABI_VERSION 2

/* Return codes of the API calls themselves (not per-agent verdicts). */
enum {
    DLSCQP_SUCCESS = 0,
    DLSCQP_ERR_INVALID_ARGUMENT = -1, /* std::invalid_argument in the reference (traj_optimizer.cpp:209,248) */
    DLSCQP_ERR_HIP = -2,              /* a HIP runtime call failed; see dlscqp_last_error() */
    DLSCQP_ERR_NO_DEVICE = -3,
    DLSCQP_ERR_UNSUPPORTED = -4,
    DLSCQP_ERR_SELFTEST = -5          /* dlscqp_selftest: the kernels of this build do not solve a known problem; see dlscqp_last_error() */
};

/* Per-agent verdicts written to out_status[]. */
enum {
    DLSCQP_OK = 0,             /* solved */
    DLSCQP_INFEASIBLE = 1,     /* CPLEX Infeasible / InfeasibleOrUnbounded -> PlanningReport::QPFAILED */
    DLSCQP_ITERATION_LIMIT = 2,/* solver did not converge (also mapped to QPFAILED by the shim)        */
    DLSCQP_NUMERICAL = 3,      /* non positive-definite reduced Hessian / breakdown (QPFAILED)         */
    DLSCQP_GOAL_FALLBACK = 16  /* goal LP: "numerical error" branch, goal = current_goal_point
                                  (goal_optimizer.cpp:79-81); not a failure                            */
};

/* Mirrors MATP::PlannerMode (sp_const.hpp:18-26), same order. */
enum {
    DLSCQP_PLANNER_DLSCGC = 0,
    DLSCQP_PLANNER_DLSC = 1,
    DLSCQP_PLANNER_LSC = 2,
    DLSCQP_PLANNER_BVC = 3,
    DLSCQP_PLANNER_ORCA = 4,
    DLSCQP_PLANNER_RECIPROCALRSFC = 5,
    DLSCQP_PLANNER_CIRCLETEST = 6
};

/* Mirrors MATP::SlackMode (sp_const.hpp:44-48), same order. */
enum {
    DLSCQP_SLACK_NONE = 0,
    DLSCQP_SLACK_CONTINUITY = 1, /* populatebyrow creates no extra variables for it; treated as NONE */
    DLSCQP_SLACK_COLLISIONCONSTRAINT = 2
};

/* The subset of MATP::Param (param.hpp) and MATP::Mission (mission.hpp:31) the path reads. */
typedef struct dlscqp_config {
    int32_t abi_version;           /* DLSCQP_ABI_VERSION */
    int32_t world_dimension;       /* param.world_dimension: 2 or 3                   */
    int32_t M;                     /* param.M  segments                               */
    int32_t n;                     /* param.n  polynomial degree (only 5)             */
    int32_t phi;                   /* param.phi (only 3)                              */
    int32_t phi_n;                 /* param.phi_n (1)                                 */
    int32_t planner_mode;          /* DLSCQP_PLANNER_*                                */
    int32_t slack_mode;            /* DLSCQP_SLACK_*                                  */
    int32_t world_use_octomap;     /* param.world_use_octomap: SFC rows on/off        */
    int32_t device;                /* HIP device ordinal                              */
    double dt;                     /* param.dt                                        */
    double control_input_weight;   /* param.control_input_weight                      */
    double terminal_weight;        /* param.terminal_weight                           */
    double slack_collision_weight; /* param.slack_collision_weight                    */
    double communication_range;    /* param.communication_range (<=0: rows omitted)   */
    double world_z_2d;             /* param.world_z_2d                                */
    double slack_threshold;        /* param.slack_threshold (collision alert)         */
    double world_min[3];           /* mission.world_min                               */
    double world_max[3];           /* mission.world_max                               */
} dlscqp_config;

/* The subset of MATP::Agent (sp_const.hpp:147-160) the path reads; float32 = octomap::point3d. */
typedef struct dlscqp_agent {
    float position[3];
    float velocity[3];
    float acceleration[3];
    float current_goal_point[3];
    float next_waypoint[3];
    int32_t id;
    int32_t n_obs;       /* constraints.getObsSize()                                              */
    int32_t n_slack_obs; /* obstacles that own slack variables: dynamic ones, or all of them in
                            DLSCQP_SLACK_COLLISIONCONSTRAINT mode (traj_optimizer.cpp:272-283)    */
    double max_vel;
    double max_acc;
    double radius;
    double nominal_velocity;
} dlscqp_agent;

/* One LSC half-space  normal . (c_{m,i} - point) - d >= 0  (collision_constraints.hpp:19-35),
 * flattened from CollisionConstraints::getLSC(oi, m, i). */
typedef struct dlscqp_lsc_row {
    float normal[3];
    float point[3];
    double d;
    int16_t m;     /* segment index 0..M-1                                                       */
    int8_t i;      /* control point index 0..n                                                   */
    uint8_t flags; /* bit0: obstacle is dynamic (type != AGENT, collision_constraints.cpp:582)   */
    int16_t obs;   /* obstacle index oi                                                          */
    int16_t slack; /* obs_slack_idx of this obstacle or -1 (traj_optimizer.cpp:413-450)          */
} dlscqp_lsc_row; /* 40 bytes */

#define DLSCQP_ROW_DYNAMIC 1

typedef struct dlscqp_ctx dlscqp_ctx;

/* ---- lifetime --------------------------------------------------------------------------- */
int dlscqp_create(const dlscqp_config *cfg, dlscqp_ctx **out_ctx);
void dlscqp_destroy(dlscqp_ctx *ctx);
/* TrajOptimizer::updateParam: replaces weights/modes/ranges; M, n, phi, dt, dimension stay as
 * constructed, exactly like the reference (traj_optimizer.cpp:167-169 does not refresh them). */
int dlscqp_update_param(dlscqp_ctx *ctx, const dlscqp_config *cfg);
/* Text of the last error on this thread ("" if none). */
const char *dlscqp_last_error(void);
/* Bytes of the smallest working set of one agent (= one wavefront) a launch of this shape can run with (the records of eight
 * active rows in it, the others in the spill records: dlscqp_device_batch.workspace); diagnostic.  Up to 160 KiB it lives in LDS;
 * above (a flag array of tens of thousands of rows per agent) the launch uses per-agent working sets in HBM owned by the
 * context -- slower per iteration, and launches of one context that need them must not overlap. */
int64_t dlscqp_lds_bytes(const dlscqp_ctx *ctx, int32_t max_slack_obs, int32_t max_rows_per_agent);

/* ---- host-buffer entry points (H2D + kernel + D2H, synchronous) --------------------------
 * rows of agent a are rows[row_offsets[a] .. row_offsets[a+1]); sfc_boxes is [n_agents][M][6]
 * = {min.x,min.y,min.z,max.x,max.y,max.z} per segment (CollisionConstraints::getSFC(m)), and
 * may be NULL when world_use_octomap == 0.
 * out_ctrl: [n_agents][M][n+1][3] float32 (2-D: z = world_z_2d, traj_optimizer.cpp:69-83)
 * out_cost: [n_agents] cplex.getObjValue() equivalent (traj_optimizer.cpp:109)
 * out_slack_cost: [n_agents][max_slack_obs] sum_m |eps_{s,m}| (traj_optimizer.cpp:85-106) or NULL
 * out_status: [n_agents] verdicts.  Outputs of failed agents are left untouched (ctrl) / 0.   */
int dlscqp_solve_batch(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents,
                       const int64_t *row_offsets, const dlscqp_lsc_row *rows,
                       const float *sfc_boxes, int32_t max_slack_obs, float *out_ctrl,
                       double *out_cost, double *out_slack_cost, int32_t *out_status);

/* TrajOptimizer::solve once more for agents whose verdict was DLSCQP_INFEASIBLE, this time with the certificate: out_conflict
 * [n_agents][conflict_stride >= dlscqp_conflict_doubles()] doubles, the records of dlscqp_device_batch.out_conflict (decode them with
 * dlscqp_conflict_decode).  What the reference does in its catch block (traj_optimizer.cpp:113-146). */
int dlscqp_solve_batch_conflict(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents, const int64_t *row_offsets,
                                const dlscqp_lsc_row *rows, const float *sfc_boxes, int32_t max_slack_obs, int32_t *out_status,
                                double *out_conflict, int32_t conflict_stride);

/* GoalOptimizer::solve for a batch: out_goal [n_agents][3]. */
int dlscqp_goal_batch(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents,
                      const int64_t *row_offsets, const dlscqp_lsc_row *rows,
                      const float *sfc_boxes, float *out_goal, int32_t *out_goal_status);

/* Fused goal LP -> trajectory QP (the QP uses the LP's goal, traj_planner.cpp:494). */
int dlscqp_plan_batch(dlscqp_ctx *ctx, int32_t n_agents, const dlscqp_agent *agents,
                      const int64_t *row_offsets, const dlscqp_lsc_row *rows,
                      const float *sfc_boxes, int32_t max_slack_obs, float *out_goal,
                      int32_t *out_goal_status, float *out_ctrl, double *out_cost,
                      double *out_slack_cost, int32_t *out_status);

/* ---- device-resident entry point ---------------------------------------------------------
 * Everything already in HBM, constraint rows in structure-of-arrays form so a wavefront reads
 * 64 consecutive rows with one coalesced load per field.  Asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the HIP null stream, exactly as for hipLaunchKernel);
 * no allocation, no synchronisation inside.                                                 */
struct dlscqp_advance;
typedef struct dlscqp_device_batch {
    int32_t n_agents;
    int32_t max_slack_obs;          /* stride of out_slack_cost                                 */
    int32_t do_goal;                /* 1: run the goal LP first and use its goal in the QP       */
    int32_t do_traj;                /* 1: run the trajectory QP                                  */
    int32_t max_rows_per_agent;     /* max over agents of row_offsets[a+1]-row_offsets[a]        */
    int32_t launch_offset;          /* plan only launch slots [launch_offset, launch_offset+launch_count); */
    int32_t launch_count;           /* 0 = all n_agents.  Lets a caller split one step into chunks on separate
                                       streams (e.g. to overlap the all-gather of a finished chunk with the
                                       stragglers of another).  All arrays keep their full-batch indexing.   */
    int32_t reserved_;
    const dlscqp_agent *agents;     /* [n_agents]                                               */
    const int32_t *row_offsets;     /* [n_agents+1]                                             */
    const float *nx, *ny, *nz;      /* [total_rows] normal                                      */
    const float *px, *py, *pz;      /* [total_rows] obstacle control point                      */
    const double *d;                /* [total_rows] margin                                      */
    const uint32_t *key;            /* [total_rows] DLSCQP_KEY(m,i,slack,dyn)                   */
    const float *sfc_boxes;         /* [n_agents][M][6] or NULL                                 */
    float *out_goal;                /* [n_agents][3] (may be NULL if !do_goal)                  */
    int32_t *out_goal_status;       /* [n_agents]                                               */
    float *out_ctrl;                /* [n_agents][M][n+1][3]                                    */
    double *out_ctrl_f64;           /* optional [n_agents][dim*M*(n+1) + M*max_slack_obs], NULL */
    double *out_cost;               /* [n_agents]                                               */
    double *out_slack_cost;         /* [n_agents][max_slack_obs] or NULL                        */
    int32_t *out_status;            /* [n_agents]                                               */
    int32_t *out_iters;             /* optional [n_agents][4] {iterations, adds, drops, active} */
    const int32_t *agent_order;     /* optional [n_agents] permutation: workgroup b plans agent agent_order[b].
                                       Agents differ in active-set iterations by 50x; launching the expensive ones
                                       first (e.g. by the previous step's out_iters) removes the tail of the launch.
                                       NULL = identity.  Results do not depend on it.                            */
    uint64_t *debug_stamps;         /* NULL; [n_agents][12] cycle stamps, read only by the separate
                                       -DDLSCQP_STAMPS diagnostic build (profiles/stamps.py)        */
    const int32_t *row_counts;      /* optional [n_agents]: agent a has row_counts[a] rows starting at row_offsets[a]
                                       (a generator that pruned rows, dlscqp_lsc_gen.prune); NULL: row_offsets[a+1] -
                                       row_offsets[a].  max_rows_per_agent bounds both.                               */
    /* ---- round 4 (appended: a caller that zero-fills the structure gets the behaviour of before) ---------------- */
    int32_t row_pitch;              /* > 0: every agent's rows start at a * row_pitch (neighbour lists of fixed pitch, the
                                       synthetic swarms): the wavefront computes the address of its rows instead of loading
                                       it, one dependent load less in front of the rows.  Agent a then has row_counts[a]
                                       rows, or row_pitch of them when row_counts is NULL; row_offsets is not read.      */
    int32_t hint_steps;             /* 0 = none, else 1..3 entries of hint_iters                                        */
    const int32_t *hint_iters[3];   /* LAUNCH ORDER INSIDE THE KERNEL, no sort between two launches: out_iters ([n_agents][4])
                                       of the previous replanning steps of the same agents, newest first.  The agents are
                                       taken in blocks of 64 consecutive ids; launch slot s plans the (s / n_blocks)-th
                                       most expensive agent of block s % n_blocks -- expensive by the key of
                                       dlscqp_launch_order_history_device, max_k (3/4)^k iterations_k, in eight classes
                                       (0, 1, 2-3, 4-7, ... 64+) -- which its wavefront finds with eight ballots over the
                                       block's 64 keys.  The first n_blocks workgroups therefore plan every block's
                                       hardest agent, the next n_blocks the second hardest ...: the order a global sort
                                       gives, up to ties between blocks, without the sort's two kernels and their scratch.
                                       Used when agent_order is NULL; not with launch_offset / launch_count.  Results
                                       do not depend on it.                                                             */
    int32_t out_record_stride;      /* > 0: out_ctrl, out_goal and out_status are fields of ONE record per agent, this many
                                       4-byte words apart from agent to agent (agent a's control points at out_ctrl + a * stride,
                                       its sub-goal at out_goal + a * stride, its verdict at out_status + a * stride) -- the plans
                                       are written in the layout the exchange sends (dlscqp_exchange_post_records: control points |
                                       sub-goal | verdict, M (n+1) 3 + 4 words), no packing pass in between.  0: three dense arrays. */
    int32_t reserved2_;
    /* ---- round 5 (appended, zero = the behaviour of before) -------------------------------------------------------------- */
    void *workspace;                /* optional device buffer of workspace_bytes bytes for the launch's SPILL RECORDS: an agent's
                                       working set keeps the records of its first 8 or 16 active rows in LDS (the inverse factor's and
                                       the orthonormalised normals' columns, csrc/gi_core.hpp: MemLayout); the few agents of a large
                                       launch that activate more rows keep the others here, in HBM.  dlscqp_workspace_bytes() says how
                                       much a launch can need.  NULL (or too small): the library uses a buffer of the CONTEXT, grown
                                       on demand -- which launches of the same context on different streams would share (do not
                                       overlap them), which cannot grow inside a stream capture, and whose growth LATER (another
                                       batch of the context, a larger launch) frees the buffer a graph captured earlier still
                                       points at: give a batch that is captured into a graph a workspace of its own.  Launches small
                                       enough to give every agent the LDS for all its records need neither.  Contents are scratch. */
    int64_t workspace_bytes;
    const uint64_t *hint_mask_in;   /* LAUNCH ORDER FROM CLASS MASKS (preferred over hint_iters: no load of 64 keys in front of every agent's
                                       record).  Three tables of [ceil(n_agents / 64)][8] 64-bit words.  A launch with hint_mask_out
                                       leaves, per block of 64 consecutive agents, eight masks -- one per class of iteration counts
                                       (0, 1, 2-3, 4-7, ... 64+; an agent's class is that of this step's count or one below the class
                                       it was scheduled with, whichever is higher) -- in which every planned agent has set its bit.
                                       The NEXT step's launch of the same agents names that table as hint_mask_in: launch slot s then
                                       plans the (s / n_blocks)-th most expensive agent of block s % n_blocks, as with hint_iters.  A
                                       table that is not a partition of its block's agents (all zero: the first step) gives id order.
                                       hint_mask_zero: a table this launch clears -- hint_mask_out must be all zero when its launch
                                       starts, so name here the table a LATER launch will write (three tables in rotation: in = what
                                       the previous launch wrote, out, zero = the next launch's out).  Used when agent_order is NULL
                                       and hint_steps is 0.  With launch_offset / launch_count the range counts LAUNCH SLOTS of the
                                       64 n_blocks: slots [0, R n_blocks) are every block's R most expensive agents, which a caller may
                                       plan in a launch of their own beside the others' (small enough for helper wavefronts: the
                                       flown crowd).  Results do not depend on it.                                                 */
    uint64_t *hint_mask_out;
    uint64_t *hint_mask_zero;
    const struct dlscqp_advance *advance_after; /* optional: the bookkeeping between two launches (dlscqp_advance, below) done by the
                                       plan kernel itself -- the wavefront that planned an agent applies the structure to that agent
                                       right behind its final stores: one launch less per replanning step (a ten-agent step is four
                                       launches of a few microseconds each).  The structure is read when the launch is queued; its
                                       status / goal_status / ctrl / goal (/ alert_slack_cost) must be this batch's dense outputs
                                       and its n_agents this batch's; with do_goal and do_traj.  Same results as
                                       dlscqp_advance_device after the launch (tests/test_gpu_closed_loop.py).                   */
    /* ---- round 6 (appended, zero = the behaviour of before) -------------------------------------------------------------- */
    double *out_conflict;           /* optional [n_agents][conflict_stride] doubles: for every agent whose trajectory QP is
                                       DLSCQP_INFEASIBLE the rows that PROVE it and their multipliers -- what CPLEX's conflict refiner
                                       prints in the reference (traj_optimizer.cpp:118-146).  Record: [0] members n (0: no certificate:
                                       the agent is not infeasible, or failed otherwise), [1] the violation (metres) of the row no
                                       step could repair, [2] that row's factor to the reference's scaling, [3] 0, then n pairs
                                       (row id, multiplier >= 0): ids as type << 24 | index << 3 | sub -- type 0: row sub = 2 kind + side
                                       (kind 0 bound / box face, 1 velocity, 2 acceleration; side 0 lower) of control-point coordinate
                                       index = axis P + point; 1: row `index` of the agent's LSC list; 2: side `sub` of structured
                                       slot `index` (bound, velocity, acceleration, communication range, waypoint range); 3: the
                                       bound eps <= 0 of slack variable `index`.  dlscqp_conflict_decode() turns a record into rows of
                                       the reference's model.  The multipliers are those of the rows in the KERNEL's scaling (metres);
                                       with them sum lambda_i a_i vanishes on the null space of the equality rows and
                                       sum lambda_i b_i exceeds what the equalities allow by [1] (Farkas).  While certificates are
                                       asked for the kernel's early infeasibility proof by weak duality is off (same verdicts, a
                                       slower launch): a debugging aid like the reference's, not for the timed path.               */
    int32_t conflict_stride;        /* doubles per agent, >= dlscqp_conflict_doubles()                                            */
    int32_t reserved3_;
} dlscqp_device_batch;

/* key layout: bits 0-7 m, 8-11 i, 12 dynamic flag, 16-31 slack index (0xFFFF = none). */
#define DLSCQP_KEY(m, i, slack, dyn)                                                     \
    ((uint32_t)((m) & 0xFF) | ((uint32_t)((i) & 0xF) << 8) | ((uint32_t)((dyn) & 1) << 12) | \
     ((uint32_t)((slack) < 0 ? 0xFFFFu : (uint32_t)(slack)) << 16))

int dlscqp_plan_batch_device(dlscqp_ctx *ctx, const dlscqp_device_batch *batch, void *stream);
/* doubles of one agent's record of dlscqp_device_batch.out_conflict: 4 + 2 (dim f + M max_slack_obs + 1) */
int32_t dlscqp_conflict_doubles(const dlscqp_ctx *ctx, int32_t max_slack_obs);

/* ---- the reference's failure path: model export and conflict (traj_optimizer.cpp:45-52, 113-146) -- host only, one agent ------------
 * Both work on the agent's model as TrajOptimizer::populatebyrow (:225-527) builds it -- same variables (x_m_i, y_m_i, z_m_i,
 * epsilon_slack_col_oi_m), same rows in the same order -- from what the host-buffer entry points take: the agent's record, its LSC rows
 * in populatebyrow's order (obstacle, segment, control point), its M corridor boxes (NULL without world_use_octomap).
 * goal: the point the terminal cost pulls to, float[3] -- the goal LP's answer in the fused launch (traj_planner.cpp:494); NULL =
 * the record's current_goal_point. */
enum {  /* kinds of rows, dlscqp_conflict_member.kind */
    DLSCQP_MODEL_EQ = 0,       /* initial state, continuity, stop at the end of the horizon (:333-381, :515-524) */
    DLSCQP_MODEL_SFC = 1,      /* corridor box face (:385-410)                                                   */
    DLSCQP_MODEL_LSC = 2,      /* DLSC-GC / dynamic-obstacle half-space (:412-450)                               */
    DLSCQP_MODEL_VEL = 3,      /* velocity limit (:453-468)                                                      */
    DLSCQP_MODEL_ACC = 4,      /* acceleration limit (:469-485)                                                  */
    DLSCQP_MODEL_COMM = 5,     /* communication range between segment ends and starts (:490-503)                 */
    DLSCQP_MODEL_WAYPOINT = 6, /* communication range around the waypoint (:505-512)                             */
    DLSCQP_MODEL_VAR_LB = 7,   /* a variable's lower bound (:251-265)                                            */
    DLSCQP_MODEL_VAR_UB = 8    /* ... upper bound (:251-265; a slack variable's eps <= 0, :276)                  */
};
typedef struct dlscqp_conflict_member {
    int32_t ref_index;  /* the row's number in the reference's `infeas` array (traj_optimizer.cpp:118-127): rows of `con` in populatebyrow's
                           order (what the refiner prints as c<i>), then for every variable v its lower (n_con + 2 v) and upper (+ 1) bound */
    int32_t kind;       /* DLSCQP_MODEL_*                                                                                          */
    int32_t n;          /* entries (<= 4)                                                                                          */
    int32_t var[4];     /* variables in the reference's order: axis offset_dim + segment offset_seg + point; slack: offset_slack_col + ... */
    double coef[4];     /* in the reference's row scaling                                                                          */
    double lo, hi;      /* lo <= sum coef x <= hi, one of them +-1e300 (none)                                                      */
    double lambda;      /* multiplier of the row as written here, >= 0 (of "sum coef x >= lo", or of "-sum coef x >= -hi")          */
} dlscqp_conflict_member;
/* Writes the agent's model as a CPLEX-LP file (cplex.exportModel). */
int dlscqp_write_lp(const dlscqp_ctx *ctx, const dlscqp_agent *agent, const dlscqp_lsc_row *rows, int32_t n_rows, const float *sfc_boxes,
                    int32_t max_slack_obs, const float *goal, const char *path);
/* One record of dlscqp_device_batch.out_conflict (copied to the host) -> the rows of the reference's model that prove the agent's QP
 * infeasible, with their multipliers: sum lambda_i (row_i as ">=") is a combination of the equality rows alone, and its right-hand
 * side exceeds theirs by *gap_out > 0 (Farkas).  n_out: members written (0: the record holds no certificate); *n_con_out: rows of
 * `con` (for the refiner's numbering of the variable bounds). */
int dlscqp_conflict_decode(const dlscqp_ctx *ctx, const dlscqp_agent *agent, const dlscqp_lsc_row *rows, int32_t n_rows, const float *sfc_boxes,
                           int32_t max_slack_obs, const float *goal, const double *record, dlscqp_conflict_member *out, int32_t capacity,
                           int32_t *n_out, int32_t *n_con_out, double *gap_out);
/* bytes of dlscqp_device_batch.workspace that suffice for any launch of up to n_agents agents of this shape (0: none needed) */
int64_t dlscqp_workspace_bytes(const dlscqp_ctx *ctx, int32_t n_agents, int32_t max_slack_obs, int32_t max_rows_per_agent);

/* Launch order for the NEXT replanning step from the iteration counts of the one just planned (out_iters of
 * dlscqp_device_batch, 4 int32 per agent): order_out[b] = the agent workgroup b plans, agents with the most active-set
 * iterations first, ties in no particular order -- pass it as dlscqp_device_batch.agent_order.  An agent that was hard to plan
 * is usually hard again 0.2 s later, and a launch is as long as the agent that finishes last; the order changes when agents
 * run, never what they compute.  Device buffers (iters [n_agents][4], order_out [n_agents]); runs on `stream` like the other
 * *_device entry points (calls on one context must be ordered, like its launches: they share a scratch block); the reference
 * has no counterpart (its agents are planned one after the other, multi_sync_simulator.cpp:516-524). */
int dlscqp_launch_order_device(dlscqp_ctx *ctx, const int32_t *iters, int32_t n_agents, int32_t *order_out, void *stream);
/* The same from the counts of up to three steps, newest first (iters[k]: out_iters of step t - k): the key is
 * max_k (3/4)^k iterations_k.  An agent that was hard two or three steps ago is likely to be hard again, and one long agent
 * that starts late is what a launch waits for (measured on the S64 loop: 0.35 -> 0.28-0.33 ms per 32 768-agent step). */
int dlscqp_launch_order_history_device(dlscqp_ctx *ctx, const int32_t *const *iters, int32_t n_steps, int32_t n_agents, int32_t *order_out,
                                       void *stream);

/* The launch order a dlscqp_device_batch with these hint_iters gets, written out (tests, diagnostics): order_out[s] = the agent
 * launch slot s plans, s < 64 * ceil(n_agents / 64), or -1 for a slot beyond its block's agents.  Same device function as the
 * plan kernel's.  iters: 1..3 device arrays [n_agents][4], newest first; order_out: device array. */
int dlscqp_hint_order_preview_device(dlscqp_ctx *ctx, const int32_t *const *iters, int32_t n_steps, int32_t n_agents, int32_t *order_out,
                                     void *stream);
/* the same for hint_mask_in: masks [ceil(n_agents / 64)][8] */
int dlscqp_mask_order_preview_device(dlscqp_ctx *ctx, const uint64_t *masks, int32_t n_agents, int32_t *order_out, void *stream);

/* ---- DLSC-GC half-space generation on the device --------------------------------------
 * TrajPlanner::generateDLSCGC for neighbouring AGENTS (traj_planner.cpp:603-666) together with the shift of the
 * previous plans that precedes it (initialTrajPlanningPrevSol / obstaclePredictionWithPrevSol, :409-421,
 * :304-314; constant-velocity trajectories on the first replanning step, :293-303, :410-411).  Input is the
 * table every rank holds after the per-step all-gather (previous plans + sub-goals of ALL agents) and a
 * neighbour list per local agent; output are the SoA rows dlscqp_plan_batch_device reads, so the rows never
 * cross PCIe.  Local agent k's rows are [per*nbr_offsets[k], per*nbr_offsets[k+1]), per = M*(n+1) - phi, in
 * populatebyrow's order (neighbour, segment, control point).  Asynchronous on `stream`. */
typedef struct dlscqp_lsc_gen {
    int32_t n_local;            /* agents to generate rows for                                      */
    int32_t first_plan;         /* 1: planner_seq < 2, constant-velocity trajectories from position/velocity */
    const int32_t *local_ids;   /* [n_local] ids into the tables, NULL = 0..n_local-1               */
    const float *plans;         /* [n_total][M][n+1][3] previous plans (unused when first_plan)     */
    const float *position;      /* [n_total][3] (first_plan only)                                   */
    const float *velocity;      /* [n_total][3] (first_plan only)                                   */
    const float *goals;         /* [n_total][3] current_goal_point of every agent                   */
    const double *radius;       /* [n_total]                                                        */
    const double *downwash;     /* [n_total]                                                        */
    const int32_t *nbr_offsets; /* [n_local+1]                                                      */
    const int32_t *nbr_ids;     /* [nbr_offsets[n_local]] neighbour ids, = obstacle order oi        */
    float *nx, *ny, *nz, *px, *py, *pz; /* outputs, [per * (nbr_offsets[n_local] + n_local * n_dyn)]  */
    double *d;
    uint32_t *key;
    /* Dynamic (non-agent) obstacles, seen by every local agent: generateDLSCGC's branch for
     * ObstacleType != AGENT (traj_planner.cpp:616-627) with normalVectorDynamicObs (:1129-1148), the
     * constant-velocity prediction (:293-303) and obstacleSizePredictionWithConstAcc (:338-368).  Local agent k's
     * rows are [per*(nbr_offsets[k] + k*n_dyn), per*(nbr_offsets[k+1] + (k+1)*n_dyn)): its neighbour agents
     * first, then the dynamic obstacles in order; obstacle s uses slack variable s and the DYNAMIC flag.
     * n_dyn = 0 (and NULL pointers) keeps the agents-only behaviour. */
    int32_t n_dyn;
    int32_t dyn_size_prediction;      /* param.obs_size_prediction                                       */
    double dyn_uncertainty_horizon;   /* param.obs_uncertainty_horizon                                   */
    const float *dyn_position;        /* [n_dyn][3]                                                      */
    const float *dyn_velocity;        /* [n_dyn][3]                                                      */
    const double *dyn_radius;         /* [n_dyn]                                                         */
    const double *dyn_max_acc;        /* [n_dyn]                                                         */
    const double *dyn_downwash;       /* [n_dyn]                                                         */
    /* TrajPlanner::checkObstacleDisturbance (traj_planner.cpp:329-336): a neighbour whose predicted trajectory starts
     * farther than reset_threshold (param.reset_threshold) from where it is observed is predicted to stand still at the
     * observed position.  NULL = no check (ideal simulation: prediction and state coincide).                      */
    const float *observed_position;   /* [n_total][3] or NULL                                            */
    double reset_threshold;
    /* Presolve (optional).  With prune_sfc_boxes given ([n_local][M][6], the corridor boxes of the local agents for THIS step;
     * NULL boxes are allowed when world_use_octomap == 0: then only the world box is used), a half-space that is implied, with a
     * margin of 0.1 mm, by the rows that confine its control point to a box -- variable bounds and corridor box,
     * traj_optimizer.cpp:251-265, 385-410 -- is not written at all: it can be neither active nor violated, so goal LP and QP
     * have the same solutions without it (in a crowd most neighbours within communication range are far outside an agent's
     * corridor).  The rows that remain are written contiguously from the agent's first row on, in their original order, and
     * out_row_counts[k] says how many: pass it as dlscqp_device_batch.row_counts.  prune != 0 switches it on.          */
    int32_t prune;
    int32_t reserved_;
    const float *prune_sfc_boxes;
    int32_t *out_row_counts;          /* [n_local], required when prune != 0                              */
    /* Neighbour lists of fixed pitch (dlscqp_neighbours_device): when nbr_counts is given, local agent k has nbr_counts[k]
     * neighbours from nbr_ids[nbr_offsets[k]] on (nbr_offsets[k] = k * pitch), not nbr_offsets[k+1] - nbr_offsets[k].   */
    const int32_t *nbr_counts;        /* [n_local] or NULL                                                */
    /* Synthetic workloads (BASELINE.json configs[1], "120 DLSC inequalities each"): the neighbours at list positions >= tail_from
     * contribute only their last tail_segments segments (dlsc_gc_planner_amd/workloads.py: four neighbours with all M segments,
     * a fifth with two: 4 x 27 + 12 rows).  tail_segments == 0: off.  Only with prune != 0 (the rows are written contiguously and
     * counted in out_row_counts); the reference has no counterpart. */
    int32_t tail_from;
    int32_t tail_segments;
    int32_t max_neighbours;           /* optional: the longest neighbour list (0 = unknown): lists of a few neighbours get one
                                         wavefront per agent instead of two                                              */
    int32_t reserved2_;
} dlscqp_lsc_gen;

int dlscqp_generate_lsc_device(dlscqp_ctx *ctx, const dlscqp_lsc_gen *gen, void *stream);

/* Self-test of the plan kernels, run by dlscqp_create (DLSCQP_NO_SELFTEST=1 in the environment skips it there).  The
 * kernels keep ~100 registers of per-lane state across a data-dependent loop and read other lanes' registers; twice a
 * developer-only compiler configuration has produced an instantiation that returned wrong plans ("NO
 * DIVERGENCE").  So that a miscompiling toolchain fails loudly on the user's machine and not silently in flight: a wall
 * problem of the context's own shape (an agent drifting at a half-space that it reaches even when it brakes as gently as
 * it can, plus an agent that is free to go) is planned by every kernel instantiation a batch of this context can be handed -- the one
 * select_instance picks with all rows in the register cache, the one for longer row lists, the one for batches with slack
 * obstacles, the runtime-shape one, the variant with its working set in HBM; each of them in every BUILD it exists in: the standard one
 * that launches of more than a thousand agents run (two wavefronts per SIMD), the one-wavefront-per-SIMD build of small
 * launches, the build with helper wavefronts -- and every answer must (1) report DLSCQP_OK,
 * (2) keep the initial state, continuity to the second derivative and the terminal stop, (3) satisfy the wall, the box and
 * the velocity limit, touch the wall (agent 0) or pass where it would be (agent 1), (4) agree with the other
 * instantiations' to 1e-6 m.  Returns DLSCQP_SUCCESS or DLSCQP_ERR_SELFTEST.
 * A few launches of two agents: well under a millisecond after the first launch of the process.                        */
int dlscqp_selftest(dlscqp_ctx *ctx);

/* The communication-range filter of MultiSyncSimulator::broadcastMsgs (multi_sync_simulator.cpp:482-503) on the device:
 * local agent k is told about every other agent qj with LInfinityDistance(position_k, position_qj) <= communication_range
 * (util.hpp:131-140: float32 differences, compared in double), in ascending id order -- the obstacle order the
 * half-space generator and populatebyrow walk.  A uniform grid with cells of one communication range (over the world box of
 * the configuration) bins the agents; every local agent then searches the 27 cells around its own.  Lists have a fixed
 * pitch, so nothing is scanned or sized on the host: out_nbr_ids[k * pitch ...] holds out_nbr_counts[k] ids; a list longer
 * than `pitch` is cut to its `pitch` smallest ids and *out_overflow is set.  communication_range <= 0 (no limit): every
 * other agent.  Asynchronous on `stream`; scratch memory belongs to the context (calls of one context must not overlap).
 * Up to 256 agents (every mission of the reference) there are no cells: one launch in which every agent is a candidate of
 * every other -- same test, same order, same lists, no scratch memory, and safe to capture into a HIP graph (the search
 * through cells is not: a captured step with it faulted on replay). */
typedef struct dlscqp_neighbour_search {
    int32_t n_total;            /* agents in the position table                                         */
    int32_t n_local;            /* agents to build lists for                                            */
    const int32_t *local_ids;   /* [n_local] ids into the table, NULL = 0..n_local-1                    */
    const float *position;      /* position of agent a: position[a * position_stride + 0..2]            */
    int32_t position_stride;    /* in floats: 3 for a dense [n_total][3] table, 26 for the dlscqp_agent records */
    int32_t pitch;              /* capacity of one list                                                 */
    int32_t *out_nbr_ids;       /* [n_local][pitch]                                                     */
    int32_t *out_nbr_counts;    /* [n_local]                                                            */
    int32_t *out_row_counts;    /* optional [n_local]: rows the half-space generator writes without its presolve,
                                 * (count + n_dyn) * (M (n+1) - phi)                                     */
    int32_t n_dyn;
    int32_t reserved_;
    int32_t *out_overflow;      /* optional: set to 1 when a list was cut (never cleared here)          */
} dlscqp_neighbour_search;

int dlscqp_neighbours_device(dlscqp_ctx *ctx, const dlscqp_neighbour_search *ns, void *stream);

/* The neighbour rule of the synthetic swarms (dlsc_gc_planner_amd/workloads.py: _neighbours; no counterpart in the reference,
 * whose rule is the communication range above): the agents in blocks of `group` consecutive ids (a swarm), every agent's list =
 * the k nearest agents of its own block (Euclidean distance of the float32 positions, computed in double; nearest first, ties by
 * id).  Lists of fixed pitch like dlscqp_neighbours_device's: out_nbr_ids[a * pitch ...], out_nbr_counts[a] = min(k, block - 1).
 * One wavefront per agent.  Asynchronous on `stream`. */
typedef struct dlscqp_neighbour_knn {
    int32_t n_total;            /* agents                                                               */
    int32_t group;              /* agents per block (a swarm)                                           */
    int32_t k;                  /* neighbours per agent, <= pitch                                        */
    int32_t pitch;              /* capacity of one list                                                  */
    const float *position;      /* position of agent a: position[a * position_stride + 0..2]            */
    int32_t position_stride;    /* in floats                                                            */
    int32_t reserved_;
    int32_t *out_nbr_ids;       /* [n_total][pitch]                                                     */
    int32_t *out_nbr_counts;    /* [n_total]                                                            */
} dlscqp_neighbour_knn;
int dlscqp_neighbours_knn_device(dlscqp_ctx *ctx, const dlscqp_neighbour_knn *nk, void *stream);

/* TrajPlanner::checkWaypointTrap (traj_planner.cpp:708-735), after the half-spaces and the corridor boxes of a step and
 * before the goal LP: if the agent's sub-goal or waypoint lies outside the feasible region of its last control point
 * (CollisionConstraints::isPointInFeasibleRegion, collision_constraints.cpp:586-598: half-spaces of the neighbour AGENTS at
 * (M-1, n), the last corridor box, the communication-range box around the waypoint), the rows of every dynamic obstacle
 * that can reach the waypoint within the horizon (Obstacle::isCollided, obstacle.hpp:26-36) are replaced by the empty
 * half-space (zero normal), which the LP and the QP skip (traj_optimizer.cpp:422-424).  `gen` is the structure the rows
 * were generated with (row layout, neighbour lists, dynamic obstacles).  One wavefront per local agent.             */
typedef struct dlscqp_waypoint_trap {
    const dlscqp_agent *agents;   /* [gen->n_local] records of the local agents: current_goal_point, next_waypoint, radius */
    const float *sfc_boxes;       /* [gen->n_local][M][6] or NULL (world_use_octomap == 0)                   */
    int32_t *out_released;        /* optional [gen->n_local]: number of dynamic obstacles whose rows were emptied */
} dlscqp_waypoint_trap;
int dlscqp_waypoint_trap_device(dlscqp_ctx *ctx, const dlscqp_lsc_gen *gen, const dlscqp_waypoint_trap *trap, void *stream);

/* ---- what happens between two launches, on the device ----------------------------------------
 * One kernel for the bookkeeping that follows dlscqp_plan_batch_device in a closed loop, so that a replanning step is
 * three or four launches (half-spaces, corridors, LP + QP, this) and can be replayed as one HIP graph:
 *   - fail-safe: an agent whose QP failed keeps its previous plan shifted by one segment (TrajPlanner::plan,
 *     traj_planner.cpp:776 with the shift of initialTrajPlanningPrevSol, :412-421);
 *   - the sub-goal becomes the goal LP's result unless the LP was infeasible (traj_planner.cpp:493-495);
 *   - the agent moves on by one replanning period: position / velocity / acceleration := the state of the new plan
 *     at time dt, i.e. the start of segment 1, from float32 control-point differences (AgentManager::doStep,
 *     agent_manager.cpp:51-57, 224-227; Trajectory::getStateAt, trajectory.cpp:110-199);
 *   - the agent records of the NEXT step are written (position, velocity, acceleration, current_goal_point and,
 *     if given, next_waypoint), and the next step's initial trajectory (the new plan shifted by one segment), which
 *     the corridor generator reads.
 * `plans` and `current_goal` are updated in place.  Asynchronous on `stream`. */
typedef struct dlscqp_advance {
    int32_t n_agents;
    const int32_t *status;       /* [n] out_status of the launch                                          */
    const int32_t *goal_status;  /* [n] out_goal_status                                                   */
    const float *ctrl;           /* [n][M][n+1][3] out_ctrl                                               */
    const float *goal;           /* [n][3] out_goal                                                       */
    float *plans;                /* [n][M][n+1][3] in: previous plans, out: the plans of this step        */
    float *current_goal;         /* [n][3] in/out                                                         */
    float *initial_traj;         /* optional [n][M][n+1][3] out: the new plans shifted by one segment     */
    dlscqp_agent *agents;        /* [n] in/out: the five float3 fields are rewritten for the next step    */
    const float *waypoints;      /* optional [n][3] -> next_waypoint                                      */
    /* without a waypoint planner (no `waypoints`): next_waypoint := the point `waypoint_step` metres ahead on the
     * straight line from the new position to desired_goal, desired_goal itself when it is nearer (optional) */
    const float *desired_goal;   /* optional [n][3]                                                       */
    double waypoint_step;
    int32_t waypoint_hold_on_lp_failure; /* (with desired_goal) an agent whose goal LP was infeasible gets its sub-goal as the next
                                            waypoint for one step: the synthetic swarms' rule (workloads.generate)          */
    int32_t reserved_;
    /* optional (with initial_traj): the next step's boxes of a world WITHOUT obstacles -- per segment the bounding box of the
     * initial trajectory's control points, `box_margin` wider on every side, cut to [box_min, box_max] (float32 arithmetic) --
     * written where dlscqp_device_batch.sfc_boxes reads them: [n][M][6] {min x,y,z, max x,y,z}                              */
    float *out_boxes;
    float box_margin;
    float box_min[3], box_max[3];
    int32_t reserved2_;
    /* optional: what a HOST-side waypoint planner reads before the next step (MultiSyncSimulator::decentralizedMAPP,
     * multi_sync_simulator.cpp:505-560: positions, sub-goals, previous plans, collision alerts), written by this kernel straight
     * into pinned host memory (hipHostMalloc: the device writes it over the bus), so that the step needs no device-to-host copy
     * of its own -- the host reads it once `stream` has run dry.  float32, in this order:
     *   position [n][3] | current_goal [n][3] | plans [n][M][n+1][3] | alert [n][n_alert_obs]
     * alert = 1.0f where the agent's slack cost of that obstacle in the last QP exceeded alert_threshold (TrajOptimizer::solve's
     * collision alert, traj_optimizer.cpp:85-106; compared in double), else 0.0f; absent when n_alert_obs is 0.            */
    float *out_host_pack;
    const double *alert_slack_cost; /* [n][alert_slack_stride] out_slack_cost of the launch (with n_alert_obs > 0)        */
    int32_t alert_slack_stride;
    int32_t n_alert_obs;
    double alert_threshold;
} dlscqp_advance;

int dlscqp_advance_device(dlscqp_ctx *ctx, const dlscqp_advance *adv, void *stream);

/* Row N2 of SURVEY.md section 8(f): safe-flight-corridor (SFC) boxes on the device.
 * A world is a list of axis-aligned boxes (the rows of the reference's world CSV: centre x,y,z, size x,y,z,
 * MapManager::updateOctreeFromCSV, map_manager.cpp:264-316) rasterised to an occupancy grid of `resolution` metres
 * (param.world_resolution) together with, for every cell, the occupied cells nearest to it within 1 m -- what the
 * reference asks octomap + DynamicEDT3D for (DynamicEDTOctomap::getDistanceAndClosestObstacle).  Built once on the
 * host, resident in HBM afterwards. */
typedef struct dlscqp_world dlscqp_world;
int dlscqp_world_create(dlscqp_ctx *ctx, const double *boxes /*[n_boxes][6]*/, int32_t n_boxes, double resolution,
                        const float world_min[3], const float world_max[3], dlscqp_world **out);
void dlscqp_world_destroy(dlscqp_world *world);
/* Optional, synchronous: build the per-radius-class tables (summed volumes of the lattice-point table) that the agents of the
 * following dlscqp_generate_sfc_device launches need -- `radius` is a HOST array of the radii that will occur.  Without this call
 * the first launch with a given (radius pointer, count) builds them itself: it synchronises with its stream once, reads the radii
 * back and allocates (never inside a stream capture) -- and does not notice radii that change behind the same pointer.  After this
 * call launches no longer look at their radii: an agent whose class has no table tests its boxes point by point (same boxes,
 * slower).  Tables are bounded (1 GiB each, 4 GiB per world); on an allocation or upload error nothing is left half-built. */
int dlscqp_world_prepare(dlscqp_world *world, const double *radius /*[n_agents], host*/, int32_t n_agents);

/* One replanning step of CollisionConstraints' corridor logic for n_agents agents, device pointers:
 *   initialize != 0: initializeSFC (collision_constraints.cpp:435-452) from `position`; out_code 0 = ok, -1 = the start
 *                    cell is blocked (the reference throws std::invalid_argument);
 *   initialize == 0: constructSFCFromInitialTraj (:502-536): boxes shift by one segment, the last box is grown from
 *                    {last point of initial_traj, current goal, next waypoint}; out_code 0 = that box, 1 = the box grown
 *                    inside the previous one without the waypoint, 2 = previous box kept.
 * sfcs: [n_agents][M][6] float32 {min x,y,z, max x,y,z}, in/out; feeds dlscqp_device_batch.sfc_boxes directly.
 * The world keeps one summed-volume table per agent radius class (what makes a box test eight reads); which classes there are
 * is only known from `radius`, which lives in device memory: the FIRST call with a given (radius pointer, n_agents)
 * synchronises with `stream`, reads the radii and builds the tables that are missing -- later calls are asynchronous like
 * every other *_device entry.  Inside a stream capture nothing is built (agents without their table grow their boxes the slow
 * way, same boxes): run one step before capturing.  Launches on ONE world from several host threads have to be serialised by
 * the caller (the table bookkeeping is not locked).  One workgroup of 256 threads per agent. */
typedef struct dlscqp_sfc_gen {
    int32_t n_agents;
    int32_t initialize;
    double grid_resolution;      /* param.grid_resolution                                  */
    const double *radius;        /* [n_agents] margin of the corridor = agent radius       */
    const double *max_vel;       /* [n_agents]                                             */
    const float *position;       /* [n_agents][3]            (initialize)                  */
    const float *initial_traj;   /* [n_agents][M][n+1][3]    (construct)                   */
    const float *goals;          /* [n_agents][3] current_goal_point (construct)           */
    const float *waypoints;      /* [n_agents][3] next_waypoint      (construct)           */
    float *sfcs;                 /* [n_agents][M][6]                                       */
    int32_t *out_code;           /* [n_agents]                                             */
} dlscqp_sfc_gen;
int dlscqp_generate_sfc_device(dlscqp_ctx *ctx, const dlscqp_world *world, const dlscqp_sfc_gen *gen, void *stream);

/* ---- the decentralised waypoint planner (host only, no device work) -----------------------------
 * MultiSyncSimulator::decentralizedMAPP (multi_sync_simulator.cpp:308-466): communication groups, per group
 * GridBasedPlanner::planMAPF (grid_based_planner.cpp:64-92; PIBT, src/mapf/pibt.cpp) on the grid of `grid_resolution`
 * metres, then the rules that let an agent move its next_waypoint to the next node of its path.  Graph search on a few
 * hundred nodes (35 us per group in the reference's log): it stays on the host, the device path reads its result from
 * the agent records (dlscqp_agent.next_waypoint).  A context needs no GPU. */
typedef struct dlscqp_mapf dlscqp_mapf;
typedef struct dlscqp_mapf_config {
    int32_t world_dimension;        /* param.world_dimension                                        */
    int32_t M;                      /* param.M                                                      */
    int32_t world_use_octomap;      /* param.world_use_octomap: static obstacles block grid nodes    */
    int32_t reserved_;
    double dt;                      /* param.dt                                                     */
    double grid_resolution;         /* param.grid_resolution                                        */
    double world_resolution;        /* param.world_resolution (octomap cell size)                   */
    double world_z_2d;              /* param.world_z_2d                                             */
    double communication_range;     /* param.communication_range (< 0: unlimited)                   */
    double agent_radius;            /* mission.agents[0].radius (the reference uses agent 0's)       */
    double agent_downwash;          /* mission.agents[0].downwash: vertical grid step = resolution * downwash */
    double obs_uncertainty_horizon; /* param.obs_uncertainty_horizon                                */
    double world_min[3], world_max[3];
} dlscqp_mapf_config;

/* boxes: the rows of the world CSV (centre x,y,z, size x,y,z), as for dlscqp_world_create; may be NULL / 0. */
int dlscqp_mapf_create(const dlscqp_mapf_config *cfg, const double *boxes, int32_t n_boxes, dlscqp_mapf **out);
void dlscqp_mapf_destroy(dlscqp_mapf *mapf);

typedef struct dlscqp_mapf_step_args {
    int32_t n_agents;
    int32_t n1;                  /* n + 1 control points per segment (layout of `plans`)                         */
    const float *position;       /* [n][3] AgentManager::getCurrentPosition                                       */
    const float *start_point;    /* [n][3] mission start points                                                   */
    const float *desired_goal;   /* [n][3] desired_goal_point                                                     */
    const float *current_goal;   /* [n][3] current_goal_point (the goal LP's last result)                         */
    float *next_waypoint;        /* [n][3] in/out                                                                 */
    const float *plans;          /* [n][M][n1][3] the plans of the previous step (getTraj()), NULL before the first */
    int32_t n_obs;               /* dynamic obstacles (obstacle_generator.getObstacles())                         */
    int32_t reserved_;
    const float *obs_position;   /* [n_obs][3] */
    const float *obs_velocity;   /* [n_obs][3] */
    const double *obs_radius;    /* [n_obs]    */
    const double *obs_max_acc;   /* [n_obs]    */
    const uint8_t *alert;        /* optional [n][n_obs]: obstacle is in the agent's collision alert (TrajOptResult) */
    float *out_desired;          /* optional [n][3]: path[1] of every agent (before the update rules)             */
    int32_t *out_group;          /* optional [n]: communication group index                                       */
} dlscqp_mapf_step_args;

/* One decentralizedMAPP pass: next_waypoint is updated in place.  Returns DLSCQP_SUCCESS, 1 if the planner failed for
 * some group (its waypoints stay as they are, like the reference's "MAPF failed"), or a negative error code. */
int dlscqp_mapf_step(dlscqp_mapf *mapf, const dlscqp_mapf_step_args *args);

/* ---- the per-step exchange of the solved plans, and one call per replanning step -----------------------
 * MultiSyncSimulator::broadcastMsgs (multi_sync_simulator.cpp:468-514) hands every agent's plan to every other agent; its
 * payload per agent is AgentManager::getAgent (agent_manager.cpp:249-263): here the control points, the sub-goal and the verdict.
 * With the agents sharded over the GPUs of a node (contiguous id ranges, every rank `n_local` agents) that is ONE ncclAllGather
 * over xGMI per step, issued by THIS library from librccl -- loaded with dlopen when the first exchange is created, a copy that
 * is already in the process (torch's) first -- on a communication stream the exchange owns (highest priority: its few workgroups
 * run beside the next step's plan kernels).  Record layout of the gathered table: [world * n_local][M (n+1) 3 + 4] float32 =
 * control points | sub-goal | verdict (the int32's bits): rank r's agents at rows [r n_local, (r+1) n_local).
 * Bootstrap: rank 0 calls dlscqp_exchange_unique_id and hands the 128 bytes to the other ranks by whatever the host has (a file,
 * MPI, torch.distributed's store) -- in a forked child BEFORE anything has touched the GPU; every rank then calls
 * dlscqp_exchange_create (collective: ncclCommInitRank).  world == 1 needs no id (and with local_only != 0 no librccl: the
 * "collective" is a device copy -- hosts without RCCL, tests). */
typedef struct dlscqp_exchange dlscqp_exchange;
#define DLSCQP_EXCHANGE_ID_BYTES 128
typedef struct dlscqp_exchange_config {
    int32_t rank, world;
    int32_t n_local;          /* agents of every rank (pad the last rank's shard)                              */
    int32_t local_only;       /* world == 1 only: no RCCL, the gathered table is a device copy of the record    */
    const void *unique_id;    /* DLSCQP_EXCHANGE_ID_BYTES from rank 0 (may be NULL when world == 1)             */
    float *recv_buffers[2];   /* optional: caller-owned device memory for the two gathered tables, each
                                 world * n_local * (M (n+1) 3 + 4) floats (a host framework that wants to see the table as
                                 one of its own tensors); NULL: the exchange allocates them                       */
} dlscqp_exchange_config;
int dlscqp_exchange_unique_id(void *id_out /* [DLSCQP_EXCHANGE_ID_BYTES] */);
int dlscqp_exchange_create(dlscqp_ctx *ctx, const dlscqp_exchange_config *cfg, dlscqp_exchange **out);
void dlscqp_exchange_destroy(dlscqp_exchange *x);
/* Post the rank's plans of a step: ctrl [n_local][M][n+1][3] float32, goal [n_local][3] float32, status [n_local] int32 (device),
 * complete on `ready_stream` when the call is made.  The library records an event there; on its communication stream it packs the
 * records and starts the all-gather.  Two buffer pairs alternate: exchange k waits for exchange k - 2.  The caller keeps the three
 * arrays untouched until dlscqp_exchange_before_overwrite() has made the stream that overwrites them wait for the packing (call
 * it ahead of the launch after next: it refers to the post before last).  Asynchronous; no host synchronisation. */
int dlscqp_exchange_post(dlscqp_exchange *x, const float *ctrl, const float *goal, const int32_t *status, void *ready_stream);
int dlscqp_exchange_before_overwrite(dlscqp_exchange *x, void *stream);
/* The same for a caller that rotates through `tables` (2..8) output tables, step k writing table k mod tables: `stream` waits for the
 * post that READ the table the next step writes -- post k - tables, a collective that ran steps ago -- instead of for the post before
 * last.  With several swarm groups per GPU this is what keeps them independent: a wait for the collective of step k - 2 starts every
 * group's step k in lockstep behind a transfer that itself waited for all groups (measured on one rank with four groups: 0.471 -> 0.441 ms per step; profiles/r06_force_dist_trace.md). */
int dlscqp_exchange_before_overwrite_rotating(dlscqp_exchange *x, void *stream, int32_t tables);
/* `stream` waits until the communication stream has read the arrays (or records) of the post just made: for a caller that goes on
 * to rewrite them on `stream` at once instead of rotating its tables (dlscqp_exchange_before_overwrite) */
int dlscqp_exchange_wait_packed(dlscqp_exchange *x, void *stream);
/* The same without the packing pass: `records` [n_local][M (n+1) 3 + 4] is what the plan kernel wrote
 * (dlscqp_device_batch.out_record_stride) and is sent as it lies.  The caller keeps it untouched until the all-gather has read
 * it: dlscqp_exchange_before_overwrite (which then waits for the collective of the post before last, not for a packing pass). */
int dlscqp_exchange_post_records(dlscqp_exchange *x, const float *records, void *ready_stream);
/* Make `stream` wait for the exchange posted last; *table_out (optional) = the gathered table (device memory of the exchange,
 * valid until the post after next). */
int dlscqp_exchange_wait(dlscqp_exchange *x, void *stream, const float **table_out);
/* Stage times of the following `capacity` exchanges (events with timing on the two streams; a diagnosis, off by default):
 * pack = the packing kernel, all_gather = the collective, lag = from "plans ready" on the compute side to "table gathered". */
int dlscqp_exchange_set_timing(dlscqp_exchange *x, int32_t capacity);
int dlscqp_exchange_get_times(dlscqp_exchange *x, float *pack_ms, float *all_gather_ms, float *lag_ms, int32_t capacity, int32_t *n_out);

/* One replanning step of this rank in ONE call (MultiSyncSimulator::plan + broadcastMsgs, multi_sync_simulator.cpp:516-524,
 * 468-514): the launches of all swarm groups on their streams, and -- with an exchange -- the events that tie the step's output
 * table to the post.  The groups' batches of a step write slices of one table (ctrl / goal / status: what is posted); a caller
 * that rotates through B >= 3 such tables may use DLSCQP_EXCHANGE_PIPELINED: the exchange of step k runs while step k + 1 is
 * planned (agents whose neighbours are all local need nothing from it), a launch waits only until the table it overwrites has
 * been packed.  DLSCQP_EXCHANGE_SERIAL: the launches of step k + 1 wait for the gathered table of step k.  No host
 * synchronisation, no allocation after the first call. */
enum { DLSCQP_EXCHANGE_NONE = 0, DLSCQP_EXCHANGE_PIPELINED = 1, DLSCQP_EXCHANGE_SERIAL = 2 };
typedef struct dlscqp_step_group {
    dlscqp_ctx *ctx;                    /* the group's context                                                   */
    const dlscqp_device_batch *batch;   /* what it plans this step (with its hint_iters: the launch order)         */
    void *stream;                       /* the group's stream                                                     */
} dlscqp_step_group;
typedef struct dlscqp_step_args {
    int32_t n_groups;
    int32_t exchange_mode;              /* DLSCQP_EXCHANGE_*; ignored without an exchange                         */
    const dlscqp_step_group *groups;
    dlscqp_exchange *exchange;          /* or NULL                                                                */
    const float *ctrl, *goal;           /* the step's output table, [n_local] agents: what the groups' batches write */
    const int32_t *status;
    const float *records;               /* ... or the table as records (out_record_stride): posted without a packing pass;
                                           ctrl / goal / status are then not read                                   */
    /* ---- round 6 (appended, zero = the behaviour of before) */
    int32_t rotation;                   /* DLSCQP_EXCHANGE_PIPELINED: how many output tables the caller rotates through (step k writes
                                           table k mod rotation; 3..8): the launches wait for post k - rotation
                                           (dlscqp_exchange_before_overwrite_rotating).  0 or 2: for the post before last.          */
    int32_t reserved_;
} dlscqp_step_args;
int dlscqp_step_device(const dlscqp_step_args *step);

/* Host helper: convert n_rows AoS rows to the SoA arrays above (host memory, caller-owned). */
void dlscqp_rows_to_soa(const dlscqp_lsc_row *rows, int64_t n_rows, float *nx, float *ny,
                        float *nz, float *px, float *py, float *pz, double *d, uint32_t *key);

/* Kernel timing with HIP events recorded on the launch stream, without synchronising inside the
 * timed region: dlscqp_set_timing(ctx, capacity) (re)arms `capacity` event pairs (0 = off); every
 * following dlscqp_plan_batch_device launch records one pair until they are used up;
 * dlscqp_get_kernel_times waits for the recorded launches and returns their durations in ms. */
int dlscqp_set_timing(dlscqp_ctx *ctx, int32_t capacity);
int dlscqp_get_kernel_times(dlscqp_ctx *ctx, float *ms_out, int32_t capacity, int32_t *n_out);

/* ---- one launch per replanning step (small missions) ----------------------------------------------------------------------
 * The steady-state step of a closed loop -- neighbour lists (dlscqp_neighbours_device), corridor boxes (dlscqp_generate_sfc_device,
 * the shift of a step, not the initialisation), half-spaces (dlscqp_generate_lsc_device), waypoint trap (dlscqp_waypoint_trap_device),
 * goal LP + QP with dlscqp_device_batch.advance_after -- as ONE kernel: a workgroup of 256 threads per agent runs the stages one
 * after the other, one grid-wide barrier separates every agent's reads of the others' previous plans from the epilogue that
 * overwrites them (the reference plans every agent on the messages of the step before, multi_sync_simulator.cpp:468-524).  Same
 * structures as the single entry points (ns, world + sfc, trap: NULL = stage not run), same buffers, same results bit for bit.
 * For missions of the reference's size: at most DLSCQP_FUSED_MAX_AGENTS agents and one workgroup per CU, every stage over all agents
 * (no local_ids), lists short enough for the one-launch neighbour search, a static-shape instantiation with rows beyond its register
 * cache or communication-range rows (every launch file of the reference).  DLSCQP_ERR_UNSUPPORTED otherwise, with nothing launched:
 * the caller then makes the calls one by one. */
#define DLSCQP_FUSED_MAX_AGENTS 256
int dlscqp_fused_step_device(dlscqp_ctx *ctx, const dlscqp_neighbour_search *ns, const dlscqp_world *world,
                             const dlscqp_sfc_gen *sfc, const dlscqp_lsc_gen *gen, const dlscqp_waypoint_trap *trap,
                             const dlscqp_device_batch *batch, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* DLSCQP_H */
