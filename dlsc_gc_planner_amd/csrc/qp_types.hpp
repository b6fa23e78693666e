// qp_types.hpp -- plain data shared by the host table builder and the solver kernel.
//
// Layout of the per-agent problem after the host-side elimination:
//   control points of axis k:  x_k = X0s_T [pos_k vel_k acc_k]' + X0g_T goal_k + N_T w_k
//   where N_T = Z L_T^{-T} "whitens" the reduced Hessian (objective = 1/2 |w - 0|^2 + const in
//   w-space) for the agent's number of terminal segments T (traj_optimizer.cpp:543-551).
#pragma once
#include <stdint.h>

namespace dlscqp {

enum SlotKind : uint8_t { SLOT_BOUND = 0, SLOT_VEL = 1, SLOT_ACC = 2, SLOT_COMM = 3, SLOT_WAYPOINT = 4 };

// One two-sided structured row  lo[bsel] <= sum_t coef[t] * x[idx[t]] <= hi[bsel]  whose pattern is
// identical for every agent: variable bounds / SFC box faces (traj_optimizer.cpp:237-270,
// 385-410), dynamic limits (:453-487) and communication range (:490-513), scaled to metres.
struct Slot {
    int16_t idx[3];  // x-space indices k*P + m*(n+1) + i, -1 = unused
    int8_t coef[3];
    uint8_t kind;
    uint16_t bsel;   // index into the per-agent bound table
    float scale;     // factor back to the reference's row scaling (n/dt, n(n-1)/dt^2, 1)
};
static_assert(sizeof(Slot) == 16, "Slot layout");

struct KParams {
    // dimensions
    int dim, M, n1, P, f, phi;
    int n_slots;   // two-sided structured rows
    int n_struct;  // the first n_struct of them are bounds, velocity and acceleration rows (static-shape kernels do not read those from the table)
    int nb;        // entries of the per-agent bound table
    int has_comm, use_sfc, stop_mode, rsfc_mode;
    int max_iter;
    int phi_ok;    // tabPhi is valid
    int dbg_exit_phase;  // profiling builds only (gi_core.hpp: DQ_PHASE_EXIT), else -1
    int wd_cut;    // the weak-duality infeasibility test of gi_core.hpp is valid (its objective bound assumes phi_n = 1)
    int dep_calibrated;  // weights and dt are the launch files' (control_input_weight 0.01, terminal_weight 1, dt 0.2, phi_n 1): the values
                         // DEP_SUSPECT was calibrated on; elsewhere every shape uses the wide threshold DEP_SUSPECT_SLACK (qp_tables.cpp)
    // weights and constants (param.hpp)
    double dt;
    double w_c, w_t, w_s;
    double comm_range;
    double z_2d;
    double world_min[3], world_max[3];
    double vel_scale, acc_scale;  // dt/n, dt^2/(n(n-1))
    double dep_suspect;           // DEP_SUSPECT where dep_calibrated, else DEP_SUSPECT_SLACK
    double dep_pivot;             // DEP_PIVOT x eps x the Hessian's condition number (qp_tables.cpp: hr_cond and the slack block): times
                                  // 1 / the smallest pivot ratio so far = the suspicion threshold a numerically singular active set asks for
    double Qb[36];                // control_input_weight * Q_base (traj_optimizer.cpp:172-187)
    double sigma[32];             // slack whitening 1/sqrt(2 w_s (M-m)/M), m < M <= 32
    int t_valid[32];              // reduced Hessian positive definite for T = idx+1
    // device (or host, in the emulation build) tables
    const double *tabN;           // [M][P][f]     whitening maps N_T: read only where a row looks dependent on the active rows (plan_agent: whitened_length)
    const double *tabG;           // [M][P][P|1]   Gram table N_T N_T' of the whitening map (row stride odd, as it is copied to LDS)
    const double *tabX0s;         // [M][P][3]
    const double *tabX0g;         // [M][P]
    const double *tabNN;          // [M][P]        |N_T[p,:]|^2  (pivot scaling of LSC rows)
    const double *tabSIW;         // [M][n_slots]  1 / |N_T' a_slot|  (pivot scaling of structured rows)
    const double *tabPhi;         // [M][10]       objective of the unconstrained minimiser as a quadratic form of (pos, vel, acc, goal) per axis
    const double *tabKIW;         // [M][3][P]     the same by kind and first control point: bound / velocity / acceleration row at p
    const Slot *slots;            // [n_slots]
};

// Tolerances.  VIOL_TOL: a row violated by less than this many metres counts as satisfied.
// CPLEX_FEAS_TOL: CPLEX 22.1.1 default feasibility tolerance in the reference's own row scaling;
// a row that cannot be satisfied but is violated by no more than this is accepted, as CPLEX would.
constexpr double VIOL_TOL = 1e-9;
constexpr double CPLEX_FEAS_TOL = 1e-6;
#ifndef DQ_DEP_SUSPECT
#define DQ_DEP_SUSPECT 1e-5
#endif
// Measured with tests/fuzz_cpu.py's batches (trace build): a . dx / |N'a|^2 of rows that DO depend on the active rows stays below
// 8e-9 in the static shapes without slack variables and below 7e-7 in the others; rows that do not depend start at 1e-5.  Three orders
// of margin each: 1e-5 where a . dx comes from registers (1e-4 there costs the S64 step 9 %: rows between 1e-5 and 1e-4 are
// common in its crowded swarms), 1e-3 in the shapes with slack variables and the runtime shapes.
// a . dx <= DEP_SUSPECT |N'a|^2: the row may depend on the active rows -- |z|^2 is measured again in the whitened space (plan_agent:
// whitened_length)
// Round 4 (tests/test_dependency_margins.py, seeds 1-12 of sixteen configurations): the first measure of a dependent row depends
// on how well conditioned the ACTIVE SET is, not only on the shape: at control_input_weight 1 / terminal_weight 20 the 2-D M = 10
// shape produced 2.2e-6 -- a factor of five under 1e-5 -- where the launch files' values give 3e-11.  The narrow threshold is
// therefore used only with the parameter values it was calibrated on (KParams::dep_calibrated); any other configuration takes the
// wide one (450 x over the largest value seen, at the price of measuring a few per cent more rows twice).
// Round 5 (tests/tools/margins_long.py, 200 fresh seeds x 13 configurations on the column-record iteration): ONE batch (seed 70162,
// 3-D M = 5 with dynamic obstacles) held a dependent row whose first measure was 8.3e-4 -- 1.2 x under the wide threshold of then,
// 1e-3 -- in an infeasible problem a few iterations before its verdict: 29 rows active, the objective at 5e13, the noise scale of the
// whitened measure 2e14 |N'a|^2 (the whitened measure itself still called the row dependent with four orders to spare).  The
// wide threshold is 1e-2 since: 12 x over that value; it costs a re-measurement for 0.01-0.03 % of the independent rows of the
// launch files' shapes more than before (16 % of the iterations in the all-slack RSFC shape, where a row's slack term dominates).
constexpr double DEP_SUSPECT = DQ_DEP_SUSPECT, DEP_SUSPECT_SLACK = 1e-2;
// Round 5, second sweep (seeds 80000-80199): the constants above assume that the active rows' records are sound.  One agent (seed 80048,
// control_input_weight 1e-4, dt 0.1, M = 10, slack variables; infeasible) had accepted a row with |z|^2 = 1.3e-12 |N'a|^2 -- independent
// by every test, and in exact arithmetic -- and from then on a.dx of rows that depend on the active ones came out at 0.12 |N'a|^2:
// above ANY constant.  Over all configurations a dependent row's first measure stays below 0.13 eps kappa / rho_min (kappa: condition
// number of the Hessian, rho_min: smallest pivot ratio so far; tests/tools/dep_threshold_study.py) unless the problem's numbers have
// blown up.  So: once DEP_PIVOT eps kappa / rho_min exceeds the wide constant it is the threshold, and from 1 on every row is
// measured in the whitened space.  With the launch files' kappa of 2.4e5 that starts at a pivot below 5e-7 |N'a|^2: the calibrated
// behaviour of the shapes the sweeps cover is untouched.
constexpr double DEP_PIVOT = 100.0;
// |z|^2 <= DEP_NOISE sum_j (sum_p |N_pj| |u_p|)^2: the whitened length of u = a - A' r is rounding noise.  r carries a relative error of
// ~1e-11 (condition of the active rows' Gram matrix), u inherits it, and the quotient of a dependent row was measured between 1e-31
// and 1e-20 over the random batches of tests/fuzz_cpu.py; rows that do not depend on the active ones start at 1e-12.
constexpr double DEP_NOISE = 1e-17;
// noise^2 <= DEP_XSPACE |N'a|^2, noise^2 = sum_j (sum_p |N_pj| |u_p|)^2 with u = a - A' r: nothing is left of the row's normal
// even in absolute values -- u is zero, to eight digits, on every variable of the whitened problem: the same half-space twice, or
// its opposite (what an infeasible pair is made of).  For such a row |z|^2 / |N'a|^2 is the squared relative error of r: 1e-23 ...
// 6e-20 measured, i.e. on both sides of DEP_TOL (round 4: fresh seeds of tests/test_dependency_margins.py found 4.5e-21 and
// 5.8e-20), and the noise test does not apply (|z|^2 is of the size of noise^2 there).  noise^2 / |N'a|^2 is <= 2.3e-19 for every
// such row and >= 3e-7 for every row that does not depend on the active ones (sixteen configurations, fifty seeds); two float32
// normals one ulp apart give >= 4e-15: they stay independent, as the oracle has them.
constexpr double DEP_XSPACE = 1e-16;
constexpr double DEP_TOL = 1e-20;  // |z|^2 <= DEP_TOL |N'a|^2  -> the row's normal depends on the active rows'
constexpr double SP_EPSILON = 1e-9;        // sp_const.hpp:3
constexpr double SP_EPSILON_FLOAT = 1e-5;  // sp_const.hpp:4

struct RowsView {  // structure-of-arrays LSC rows of the whole batch (include/dlscqp.h)
    const float *nx, *ny, *nz, *px, *py, *pz;
    const double *d;
    const uint32_t *key;
};

}  // namespace dlscqp
