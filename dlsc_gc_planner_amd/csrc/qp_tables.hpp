// qp_tables.hpp -- host-side precomputation shared by every agent (see qp_tables.cpp).
#pragma once
#include <string>
#include <vector>

#include "../../include/dlscqp.h"
#include "qp_types.hpp"

namespace dlscqp {

struct HostTables {
    KParams kp;                  // table pointers left null; the owner fills them in
    std::vector<double> tabN;    // [M][P][f]     whitening maps (the kernel reads their Gram tables, and these where a row looks dependent)
    std::vector<double> tabG;    // [M][P][P|1]   G_T = N_T N_T'
    std::vector<double> tabX0s;  // [M][P][3]
    std::vector<double> tabX0g;  // [M][P]
    std::vector<double> tabNN;   // [M][P]
    std::vector<double> tabSIW;  // [M][n_slots]
    std::vector<double> tabKIW;  // [M][3][P]
    std::vector<double> tabPhi;  // [M][10]
    std::vector<double> hr_lmax, hr_lmin;  // [M] its largest and smallest eigenvalue
    std::vector<double> hr_cond; // [M]           condition number of the reduced Hessian of T terminal segments (0 where it is not positive definite)
    std::vector<Slot> slots;
};

// dim/M/n/phi/dt are passed separately: TrajOptimizer caches them at construction and
// updateParam does not refresh them (traj_optimizer.cpp:7-11, 167-169).
bool build_tables(const dlscqp_config &cfg, int dim, int M, int n, int phi, double dt, HostTables &out,
                  std::string &err);

}  // namespace dlscqp
