// sfc_world.hpp -- host-side construction of the occupancy grid and nearest-occupied-cell table that sfc_core.hpp reads.
#pragma once
#include <cstdint>
#include <vector>

#include "sfc_core.hpp"

namespace dlscqp {

struct SfcWorldHost {
    double res = 0.1;
    float wmin[3] = {0, 0, 0}, wmax[3] = {0, 0, 0};
    int lo[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    std::vector<unsigned char> occ;
    std::vector<int> near_d2;
    std::vector<signed char> near_off;
    std::vector<unsigned char> near_n;
    std::vector<uint32_t> near_packed;  // the three arrays above as one 32-byte record per cell (SfcWorldView::near)
    std::vector<uint32_t> gaps;         // sfc_corner_gaps of every record (SfcWorldView::gaps)
    size_t cells() const { return (size_t)n[0] * n[1] * n[2]; }
};

// boxes: [n][6] centre x,y,z and size x,y,z of axis-aligned obstacles (the reference's world CSV rows,
// MapManager::updateOctreeFromCSV, map_manager.cpp:264-316).
void build_sfc_world(const double *boxes, int n_boxes, double resolution, const float wmin[3], const float wmax[3],
                     SfcWorldHost &w);

// SfcWorldView::svt[gap_max] of a built world: (n_0 + 2)(n_1 + 2)(n_2 + 2) entries.
void build_sfc_svt(const SfcWorldHost &w, int gap_max, std::vector<int32_t> &svt);

}  // namespace dlscqp
