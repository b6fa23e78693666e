// sfc_world.cpp -- see sfc_world.hpp.  Host code, runs once per world.
#include "sfc_world.hpp"

#include <cmath>

namespace dlscqp {

void build_sfc_world(const double *boxes, int n_boxes, double resolution, const float wmin[3], const float wmax[3],
                     SfcWorldHost &w) {
    w.res = resolution;
    const int R = (int)std::ceil(1.0 / resolution);  // DynamicEDTOctomap maxdist = 1.0 m (map_manager.cpp:14)
    for (int k = 0; k < 3; k++) {
        w.wmin[k] = wmin[k];
        w.wmax[k] = wmax[k];
        w.lo[k] = (int)std::floor(wmin[k] / resolution) - R - 2;
        w.n[k] = (int)std::ceil(wmax[k] / resolution) + R + 2 - w.lo[k];
    }
    const size_t nc = w.cells();
    w.occ.assign(nc, 0);
    auto at = [&](int i, int j, int k) -> long {
        i -= w.lo[0]; j -= w.lo[1]; k -= w.lo[2];
        if (i < 0 || j < 0 || k < 0 || i >= w.n[0] || j >= w.n[1] || k >= w.n[2]) return -1;
        return ((long)i * w.n[1] + j) * w.n[2] + k;
    };
    for (int b = 0; b < n_boxes; b++) {
        const double *c = boxes + 6 * b, *s = c + 3;
        int a0[3], a1[3];
        for (int k = 0; k < 3; k++) {  // cells [round((c - s/2) / res), round((c + s/2) / res))
            a0[k] = (int)std::round((c[k] - 0.5 * s[k]) / resolution);
            a1[k] = (int)std::round((c[k] + 0.5 * s[k]) / resolution);
        }
        for (int i = a0[0]; i < a1[0]; i++)
            for (int j = a0[1]; j < a1[1]; j++)
                for (int k = a0[2]; k < a1[2]; k++) {
                    const long id = at(i, j, k);
                    if (id >= 0) w.occ[id] = 1;
                }
    }
    // nearest occupied cells of every cell within R cells: relaxation from the occupied cells outwards would be the
    // classic distance transform; the worlds are small (~1e5 cells), so each occupied cell simply offers itself to
    // its (2R+1)^3 neighbourhood.
    w.near_d2.assign(nc, -1);
    w.near_n.assign(nc, 0);
    w.near_off.assign(nc * 24, 0);
    for (int i = 0; i < w.n[0]; i++)
        for (int j = 0; j < w.n[1]; j++)
            for (int k = 0; k < w.n[2]; k++) {
                if (!w.occ[((size_t)i * w.n[1] + j) * w.n[2] + k]) continue;
                for (int di = -R; di <= R; di++)
                    for (int dj = -R; dj <= R; dj++)
                        for (int dk = -R; dk <= R; dk++) {
                            const int qi = i - di, qj = j - dj, qk = k - dk;  // the query cell sees (i,j,k) at offset (di,dj,dk)
                            if (qi < 0 || qj < 0 || qk < 0 || qi >= w.n[0] || qj >= w.n[1] || qk >= w.n[2]) continue;
                            const size_t q = ((size_t)qi * w.n[1] + qj) * w.n[2] + qk;
                            const int d2 = di * di + dj * dj + dk * dk;
                            if (w.near_d2[q] >= 0 && d2 > w.near_d2[q]) continue;
                            if (w.near_d2[q] < 0 || d2 < w.near_d2[q]) { w.near_d2[q] = d2; w.near_n[q] = 0; }
                            if (w.near_n[q] < 8) {
                                signed char *o = &w.near_off[q * 24 + 3 * w.near_n[q]];
                                o[0] = (signed char)di; o[1] = (signed char)dj; o[2] = (signed char)dk;
                                w.near_n[q]++;
                            }
                        }
            }
    w.near_packed.assign(nc * 8, 0u);
    for (size_t q = 0; q < nc; q++) {
        uint32_t *rec = &w.near_packed[q * 8];
        rec[0] = (uint32_t)w.near_d2[q];
        rec[1] = (uint32_t)w.near_n[q];
        for (int i = 0; i < 24; i++) rec[2 + (i >> 2)] |= (uint32_t)(uint8_t)w.near_off[q * 24 + i] << (8 * (i & 3));
    }
    w.gaps.assign(nc, 0u);
    for (size_t q = 0; q < nc; q++) w.gaps[q] = sfc_corner_gaps(&w.near_packed[q * 8], resolution);
}

void build_sfc_svt(const SfcWorldHost &w, int gap_max, std::vector<int32_t> &svt) {
    const int n0 = w.n[0], n1 = w.n[1], n2 = w.n[2];
    const size_t s1 = (size_t)n2 + 2, s0 = ((size_t)n1 + 2) * s1;
    svt.assign(((size_t)n0 + 2) * s0, 0);
    // blocked[i][j][k] of lattice point (i, j, k), 0 <= i <= n0 ...: some cell c = I - v (v in {0,1}^3) of the grid reads its
    // corner v as at most gap_max cells from an occupied cell of its record; stored at [i + 1][j + 1][k + 1], then summed in place
    for (int i = 0; i <= n0; i++)
        for (int j = 0; j <= n1; j++)
            for (int k = 0; k <= n2; k++) {
                int blocked = 0;
                for (int v = 0; v < 8; v++) {
                    const int ci = i - (v & 1), cj = j - ((v >> 1) & 1), ck = k - ((v >> 2) & 1);
                    if (ci < 0 || cj < 0 || ck < 0 || ci >= n0 || cj >= n1 || ck >= n2) continue;
                    const uint32_t word = w.gaps[((size_t)ci * n1 + cj) * n2 + ck];
                    if ((int)((word >> (4 * v)) & 15u) <= gap_max) blocked = 1;
                }
                svt[(size_t)(i + 1) * s0 + (size_t)(j + 1) * s1 + (k + 1)] = blocked;
            }
    for (int i = 1; i <= n0 + 1; i++)  // prefix sums along k, j, i in turn
        for (int j = 1; j <= n1 + 1; j++)
            for (int k = 2; k <= n2 + 1; k++) svt[i * s0 + j * s1 + k] += svt[i * s0 + j * s1 + k - 1];
    for (int i = 1; i <= n0 + 1; i++)
        for (int j = 2; j <= n1 + 1; j++)
            for (int k = 1; k <= n2 + 1; k++) svt[i * s0 + j * s1 + k] += svt[i * s0 + (j - 1) * s1 + k];
    for (int i = 2; i <= n0 + 1; i++)
        for (int j = 1; j <= n1 + 1; j++)
            for (int k = 1; k <= n2 + 1; k++) svt[i * s0 + j * s1 + k] += svt[(i - 1) * s0 + j * s1 + k];
}

}  // namespace dlscqp
