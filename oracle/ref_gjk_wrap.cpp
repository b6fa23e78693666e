// ref_gjk_wrap.cpp -- C entry point around the REFERENCE's own openGJK (test infrastructure).
// Compiled together with /root/reference/src/openGJK/openGJK.cpp, where it lies, into
// oracle/_ref/libref_gjk.so by oracle/Makefile (only when /root/reference is present).  The call mirrors
// closestPointsBetweenPointAndConvexHull (/root/reference/include/geometry.hpp:276-306): body 1 = the hull,
// body 2 = the single point (0,0,0).
#include <array>
#include <vector>

#include <openGJK/openGJK.hpp>

extern "C" double ref_gjk_point_hull(const double *pts, int n, double v[3]) {
    struct bd bd1, bd2;
    struct simplex s;
    bd1.numpoints = n;
    for (int i = 0; i < n; i++) bd1.coord.push_back({{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}});
    bd2.numpoints = 1;
    bd2.coord.push_back({{0.0, 0.0, 0.0}});
    s.nvrtx = 0;
    return gjk(bd1, bd2, &s, v);
}
