// TEST DOUBLE of <octomap/octomap_types.h> -- test infrastructure, not the product and not the library.
//
// matp.hpp can reuse the real octomap::point3d (define MATP_HAVE_OCTOMAP before including it: INTEGRATION.md section 2), which is
// what a maintainer of the reference does.  octomap is not installed in this image, so that branch had never been compiled.  This
// header declares octomath::Vector3 with the signatures of octomap 1.9's math/Vector3.h that the mirror relies on (float storage,
// x()/y()/z(), operator(), + - * with float arithmetic, dot / norm / norm_sq / distance returning double) and the typedef
// octomap::point3d: enough to compile and run tests/cpp/shim_test.cpp through the MATP_HAVE_OCTOMAP branch.
#pragma once
#include <cmath>

namespace octomath {
class Vector3 {
public:
    Vector3() { data[0] = data[1] = data[2] = 0.0f; }
    Vector3(const Vector3 &other) { data[0] = other.data[0]; data[1] = other.data[1]; data[2] = other.data[2]; }
    Vector3(float x, float y, float z) { data[0] = x; data[1] = y; data[2] = z; }
    inline Vector3 &operator=(const Vector3 &other) { data[0] = other.data[0]; data[1] = other.data[1]; data[2] = other.data[2]; return *this; }
    inline double dot(const Vector3 &other) const { return x() * other.x() + y() * other.y() + z() * other.z(); }
    inline const float &operator()(unsigned int i) const { return data[i]; }
    inline float &operator()(unsigned int i) { return data[i]; }
    inline float &x() { return operator()(0); }
    inline float &y() { return operator()(1); }
    inline float &z() { return operator()(2); }
    inline const float &x() const { return operator()(0); }
    inline const float &y() const { return operator()(1); }
    inline const float &z() const { return operator()(2); }
    inline Vector3 operator-() const { return Vector3(-data[0], -data[1], -data[2]); }
    inline Vector3 operator+(const Vector3 &other) const { return Vector3(data[0] + other.data[0], data[1] + other.data[1], data[2] + other.data[2]); }
    inline Vector3 operator*(float x) const { return Vector3(data[0] * x, data[1] * x, data[2] * x); }
    inline Vector3 operator-(const Vector3 &other) const { return Vector3(data[0] - other.data[0], data[1] - other.data[1], data[2] - other.data[2]); }
    inline double norm() const { return std::sqrt(norm_sq()); }
    inline double norm_sq() const { return x() * x() + y() * y() + z() * z(); }
    inline double distance(const Vector3 &other) const {
        const double dist_x = x() - other.x(), dist_y = y() - other.y(), dist_z = z() - other.z();
        return std::sqrt(dist_x * dist_x + dist_y * dist_y + dist_z * dist_z);
    }

protected:
    float data[3];
};
}  // namespace octomath

namespace octomap {
typedef octomath::Vector3 point3d;
}
