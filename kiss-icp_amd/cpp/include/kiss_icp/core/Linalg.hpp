// Linalg.hpp -- the vector / rigid-transform types of the reference's public C++ API.
//
// The reference's headers expose Eigen::Vector3d, Eigen::Vector3i, Eigen::Matrix4d and
// Sophus::SE3d (cpp/kiss_icp/core/*.hpp, pipeline/KissICP.hpp).  When Eigen 3.4 and Sophus are
// installed they are used as they are and this header only includes them.  When they are not (as in
// the build image of this repository: the reference fetches them from the network at configure time)
// a minimal stand-in with the same names, the same memory layout (3 contiguous doubles; column-major
// 4x4) and the handful of members the KISS-ICP API needs is provided, so that code written against
// the reference's headers compiles unchanged.  None of this runs on the hot path: the arithmetic
// that matters happens in the HIP kernels behind include/kicp.h.
#pragma once

#if __has_include(<Eigen/Core>) && __has_include(<sophus/se3.hpp>) && !defined(KISS_ICP_HIP_FORCE_COMPAT)
#include <Eigen/Core>
#include <sophus/se3.hpp>
#define KISS_ICP_HIP_HAVE_EIGEN 1
#else
#define KISS_ICP_HIP_HAVE_EIGEN 0

#include <array>
#include <cmath>
#include <cstddef>
#include <stdexcept>

namespace Eigen {

template <typename T>
struct Vec3 {
    T v[3];
    Vec3() : v{T(0), T(0), T(0)} {}
    Vec3(T x, T y, T z) : v{x, y, z} {}
    static Vec3 Zero() { return Vec3(); }
    T &operator[](std::size_t i) { return v[i]; }
    const T &operator[](std::size_t i) const { return v[i]; }
    T &operator()(std::size_t i) { return v[i]; }
    const T &operator()(std::size_t i) const { return v[i]; }
    T &x() { return v[0]; }
    T &y() { return v[1]; }
    T &z() { return v[2]; }
    const T &x() const { return v[0]; }
    const T &y() const { return v[1]; }
    const T &z() const { return v[2]; }
    T *data() { return v; }
    const T *data() const { return v; }
    Vec3 operator+(const Vec3 &o) const { return Vec3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3 operator-(const Vec3 &o) const { return Vec3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3 operator*(T s) const { return Vec3(v[0] * s, v[1] * s, v[2] * s); }
    bool operator==(const Vec3 &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
    T squaredNorm() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
    double norm() const { return std::sqrt(static_cast<double>(squaredNorm())); }
};
using Vector3d = Vec3<double>;
using Vector3i = Vec3<int>;
static_assert(sizeof(Vector3d) == 24, "std::vector<Eigen::Vector3d>::data() must be N x 3 doubles");

struct Matrix3d {
    double m[9];  // column-major like Eigen
    Matrix3d() : m{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
    static Matrix3d Identity() { return Matrix3d(); }
    double &operator()(int r, int c) { return m[c * 3 + r]; }
    const double &operator()(int r, int c) const { return m[c * 3 + r]; }
    Vector3d operator*(const Vector3d &p) const {
        return Vector3d((*this)(0, 0) * p[0] + (*this)(0, 1) * p[1] + (*this)(0, 2) * p[2],
                        (*this)(1, 0) * p[0] + (*this)(1, 1) * p[1] + (*this)(1, 2) * p[2],
                        (*this)(2, 0) * p[0] + (*this)(2, 1) * p[1] + (*this)(2, 2) * p[2]);
    }
};

struct Matrix4d {
    double m[16];  // column-major like Eigen
    Matrix4d() : m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1} {}
    static Matrix4d Identity() { return Matrix4d(); }
    double &operator()(int r, int c) { return m[c * 4 + r]; }
    const double &operator()(int r, int c) const { return m[c * 4 + r]; }
    double *data() { return m; }
    const double *data() const { return m; }
};

}  // namespace Eigen

namespace Sophus {

// unit quaternion (x, y, z, w) + translation: Sophus::SE3d's own storage
class SE3d {
public:
    SE3d() : q_{0, 0, 0, 1}, t_() {}
    SE3d(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) : t_(t) { set_rotation(R, true); }
    explicit SE3d(const Eigen::Matrix4d &T) : t_(T(0, 3), T(1, 3), T(2, 3)) {
        Eigen::Matrix3d R;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R(r, c) = T(r, c);
        set_rotation(R, true);
    }
    static SE3d from_quaternion(const double q[4], const double t[3]) {
        SE3d s;
        for (int i = 0; i < 4; ++i) s.q_[i] = q[i];
        s.t_ = Eigen::Vector3d(t[0], t[1], t[2]);
        return s;
    }

    const Eigen::Vector3d &translation() const { return t_; }
    Eigen::Vector3d &translation() { return t_; }
    const std::array<double, 4> &unit_quaternion_coeffs() const { return q_; }

    Eigen::Matrix3d rotationMatrix() const {
        const double x = q_[0], y = q_[1], z = q_[2], w = q_[3];
        Eigen::Matrix3d R;
        R(0, 0) = 1 - 2 * (y * y + z * z);
        R(0, 1) = 2 * (x * y - w * z);
        R(0, 2) = 2 * (x * z + w * y);
        R(1, 0) = 2 * (x * y + w * z);
        R(1, 1) = 1 - 2 * (x * x + z * z);
        R(1, 2) = 2 * (y * z - w * x);
        R(2, 0) = 2 * (x * z - w * y);
        R(2, 1) = 2 * (y * z + w * x);
        R(2, 2) = 1 - 2 * (x * x + y * y);
        return R;
    }
    Eigen::Matrix4d matrix() const {
        const Eigen::Matrix3d R = rotationMatrix();
        Eigen::Matrix4d T;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T(r, c) = R(r, c);
            T(r, 3) = t_[r];
        }
        return T;
    }
    SE3d inverse() const {
        SE3d r;
        r.q_ = {-q_[0], -q_[1], -q_[2], q_[3]};
        const Eigen::Vector3d nt(-t_[0], -t_[1], -t_[2]);
        r.t_ = r.rotationMatrix() * nt;
        return r;
    }
    SE3d operator*(const SE3d &o) const {
        SE3d r;
        const auto &a = q_;
        const auto &b = o.q_;
        r.q_ = {a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
                a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0],
                a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]};
        const double n = std::sqrt(r.q_[0] * r.q_[0] + r.q_[1] * r.q_[1] + r.q_[2] * r.q_[2] + r.q_[3] * r.q_[3]);
        for (auto &c : r.q_) c /= n;
        r.t_ = t_ + rotationMatrix() * o.t_;
        return r;
    }
    Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return rotationMatrix() * p + t_; }

private:
    void set_rotation(const Eigen::Matrix3d &R, bool check) {
        if (check) {  // SOPHUS_ENSURE(isOrthogonal(R)) and det(R) > 0
            double err = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += R(i, k) * R(j, k);
                    s -= (i == j) ? 1.0 : 0.0;
                    err += s * s;
                }
            const double det = R(0, 0) * (R(1, 1) * R(2, 2) - R(1, 2) * R(2, 1)) -
                               R(0, 1) * (R(1, 0) * R(2, 2) - R(1, 2) * R(2, 0)) +
                               R(0, 2) * (R(1, 0) * R(2, 1) - R(1, 1) * R(2, 0));
            if (!(std::sqrt(err) < 1e-10) || !(det > 0)) throw std::invalid_argument("SE3d: R is not a rotation");
        }
        // Shepperd's method (Eigen's quaternion-from-matrix)
        double t = R(0, 0) + R(1, 1) + R(2, 2);
        if (t > 0) {
            t = std::sqrt(t + 1.0);
            q_[3] = 0.5 * t;
            t = 0.5 / t;
            q_[0] = (R(2, 1) - R(1, 2)) * t;
            q_[1] = (R(0, 2) - R(2, 0)) * t;
            q_[2] = (R(1, 0) - R(0, 1)) * t;
        } else {
            int i = 0;
            if (R(1, 1) > R(0, 0)) i = 1;
            if (R(2, 2) > R(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
            q_[i] = 0.5 * t;
            t = 0.5 / t;
            q_[3] = (R(k, j) - R(j, k)) * t;
            q_[j] = (R(j, i) + R(i, j)) * t;
            q_[k] = (R(k, i) + R(i, k)) * t;
        }
    }
    std::array<double, 4> q_;
    Eigen::Vector3d t_;
};

}  // namespace Sophus
#endif  // compat

namespace kiss_icp::detail {
// row-major 4x4 (the C-ABI's layout) <-> Sophus::SE3d
inline void se3_to_rowmajor(const Sophus::SE3d &T, double out[16]) {
    const auto M = T.matrix();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = M(r, c);
}
inline Sophus::SE3d se3_from_rowmajor(const double in[16]) {
    Eigen::Matrix4d M;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) M(r, c) = in[r * 4 + c];
    return Sophus::SE3d(M);
}
}  // namespace kiss_icp::detail
