// Stand-in for <sophus/se3.hpp> (Sophus 1.24.6 is not installed): the SE3d surface the reference's core/ and
// pipeline/ sources use, with the group arithmetic forwarded to the oracle's restatement (ko_se3_*).
// TEST INFRASTRUCTURE (see Eigen/Core in this directory).
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>

#include "so3.hpp"

extern "C" {
void ko_se3_identity(ko_se3 *T);
void ko_se3_matrix(const ko_se3 *T, double M[16]);
void ko_se3_mul(const ko_se3 *A, const ko_se3 *B, ko_se3 *out);
void ko_se3_inverse(const ko_se3 *A, ko_se3 *out);
void ko_se3_exp(const double a[6], ko_se3 *out);
void ko_se3_log(const ko_se3 *A, double a[6]);
void ko_se3_act(const ko_se3 *T, const double p[3], double out[3]);
}

namespace Sophus {
struct SE3d {
    using Tangent = Eigen::Matrix<double, 6, 1>;
    ko_se3 T;
    SE3d() { ko_se3_identity(&T); }
    explicit SE3d(const ko_se3 &t) : T(t) {}
    SE3d operator*(const SE3d &o) const {
        SE3d r;
        ko_se3_mul(&T, &o.T, &r.T);
        return r;
    }
    Eigen::Vector3d operator*(const Eigen::Vector3d &p) const {
        Eigen::Vector3d o;
        ko_se3_act(&T, p.data(), o.data());
        return o;
    }
    SE3d inverse() const {
        SE3d r;
        ko_se3_inverse(&T, &r.T);
        return r;
    }
    Tangent log() const {
        Tangent a;
        ko_se3_log(&T, a.data());
        return a;
    }
    static SE3d exp(const Tangent &a) {
        SE3d r;
        ko_se3_exp(a.data(), &r.T);
        return r;
    }
    Eigen::Vector3d translation() const { return Eigen::Vector3d(T.t[0], T.t[1], T.t[2]); }
    Eigen::Matrix3d rotationMatrix() const {
        double M[16];
        ko_se3_matrix(&T, M);
        Eigen::Matrix3d R;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R(r, c) = M[4 * r + c];
        return R;
    }
};
}  // namespace Sophus
