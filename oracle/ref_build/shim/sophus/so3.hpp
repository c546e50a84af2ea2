// Stand-in for <sophus/so3.hpp>: SO3d::hat only (core/Registration.cpp:86).  TEST INFRASTRUCTURE.
#pragma once
#include <Eigen/Core>
namespace Sophus {
struct SO3d {
    static Eigen::Matrix3d hat(const Eigen::Vector3d &w) {
        Eigen::Matrix3d O;
        O(0, 1) = -w.z();
        O(0, 2) = w.y();
        O(1, 0) = w.z();
        O(1, 2) = -w.x();
        O(2, 0) = -w.y();
        O(2, 1) = w.x();
        return O;
    }
};
}  // namespace Sophus
