// Registration.hpp -- mirrors cpp/kiss_icp/core/Registration.hpp:33-45 of PRBonn/kiss-icp v1.2.3.
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <vector>

#include "Linalg.hpp"
#include "VoxelHashMap.hpp"

struct kicp_registration;

namespace kiss_icp {

struct Registration {
    explicit Registration(int max_num_iteration, double convergence_criterion, int max_num_threads);
    Registration(int max_num_iteration, double convergence_criterion, int max_num_threads, int device_id);
    ~Registration();
    // The reference's Registration is an implicitly copyable struct of three parameters (core/Registration.hpp:33-45).  A
    // copy here is a second registration with the same parameters on the same device (its own stream and scratch buffers;
    // the statistics of the last call travel with it); moving hands the handle over.
    Registration(const Registration &other);
    Registration &operator=(const Registration &other);
    Registration(Registration &&other) noexcept;
    Registration &operator=(Registration &&other) noexcept;

    Sophus::SE3d AlignPointsToMap(const std::vector<Eigen::Vector3d> &frame,
                                  const VoxelHashMap &voxel_map,
                                  const Sophus::SE3d &initial_guess,
                                  const double max_correspondence_distance,
                                  const double kernel_scale);

    Sophus::SE3d AlignPointsToMap(PointSpan frame,
                                  const VoxelHashMap &voxel_map,
                                  const Sophus::SE3d &initial_guess,
                                  const double max_correspondence_distance,
                                  const double kernel_scale);

    int max_num_iterations_;
    double convergence_criterion_;
    int max_num_threads_;  // kept for signature compatibility; the device schedules itself

    // statistics of the last call (iterations executed, correspondences, map points examined)
    int last_iterations_ = 0;
    bool last_converged_ = false;
    unsigned long long last_points_examined_ = 0;

    kicp_registration *handle_ = nullptr;
    int device_id_ = 0;
};
}  // namespace kiss_icp
