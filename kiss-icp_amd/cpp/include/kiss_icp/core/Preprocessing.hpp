// Preprocessing.hpp -- mirrors cpp/kiss_icp/core/Preprocessing.hpp:32-46 of PRBonn/kiss-icp v1.2.3.
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <vector>

#include "Linalg.hpp"
#include "VoxelUtils.hpp"

namespace kiss_icp {

struct Preprocessor {
    Preprocessor(const double max_range,
                 const double min_range,
                 const bool deskew,
                 const int max_num_threads);

    /// throws std::out_of_range when 0 < timestamps.size() < frame.size() and deskewing is on
    /// (the reference's std::vector::at, core/Preprocessing.cpp:76-77)
    std::vector<Eigen::Vector3d> Preprocess(const std::vector<Eigen::Vector3d> &frame,
                                            const std::vector<double> &timestamps,
                                            const Sophus::SE3d &relative_motion) const;
    std::vector<Eigen::Vector3d> Preprocess(PointSpan frame,
                                            const double *timestamps,
                                            std::size_t n_timestamps,
                                            const Sophus::SE3d &relative_motion) const;
    double max_range_;
    double min_range_;
    bool deskew_;
    int max_num_threads_;
    int device_id_ = -1;  // -1: kiss_icp::DefaultDevice()
};
}  // namespace kiss_icp
