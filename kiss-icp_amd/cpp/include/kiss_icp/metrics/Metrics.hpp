// Metrics.hpp -- kiss_icp::metrics of the reference's public C++ API
// (cpp/kiss_icp/metrics/Metrics.hpp:33-37 of PRBonn/kiss-icp v1.2.3): the KITTI dev-kit segment
// errors and the absolute trajectory error.  Offline evaluation, host arithmetic only -- kept so the
// reference's pybind names `_kitti_seq_error` / `_absolute_trajectory_error`
// (python/kiss_icp/pybind/kiss_icp_pybind.cpp:141-143) and its Python `metrics.py` keep working on
// this build.  Nothing here is on the registration path.
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <tuple>
#include <vector>

#include "kiss_icp/core/Linalg.hpp"

namespace kiss_icp::metrics {

// (average translation error [%], average rotation error [deg/m]) over all 100..800 m segments
// starting every 10th frame (Metrics.cpp:91-156)
std::tuple<float, float> SeqError(const std::vector<Eigen::Matrix4d> &poses_gt,
                                  const std::vector<Eigen::Matrix4d> &poses_result);

// (rotation RMSE [rad], translation RMSE [m]) after a rigid (no scale) Umeyama alignment of the
// estimated positions to the ground truth (Metrics.cpp:158-189)
std::tuple<float, float> AbsoluteTrajectoryError(const std::vector<Eigen::Matrix4d> &poses_gt,
                                                 const std::vector<Eigen::Matrix4d> &poses_result);

}  // namespace kiss_icp::metrics
