// Threshold.hpp -- mirrors cpp/kiss_icp/core/Threshold.hpp:30-50 and Threshold.cpp:30-49 of
// PRBonn/kiss-icp v1.2.3.  O(1) host arithmetic per frame (the fused device pipeline keeps its own
// copy of this state in HBM; this class serves code that composes the stages by hand).
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <cmath>

#include "Linalg.hpp"

namespace kiss_icp {

struct AdaptiveThreshold {
    explicit AdaptiveThreshold(double initial_threshold, double min_motion_threshold, double max_range)
        : min_motion_threshold_(min_motion_threshold),
          max_range_(max_range),
          model_sse_(initial_threshold * initial_threshold),
          num_samples_(1) {}

    /// Update the current belief of the deviation from the prediction model
    void UpdateModelDeviation(const Sophus::SE3d &current_deviation);

    /// Returns the KISS-ICP adaptive threshold used in registration
    inline double ComputeThreshold() const { return std::sqrt(model_sse_ / num_samples_); }

    // configurable parameters
    double min_motion_threshold_;
    double max_range_;

    // Local cache for computation
    double model_sse_;
    int num_samples_;
};

}  // namespace kiss_icp
