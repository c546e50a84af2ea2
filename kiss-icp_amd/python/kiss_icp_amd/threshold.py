"""Adaptive / fixed threshold -- python/kiss_icp/threshold.py:30-58.  An O(1) scalar recurrence per
frame (core/Threshold.cpp:30-49, Threshold.hpp:38): host arithmetic here; inside the fused device
pipeline (KissICP in kiss_icp.py) the same recurrence runs in the ICP kernel's epilogue."""
import numpy as np


def get_threshold_estimator(config):
    if config.adaptive_threshold.fixed_threshold is not None:
        return FixedThreshold(config.adaptive_threshold.fixed_threshold)
    return AdaptiveThreshold(config)


class FixedThreshold:
    def __init__(self, fixed_threshold: float):
        self.fixed_threshold = fixed_threshold

    def get_threshold(self):
        return self.fixed_threshold

    def update_model_deviation(self, model_deviation):
        pass


class AdaptiveThreshold:
    def __init__(self, config=None, *, initial_threshold=None, min_motion_th=None, max_range=None):
        if config is not None:
            initial_threshold = config.adaptive_threshold.initial_threshold
            min_motion_th = config.adaptive_threshold.min_motion_th
            max_range = config.data.max_range
        self.min_motion_threshold = float(min_motion_th)
        self.max_range = float(max_range)
        self.model_sse = float(initial_threshold) ** 2  # Threshold.cpp:35
        self.num_samples = 1

    def get_threshold(self):
        return float(np.sqrt(self.model_sse / self.num_samples))  # Threshold.hpp:38

    def update_model_deviation(self, model_deviation: np.ndarray):
        T = np.asarray(model_deviation, dtype=np.float64)
        # Eigen::AngleAxisd(R).angle() == 2*atan2(|q.vec|, |q.w|) of the rotation's quaternion
        R = T[:3, :3]
        cos_t = min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0))
        sin_t = 0.5 * np.sqrt((R[2, 1] - R[1, 2]) ** 2 + (R[0, 2] - R[2, 0]) ** 2 + (R[1, 0] - R[0, 1]) ** 2)
        theta = float(np.arctan2(sin_t, cos_t))
        delta_rot = 2.0 * self.max_range * np.sin(theta / 2.0)
        delta_trans = float(np.linalg.norm(T[:3, 3]))
        model_error = delta_trans + delta_rot
        if model_error > self.min_motion_threshold:
            self.model_sse += model_error * model_error
            self.num_samples += 1
