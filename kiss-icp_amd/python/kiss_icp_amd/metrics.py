"""Trajectory metrics -- python/kiss_icp/metrics.py:29-38 of the reference: thin wrappers over the pybind
module's _kitti_seq_error / _absolute_trajectory_error (host arithmetic, kiss-icp_amd/cpp/src/metrics.cpp)."""
import os
import sys
from typing import Tuple

import numpy as np


def _pybind():
    try:
        import kiss_icp_pybind
    except ImportError:
        cpp = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cpp"))
        if cpp not in sys.path:
            sys.path.insert(0, cpp)
        import kiss_icp_pybind  # raises when the module has not been built (make -C kiss-icp_amd/cpp)
    return kiss_icp_pybind


def sequence_error(gt_poses: np.ndarray, results_poses: np.ndarray) -> Tuple[float, float]:
    """Sptis the sequence error for a given trajectory in camera coordinate frames."""
    return _pybind()._kitti_seq_error(np.asarray(gt_poses, dtype=np.float64), np.asarray(results_poses, dtype=np.float64))


def absolute_trajectory_error(gt_poses: np.ndarray, results_poses: np.ndarray) -> Tuple[float, float]:
    """Computes the Absolute Trajectory Error (ATE) between the ground truth and estimated poses."""
    return _pybind()._absolute_trajectory_error(np.asarray(gt_poses, dtype=np.float64), np.asarray(results_poses, dtype=np.float64))
