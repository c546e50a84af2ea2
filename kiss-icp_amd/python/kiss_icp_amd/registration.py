"""Registration -- same surface as the reference's python/kiss_icp/registration.py:38-65, backed by
kicp_align_points_to_map (the persistent HIP ICP kernel)."""
import ctypes as C

import numpy as np

from . import _cabi
from .mapping import VoxelHashMap


def get_registration(config, device_id=0):
    return Registration(
        max_num_iterations=config.registration.max_num_iterations,
        convergence_criterion=config.registration.convergence_criterion,
        max_num_threads=config.registration.max_num_threads,
        device_id=device_id,
    )


class Registration:
    def __init__(self, max_num_iterations: int, convergence_criterion: float, max_num_threads: int = 0,
                 device_id: int = 0):
        h = C.c_void_p()
        _cabi.check(_cabi.lib().kicp_registration_create(max_num_iterations, convergence_criterion, max_num_threads, device_id, C.byref(h)))
        self._registration = h
        self.last_stats = None

    def __del__(self):
        if getattr(self, "_registration", None):
            _cabi.lib().kicp_registration_destroy(self._registration)
            self._registration = None

    def align_points_to_map(self, points: np.ndarray, voxel_map: VoxelHashMap, initial_guess: np.ndarray,
                            max_correspondance_distance: float, kernel: float) -> np.ndarray:
        pts = _cabi.points(points)
        T0 = _cabi.mat4(initial_guess)
        T = np.empty((4, 4))
        st = _cabi.IcpStats()
        _cabi.check(_cabi.lib().kicp_align_points_to_map(
            self._registration, _cabi.ptr(pts), len(pts), voxel_map._internal_map, _cabi.dptr(T0),
            max_correspondance_distance, kernel, _cabi.dptr(T), C.byref(st)))
        self.last_stats = st.asdict()
        return T

    def last_system(self):
        """(JTJ (6,6), JTr (6,), n_corr) accumulated by the last iteration of the most recent
        align_points_to_map (BuildLinearSystem, Registration.cpp:80-121)"""
        JTJ, JTr, n = np.empty((6, 6)), np.empty(6), C.c_uint64(0)
        _cabi.check(_cabi.lib().kicp_registration_last_system(self._registration, _cabi.dptr(JTJ), _cabi.dptr(JTr), C.byref(n)))
        return JTJ, JTr, n.value
