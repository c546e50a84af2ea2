"""Preprocessor -- python/kiss_icp/preprocess.py:38-51 over kicp_preprocess."""
import ctypes as C

import numpy as np

from . import _cabi


def get_preprocessor(config, device_id=0):
    return Preprocessor(
        max_range=config.data.max_range,
        min_range=config.data.min_range,
        deskew=config.data.deskew,
        max_num_threads=config.registration.max_num_threads,
        device_id=device_id,
    )


class Preprocessor:
    def __init__(self, max_range, min_range, deskew, max_num_threads, device_id=0):
        self.max_range, self.min_range, self.deskew = max_range, min_range, deskew
        self.max_num_threads, self.device_id = max_num_threads, device_id

    def preprocess(self, frame: np.ndarray, timestamps: np.ndarray, relative_motion: np.ndarray):
        pts = _cabi.points(frame)
        ts = np.ascontiguousarray(np.asarray(timestamps, dtype=np.float64).ravel())
        T = _cabi.mat4(relative_motion)
        out = np.empty_like(pts)
        n = C.c_size_t(0)
        st = _cabi.lib().kicp_preprocess(
            _cabi.ptr(pts), len(pts), _cabi.ptr(ts) if len(ts) else None, len(ts), _cabi.dptr(T),
            self.max_range, self.min_range, int(bool(self.deskew)), self.device_id, _cabi.ptr(out), C.byref(n))
        if st == 8:  # KICP_ERR_TIMESTAMPS: std::vector::at would throw out_of_range -> IndexError in pybind
            raise IndexError(_cabi.lib().kicp_last_error().decode())
        _cabi.check(st)
        return out[: n.value]
