"""voxel_down_sample -- python/kiss_icp/voxelization.py:28-30 over kicp_voxel_downsample."""
import ctypes as C

import numpy as np

from . import _cabi


def voxel_down_sample(points: np.ndarray, voxel_size: float, device_id: int = 0):
    pts = _cabi.points(points)
    out = np.empty_like(pts)
    n = C.c_size_t(0)
    _cabi.check(_cabi.lib().kicp_voxel_downsample(_cabi.ptr(pts), len(pts), voxel_size, device_id, _cabi.ptr(out), C.byref(n)))
    return out[: n.value]
