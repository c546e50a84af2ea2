"""VoxelHashMap -- same surface as the reference's python/kiss_icp/mapping.py:37-68, backed by the
device-resident map of libkicp (kicp_map_*, include/kicp.h)."""
import ctypes as C

import numpy as np

from . import _cabi


def get_voxel_hash_map(config, device_id=0):
    return VoxelHashMap(
        voxel_size=config.mapping.voxel_size,
        max_distance=config.data.max_range,
        max_points_per_voxel=config.mapping.max_points_per_voxel,
        device_id=device_id,
    )


class VoxelHashMap:
    def __init__(self, voxel_size: float, max_distance: float, max_points_per_voxel: int, device_id: int = 0,
                 _borrowed=None, _owner=None):
        self._owned = _borrowed is None
        self._owner = _owner  # a borrowed handle (KissICP.local_map) keeps the object that owns it alive
        if _borrowed is None:
            h = C.c_void_p()
            _cabi.check(_cabi.lib().kicp_map_create(voxel_size, max_distance, max_points_per_voxel, device_id, C.byref(h)))
            self._internal_map = h
        else:
            self._internal_map = _borrowed

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_internal_map", None):
            _cabi.lib().kicp_map_destroy(self._internal_map)
            self._internal_map = None

    def copy(self):
        """an independent map with the same content (the reference's VoxelHashMap is a copyable value type)"""
        h = C.c_void_p()
        _cabi.check(_cabi.lib().kicp_map_clone(self._internal_map, C.byref(h)))
        m = VoxelHashMap.__new__(VoxelHashMap)
        m._owned, m._owner, m._internal_map = True, None, h
        return m

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()

    def clear(self):
        _cabi.check(_cabi.lib().kicp_map_clear(self._internal_map))

    def empty(self):
        e = C.c_int(0)
        _cabi.check(_cabi.lib().kicp_map_empty(self._internal_map, C.byref(e)))
        return bool(e.value)

    def update(self, points: np.ndarray, pose: np.ndarray = np.eye(4)):
        """Add points to the map and drop the voxels far from the pose's origin.
        `pose` is a 4x4 matrix or (like the first pybind overload) a 3-vector origin."""
        pts = _cabi.points(points)
        pose = np.asarray(pose, dtype=np.float64)
        if pose.shape == (3,):
            o = np.ascontiguousarray(pose)
            _cabi.check(_cabi.lib().kicp_map_update_origin(self._internal_map, _cabi.ptr(pts), len(pts), _cabi.dptr(o)))
        else:
            T = _cabi.mat4(pose)
            _cabi.check(_cabi.lib().kicp_map_update_pose(self._internal_map, _cabi.ptr(pts), len(pts), _cabi.dptr(T)))

    def add_points(self, points):
        pts = _cabi.points(points)
        _cabi.check(_cabi.lib().kicp_map_add_points(self._internal_map, _cabi.ptr(pts), len(pts)))

    def remove_far_away_points(self, origin):
        o = np.ascontiguousarray(origin, dtype=np.float64).reshape(3)
        _cabi.check(_cabi.lib().kicp_map_remove_far(self._internal_map, _cabi.dptr(o)))

    def point_cloud(self) -> np.ndarray:
        nv, npts = C.c_size_t(0), C.c_size_t(0)
        _cabi.check(_cabi.lib().kicp_map_size(self._internal_map, C.byref(nv), C.byref(npts)))
        out = np.empty((npts.value, 3))
        n = C.c_size_t(0)
        _cabi.check(_cabi.lib().kicp_map_pointcloud(self._internal_map, _cabi.ptr(out), npts.value, C.byref(n)))
        return out[: n.value]

    # -- beyond the reference's Python wrapper (C++ VoxelHashMap has it: VoxelHashMap.hpp:51) --
    def closest_neighbor(self, queries):
        """batched GetClosestNeighbor: (nn (N,3), dist (N,))"""
        q = _cabi.points(np.atleast_2d(queries))
        nn = np.empty_like(q)
        dist = np.empty(len(q))
        _cabi.check(_cabi.lib().kicp_map_closest_neighbor(self._internal_map, _cabi.ptr(q), len(q), _cabi.ptr(nn), _cabi.ptr(dist)))
        return nn, dist

    def num_voxels(self):
        nv = C.c_size_t(0)
        _cabi.check(_cabi.lib().kicp_map_size(self._internal_map, C.byref(nv), None))
        return nv.value
