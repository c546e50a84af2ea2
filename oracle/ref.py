"""ctypes face of oracle/_ref/libkiss_ref.so: the REFERENCE's own sources (cpp/kiss_icp/{core,pipeline}/*.cpp)
compiled unmodified from /root/reference against stand-in third-party headers (oracle/ref_build/).

TEST INFRASTRUCTURE.  Used to pin the oracle (tests/test_ref_pins_oracle.py) and to generate the golden
fixtures tests/golden/ref_*.npz (tests/golden/make_ref_golden.py).  The library can only be built where
/root/reference exists; once built it travels with the repository snapshot.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libkiss_ref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH) or os.path.isdir("/root/reference/cpp/kiss_icp")


def build():
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_build")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, sz, d, i = C.c_void_p, C.c_size_t, C.c_double, C.c_int
        L.kr_voxel_downsample.restype = sz
        L.kr_voxel_downsample.argtypes = [vp, sz, d, vp]
        L.kr_preprocess.restype = C.c_long
        L.kr_preprocess.argtypes = [vp, sz, vp, sz, vp, d, d, i, vp]
        L.kr_map_create.restype = vp
        L.kr_map_create.argtypes = [d, d, C.c_uint]
        for name in ("kr_map_destroy", "kr_map_clear", "kr_pipeline_destroy"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp]
        L.kr_map_empty.restype = i
        L.kr_map_empty.argtypes = [vp]
        for name in ("kr_map_num_voxels", "kr_map_num_points"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [vp]
        L.kr_map_add_points.restype = None
        L.kr_map_add_points.argtypes = [vp, vp, sz]
        L.kr_map_remove_far.restype = None
        L.kr_map_remove_far.argtypes = [vp, vp]
        L.kr_map_update_origin.restype = None
        L.kr_map_update_origin.argtypes = [vp, vp, sz, vp]
        L.kr_map_update_pose.restype = None
        L.kr_map_update_pose.argtypes = [vp, vp, sz, vp]
        L.kr_map_pointcloud.restype = sz
        L.kr_map_pointcloud.argtypes = [vp, vp]
        L.kr_map_closest_neighbor.restype = d
        L.kr_map_closest_neighbor.argtypes = [vp, vp, vp]
        L.kr_align_points_to_map.restype = None
        L.kr_align_points_to_map.argtypes = [vp, sz, vp, vp, d, d, i, d, vp]
        L.kr_threshold_step.restype = None
        L.kr_threshold_step.argtypes = [C.POINTER(d), C.POINTER(i), d, d, vp, C.POINTER(d)]
        L.kr_pipeline_create.restype = vp
        L.kr_pipeline_create.argtypes = [d, d, d, i, d, d, i, d, i]
        L.kr_pipeline_register_frame.restype = None
        L.kr_pipeline_register_frame.argtypes = [vp, vp, sz, vp, sz]
        for name in ("kr_pipeline_pose", "kr_pipeline_delta"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp, vp]
        L.kr_pipeline_map.restype = vp
        L.kr_pipeline_map.argtypes = [vp]
        L.kr_pipeline_output_size.restype = sz
        L.kr_pipeline_output_size.argtypes = [vp, i]
        L.kr_pipeline_output.restype = sz
        L.kr_pipeline_output.argtypes = [vp, i, vp]
        L.kr_build_mode.restype = C.c_char_p
        L.kr_build_mode.argtypes = []
        for name in ("kr_se3_exp", "kr_se3_log"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp, vp]
        L.kr_se3_mul.restype = None
        L.kr_se3_mul.argtypes = [vp, vp, vp]
        L.kr_ldlt6_solve.restype = None
        L.kr_ldlt6_solve.argtypes = [vp, vp, vp]
        _lib = L
    return _lib


def build_mode():
    """"shim": Eigen / Sophus / tsl / TBB are oracle/ref_build/shim (their arithmetic is the oracle's restatement);
    "thirdparty": the library was built against the real headers (make -C oracle/ref_build THIRDPARTY=...)"""
    return lib().kr_build_mode().decode()


def se3_exp(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    M = np.empty((4, 4))
    lib().kr_se3_exp(a.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p))
    return M


def se3_log(M):
    M = np.ascontiguousarray(M, dtype=np.float64)
    a = np.empty(6)
    lib().kr_se3_log(M.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
    return a


def se3_mul(A, B):
    A, B = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(B, dtype=np.float64)
    M = np.empty((4, 4))
    lib().kr_se3_mul(A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p))
    return M


def ldlt6_solve(A, b):
    A, b = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(6)
    lib().kr_ldlt6_solve(A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
    return x


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _mat(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    assert T.shape == (4, 4)
    return T


def voxel_down_sample(frame, voxel_size):
    frame = _pts(frame)
    out = np.empty_like(frame)
    n = lib().kr_voxel_downsample(_p(frame), len(frame), voxel_size, _p(out))
    return out[:n].copy()


def preprocess(frame, timestamps, relative_motion, max_range, min_range, deskew):
    frame = _pts(frame)
    ts = np.ascontiguousarray(timestamps, dtype=np.float64).ravel()
    out = np.empty_like(frame)
    n = lib().kr_preprocess(_p(frame), len(frame), _p(ts), len(ts), _p(_mat(relative_motion)), max_range, min_range, int(deskew), _p(out))
    if n < 0:
        raise IndexError("timestamps shorter than frame")
    return out[:n].copy()


class VoxelHashMap:
    def __init__(self, voxel_size, max_distance, max_points_per_voxel, _borrow=None, _owner=None):
        self._owned = _borrow is None
        self._owner = _owner
        self._h = lib().kr_map_create(voxel_size, max_distance, max_points_per_voxel) if _borrow is None else _borrow

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib().kr_map_destroy(self._h)
            self._h = None

    def clear(self):
        lib().kr_map_clear(self._h)

    def empty(self):
        return bool(lib().kr_map_empty(self._h))

    def num_voxels(self):
        return lib().kr_map_num_voxels(self._h)

    def add_points(self, points):
        p = _pts(points)
        lib().kr_map_add_points(self._h, _p(p), len(p))

    def remove_far_away_points(self, origin):
        o = np.ascontiguousarray(origin, dtype=np.float64)
        lib().kr_map_remove_far(self._h, _p(o))

    def update(self, points, pose):
        p = _pts(points)
        pose = np.asarray(pose, dtype=np.float64)
        if pose.shape == (3,):
            o = np.ascontiguousarray(pose)
            lib().kr_map_update_origin(self._h, _p(p), len(p), _p(o))
        else:
            lib().kr_map_update_pose(self._h, _p(p), len(p), _p(_mat(pose)))

    def point_cloud(self):
        out = np.empty((lib().kr_map_num_points(self._h), 3))
        n = lib().kr_map_pointcloud(self._h, _p(out))
        return out[:n]

    def closest_neighbor(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        nn = np.empty(3)
        dist = lib().kr_map_closest_neighbor(self._h, _p(q), _p(nn))
        return nn, dist


def align_points_to_map(points, voxel_map, initial_guess, max_correspondance_distance, kernel, max_num_iterations=500,
                        convergence_criterion=1e-4):
    p = _pts(points)
    T = np.empty((4, 4))
    lib().kr_align_points_to_map(_p(p), len(p), voxel_map._h, _p(_mat(initial_guess)), max_correspondance_distance, kernel,
                                 max_num_iterations, convergence_criterion, _p(T))
    return T


def threshold_step(model_sse, num_samples, min_motion_th, max_range, model_deviation):
    sse, ns, sig = C.c_double(model_sse), C.c_int(num_samples), C.c_double(0)
    lib().kr_threshold_step(C.byref(sse), C.byref(ns), min_motion_th, max_range, _p(_mat(model_deviation)), C.byref(sig))
    return sse.value, ns.value, sig.value


class KissICP:
    def __init__(self, voxel_size=1.0, max_range=100.0, min_range=0.0, max_points_per_voxel=20, min_motion_th=0.1,
                 initial_threshold=2.0, max_num_iterations=500, convergence_criterion=1e-4, deskew=1):
        self._h = lib().kr_pipeline_create(voxel_size, max_range, min_range, max_points_per_voxel, min_motion_th, initial_threshold,
                                           max_num_iterations, convergence_criterion, int(deskew))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kr_pipeline_destroy(self._h)
            self._h = None

    def register_frame(self, frame, timestamps=()):
        f = _pts(frame)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64).ravel()
        lib().kr_pipeline_register_frame(self._h, _p(f), len(f), _p(ts), len(ts))
        return self.output(0), self.output(1)

    def output(self, which):
        out = np.empty((lib().kr_pipeline_output_size(self._h, which), 3))
        lib().kr_pipeline_output(self._h, which, _p(out))
        return out

    @property
    def last_pose(self):
        T = np.empty((4, 4))
        lib().kr_pipeline_pose(self._h, _p(T))
        return T

    @property
    def last_delta(self):
        T = np.empty((4, 4))
        lib().kr_pipeline_delta(self._h, _p(T))
        return T

    @property
    def local_map(self):
        return VoxelHashMap(0, 0, 0, _borrow=lib().kr_pipeline_map(self._h), _owner=self)
