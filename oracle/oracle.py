"""ctypes front-end of the CPU oracle (oracle/kiss_oracle.c).

TEST INFRASTRUCTURE -- pinned against the reference's own sources (oracle/ref.py), PARITY UNPINNED for the
third-party arithmetic only (see kiss_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does.

The class/method names mirror the reference's Python wrappers so the parity tests read like
the reference's own usage (python/kiss_icp/{registration,mapping,voxelization,preprocess,
threshold,kiss_icp}.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkiss_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "kiss_oracle.c")
    hdr = os.path.join(_HERE, "kiss_oracle.h")
    if (
        not force
        and os.path.exists(_LIB_PATH)
        and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libkiss_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("converged", C.c_int32),
        ("n_source", C.c_uint64),
        ("n_corr_last", C.c_uint64),
        ("points_examined", C.c_uint64),
        ("n_corr_total", C.c_uint64),
    ]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class _Config(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_double),
        ("max_range", C.c_double),
        ("min_range", C.c_double),
        ("max_points_per_voxel", C.c_int),
        ("min_motion_th", C.c_double),
        ("initial_threshold", C.c_double),
        ("max_num_iterations", C.c_int),
        ("convergence_criterion", C.c_double),
        ("max_num_threads", C.c_int),
        ("deskew", C.c_int),
    ]


class _Threshold(C.Structure):
    _fields_ = [
        ("min_motion_threshold", C.c_double),
        ("max_range", C.c_double),
        ("model_sse", C.c_double),
        ("num_samples", C.c_int),
    ]


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # KISS_ORACLE_LIB: another build of the same restatement (oracle/Makefile `sanitize`: ASan + UBSan)
    alt = os.environ.get("KISS_ORACLE_LIB")
    if not alt:
        build()
    L = C.CDLL(alt or _LIB_PATH)
    vp, sz, d, i = C.c_void_p, C.c_size_t, C.c_double, C.c_int
    sig = {
        "ko_se3_from_matrix": (i, [_dp, vp]),
        "ko_se3_matrix": (None, [vp, _dp]),
        "ko_se3_mul": (None, [vp, vp, vp]),
        "ko_se3_inverse": (None, [vp, vp]),
        "ko_se3_exp": (None, [_dp, vp]),
        "ko_se3_log": (None, [vp, _dp]),
        "ko_se3_act": (None, [vp, _dp, _dp]),
        "ko_ldlt6_solve": (None, [_dp, _dp, _dp]),
        "ko_point_to_voxel": (None, [_dp, d, C.POINTER(C.c_int32)]),
        "ko_voxel_downsample": (sz, [_dp, sz, d, _dp]),
        "ko_set_downsample_order": (None, [i]),
        "ko_get_downsample_order": (i, []),
        "ko_map_create": (vp, [d, d, C.c_uint]),
        "ko_map_destroy": (None, [vp]),
        "ko_map_clear": (None, [vp]),
        "ko_map_empty": (i, [vp]),
        "ko_map_num_voxels": (sz, [vp]),
        "ko_map_num_points": (sz, [vp]),
        "ko_map_add_points": (None, [vp, _dp, sz]),
        "ko_map_remove_far": (None, [vp, _dp]),
        "ko_map_update_origin": (None, [vp, _dp, sz, _dp]),
        "ko_map_update_pose": (None, [vp, _dp, sz, _dp]),
        "ko_map_pointcloud": (sz, [vp, _dp]),
        "ko_map_closest_neighbor": (d, [vp, _dp, _dp]),
        "ko_map_closest_neighbor_counted": (d, [vp, _dp, _dp, C.POINTER(C.c_uint64)]),
        "ko_align_points_to_map": (i, [_dp, sz, vp, _dp, d, d, i, d, i, _dp, C.POINTER(_Stats)]),
        "ko_build_linear_system": (None, [_dp, sz, vp, d, d, _dp, _dp, C.POINTER(C.c_uint64)]),
        "ko_preprocess": (sz, [_dp, sz, _dp, sz, _dp, d, d, i, i, _dp]),
        "ko_threshold_init": (None, [C.POINTER(_Threshold), d, d, d]),
        "ko_threshold_compute": (d, [C.POINTER(_Threshold)]),
        "ko_threshold_update": (None, [C.POINTER(_Threshold), _dp]),
        "ko_config_default": (None, [C.POINTER(_Config)]),
        "ko_pipeline_create": (vp, [C.POINTER(_Config)]),
        "ko_pipeline_destroy": (None, [vp]),
        "ko_pipeline_register_frame": (i, [vp, _dp, sz, _dp, sz]),
        "ko_pipeline_pose": (None, [vp, _dp]),
        "ko_pipeline_delta": (None, [vp, _dp]),
        "ko_pipeline_map": (vp, [vp]),
        "ko_pipeline_output_size": (sz, [vp, i]),
        "ko_pipeline_output": (None, [vp, i, _dp]),
        "ko_pipeline_last_stats": (None, [vp, C.POINTER(_Stats), _dp]),
        "ko_num_procs": (i, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("points must be (N, 3)")
    return a


def _p(a):
    return a.ctypes.data_as(_dp)


def _mat(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    assert T.shape == (4, 4)
    return T


# --- SE(3) helpers (for the known-answer tests) -------------------------------------------
class _SE3(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]


def se3_exp(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    T = _SE3()
    lib().ko_se3_exp(_p(a), C.byref(T))
    M = np.empty((4, 4))
    lib().ko_se3_matrix(C.byref(T), _p(M))
    return M


def se3_log(M):
    M = _mat(M)
    T = _SE3()
    lib().ko_se3_from_matrix(_p(M), C.byref(T))
    a = np.empty(6)
    lib().ko_se3_log(C.byref(T), _p(a))
    return a


def se3_roundtrip(M):
    """matrix -> SE3d -> matrix; returns (matrix, status) (status -1: SOPHUS_ENSURE failure)"""
    M = _mat(M)
    T = _SE3()
    st = lib().ko_se3_from_matrix(_p(M), C.byref(T))
    out = np.empty((4, 4))
    lib().ko_se3_matrix(C.byref(T), _p(out))
    return out, st


def se3_inverse(M):
    M = _mat(M)
    T, R = _SE3(), _SE3()
    lib().ko_se3_from_matrix(_p(M), C.byref(T))
    lib().ko_se3_inverse(C.byref(T), C.byref(R))
    out = np.empty((4, 4))
    lib().ko_se3_matrix(C.byref(R), _p(out))
    return out


def se3_mul(A, B):
    a, b, r = _SE3(), _SE3(), _SE3()
    lib().ko_se3_from_matrix(_p(_mat(A)), C.byref(a))
    lib().ko_se3_from_matrix(_p(_mat(B)), C.byref(b))
    lib().ko_se3_mul(C.byref(a), C.byref(b), C.byref(r))
    out = np.empty((4, 4))
    lib().ko_se3_matrix(C.byref(r), _p(out))
    return out


def ldlt6_solve(A, b):
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(6)
    lib().ko_ldlt6_solve(_p(A), _p(b), _p(x))
    return x


def point_to_voxel(p, voxel_size):
    p = np.ascontiguousarray(p, dtype=np.float64)
    v = (C.c_int32 * 3)()
    lib().ko_point_to_voxel(_p(p), voxel_size, v)
    return np.array(v[:], dtype=np.int32)


REFERENCE_ORDER, INDEX_ORDER = 1, 0


def set_downsample_order(order):
    """output order of VoxelDownsample, process-wide: REFERENCE_ORDER (default) = bucket order of the tsl::robin_map the
    reference iterates (core/VoxelUtils.cpp:17-19); INDEX_ORDER = ascending original index.  Returns the previous one."""
    old = lib().ko_get_downsample_order()
    lib().ko_set_downsample_order(int(order))
    return old


def voxel_down_sample(frame, voxel_size):
    """kiss_icp/voxelization.py:28-30"""
    frame = _pts(frame)
    out = np.empty_like(frame)
    n = lib().ko_voxel_downsample(_p(frame), len(frame), voxel_size, _p(out))
    return out[:n].copy()


class VoxelHashMap:
    """kiss_icp/mapping.py:37-68 over the oracle map"""

    def __init__(self, voxel_size, max_distance, max_points_per_voxel, _borrow=None):
        self._owned = _borrow is None
        self._h = (
            lib().ko_map_create(voxel_size, max_distance, max_points_per_voxel)
            if _borrow is None
            else _borrow
        )

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib().ko_map_destroy(self._h)
            self._h = None

    def clear(self):
        lib().ko_map_clear(self._h)

    def empty(self):
        return bool(lib().ko_map_empty(self._h))

    def num_voxels(self):
        return lib().ko_map_num_voxels(self._h)

    def update(self, points, pose=None):
        points = _pts(points)
        pose = np.eye(4) if pose is None else pose
        pose = np.asarray(pose, dtype=np.float64)
        if pose.shape == (3,):
            lib().ko_map_update_origin(self._h, _p(points), len(points), _p(np.ascontiguousarray(pose)))
        else:
            lib().ko_map_update_pose(self._h, _p(points), len(points), _p(_mat(pose)))

    def add_points(self, points):
        points = _pts(points)
        lib().ko_map_add_points(self._h, _p(points), len(points))

    def remove_far_away_points(self, origin):
        origin = np.ascontiguousarray(origin, dtype=np.float64)
        lib().ko_map_remove_far(self._h, _p(origin))

    def point_cloud(self):
        n = lib().ko_map_num_points(self._h)
        out = np.empty((n, 3))
        lib().ko_map_pointcloud(self._h, _p(out))
        return out

    def closest_neighbor(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        nn = np.empty(3)
        d = lib().ko_map_closest_neighbor(self._h, _p(q), _p(nn))
        return nn, d


class Registration:
    """kiss_icp/registration.py:38-65"""

    def __init__(self, max_num_iterations, convergence_criterion, max_num_threads=0):
        self.max_num_iterations = max_num_iterations
        self.convergence_criterion = convergence_criterion
        self.max_num_threads = max_num_threads
        self.last_stats = None

    def align_points_to_map(self, points, voxel_map, initial_guess, max_correspondance_distance, kernel):
        points = _pts(points)
        T = np.empty((4, 4))
        st = _Stats()
        lib().ko_align_points_to_map(
            _p(points), len(points), voxel_map._h, _p(_mat(initial_guess)),
            max_correspondance_distance, kernel, self.max_num_iterations,
            self.convergence_criterion, self.max_num_threads, _p(T), C.byref(st),
        )
        self.last_stats = st.asdict()
        return T


def build_linear_system(source, voxel_map, max_correspondance_distance, kernel):
    source = _pts(source)
    JTJ = np.empty((6, 6))
    JTr = np.empty(6)
    nc = C.c_uint64(0)
    lib().ko_build_linear_system(
        _p(source), len(source), voxel_map._h, max_correspondance_distance, kernel, _p(JTJ), _p(JTr), C.byref(nc)
    )
    return JTJ, JTr, nc.value


class Preprocessor:
    """kiss_icp/preprocess.py:38-51"""

    def __init__(self, max_range, min_range, deskew, max_num_threads=0):
        self.max_range, self.min_range, self.deskew, self.max_num_threads = max_range, min_range, deskew, max_num_threads

    def preprocess(self, frame, timestamps, relative_motion):
        frame = _pts(frame)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64).ravel()
        out = np.empty_like(frame)
        n = lib().ko_preprocess(
            _p(frame), len(frame), _p(ts) if len(ts) else None, len(ts), _p(_mat(relative_motion)),
            self.max_range, self.min_range, int(self.deskew), self.max_num_threads, _p(out),
        )
        if n == C.c_size_t(-1).value:
            raise IndexError("timestamps shorter than frame (std::vector::at)")
        return out[:n].copy()


class AdaptiveThreshold:
    """kiss_icp/threshold.py:40-58"""

    def __init__(self, initial_threshold, min_motion_th, max_range):
        self._t = _Threshold()
        lib().ko_threshold_init(C.byref(self._t), initial_threshold, min_motion_th, max_range)

    def get_threshold(self):
        return lib().ko_threshold_compute(C.byref(self._t))

    def update_model_deviation(self, model_deviation):
        lib().ko_threshold_update(C.byref(self._t), _p(_mat(model_deviation)))


def default_config(**kw):
    c = _Config()
    lib().ko_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class KissICP:
    """pipeline::KissICP (cpp/kiss_icp/pipeline/KissICP.{hpp,cpp}) over the oracle"""

    def __init__(self, **config):
        self._cfg = default_config(**config)
        self._h = lib().ko_pipeline_create(C.byref(self._cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ko_pipeline_destroy(self._h)
            self._h = None

    def register_frame(self, frame, timestamps=()):
        frame = _pts(frame)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64).ravel()
        rc = lib().ko_pipeline_register_frame(self._h, _p(frame), len(frame), _p(ts) if len(ts) else None, len(ts))
        if rc != 0:
            raise IndexError("timestamps shorter than frame")
        return self.output(0), self.output(1)

    def register_frame_noout(self, frame, timestamps=()):
        """same, without copying the outputs back (for timing)"""
        ts = timestamps
        rc = lib().ko_pipeline_register_frame(self._h, _p(frame), len(frame), _p(ts) if len(ts) else None, len(ts))
        assert rc == 0

    def output(self, which):
        n = lib().ko_pipeline_output_size(self._h, which)
        out = np.empty((n, 3))
        lib().ko_pipeline_output(self._h, which, _p(out))
        return out

    @property
    def last_pose(self):
        T = np.empty((4, 4))
        lib().ko_pipeline_pose(self._h, _p(T))
        return T

    @property
    def last_delta(self):
        T = np.empty((4, 4))
        lib().ko_pipeline_delta(self._h, _p(T))
        return T

    @property
    def local_map(self):
        return VoxelHashMap(0, 0, 0, _borrow=lib().ko_pipeline_map(self._h))

    def last_stats(self):
        st = _Stats()
        sg = C.c_double(0)
        lib().ko_pipeline_last_stats(self._h, C.byref(st), C.byref(sg))
        d = st.asdict()
        d["sigma"] = sg.value
        return d


def num_procs():
    return lib().ko_num_procs()
