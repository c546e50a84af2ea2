"""ctypes binding of libkicp.so (include/kicp.h) -- the only way Python reaches the device.

There is no CPU fallback: if the library is missing it must be built (``__graft_entry__.build()``
or ``make -C kiss-icp_amd/csrc``); if no gfx950 GPU is present every ``*_create`` call raises
``KicpError(KICP_ERR_NO_DEVICE)``.
"""
import ctypes as C
import os

import numpy as np

_CSRC = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "csrc"))
LIB_PATH = os.environ.get("KICP_LIB") or os.path.join(_CSRC, "libkicp.so")  # KICP_LIB: A/B builds of the same library

KICP_OK = 0
STATUS_NAMES = {
    0: "KICP_OK", 1: "KICP_ERR_INVALID_ARG", 2: "KICP_ERR_HIP", 3: "KICP_ERR_OOM", 4: "KICP_ERR_CAPACITY",
    5: "KICP_ERR_RANGE", 6: "KICP_ERR_TIMEOUT", 7: "KICP_ERR_NO_DEVICE", 8: "KICP_ERR_TIMESTAMPS",
}


class KicpError(RuntimeError):
    def __init__(self, status, message):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class IcpStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("converged", C.c_int32), ("n_source", C.c_uint64),
        ("n_corr_last", C.c_uint64), ("points_examined", C.c_uint64), ("n_corr_total", C.c_uint64),
        ("kernel_ms", C.c_double),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Config(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_double), ("max_range", C.c_double), ("min_range", C.c_double),
        ("max_points_per_voxel", C.c_int), ("min_motion_th", C.c_double), ("initial_threshold", C.c_double),
        ("max_num_iterations", C.c_int), ("convergence_criterion", C.c_double), ("max_num_threads", C.c_int),
        ("deskew", C.c_int),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("n_raw", C.c_uint64), ("n_preprocessed", C.c_uint64), ("n_frame_downsample", C.c_uint64),
        ("n_source", C.c_uint64), ("map_voxels", C.c_uint64), ("sigma", C.c_double), ("icp", IcpStats),
    ]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "icp"}
        d["icp"] = self.icp.asdict()
        return d


class HostStats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64), ("backpressure_waits", C.c_uint64), ("capacity_waits", C.c_uint64), ("staging_waits", C.c_uint64),
        ("ring_syncs", C.c_uint64), ("counter_refreshes", C.c_uint64), ("map_grows", C.c_uint64), ("map_rehashes", C.c_uint64),
        ("buffer_grows", C.c_uint64), ("stage_ms", C.c_double), ("enqueue_ms", C.c_double), ("backpressure_ms", C.c_double),
        ("wait_ms", C.c_double), ("device_gap_ms", C.c_double), ("max_device_gap_ms", C.c_double), ("max_call_ms", C.c_double),
        ("device_numa_node", C.c_int32), ("staging_numa_node", C.c_int32), ("staging_helpers", C.c_int32), ("helpers_bound", C.c_int32),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_dp = C.POINTER(C.c_double)
_vp, _sz, _d, _i, _u64p, _szp = C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)

# name -> argtypes (every entry point returns int status unless noted).  tests/test_cabi_symbols.py
# checks this table against include/kicp.h.
SIGNATURES = {
    "kicp_version": [C.POINTER(_i), C.POINTER(_i)],
    "kicp_device_count": [C.POINTER(_i)],
    "kicp_device_name": [_i, C.c_char_p, _sz],
    "kicp_set_option": [C.c_char_p, C.c_long],
    "kicp_map_create": [_d, _d, C.c_uint, _i, C.POINTER(_vp)],
    "kicp_map_destroy": [_vp],
    "kicp_map_clone": [_vp, C.POINTER(_vp)],
    "kicp_map_clear": [_vp],
    "kicp_map_empty": [_vp, C.POINTER(_i)],
    "kicp_map_size": [_vp, _szp, _szp],
    "kicp_map_add_points": [_vp, _vp, _sz],
    "kicp_map_remove_far": [_vp, _dp],
    "kicp_map_update_origin": [_vp, _vp, _sz, _dp],
    "kicp_map_update_pose": [_vp, _vp, _sz, _dp],
    "kicp_map_pointcloud": [_vp, _vp, _sz, _szp],
    "kicp_map_closest_neighbor": [_vp, _vp, _sz, _vp, _vp],
    "kicp_registration_create": [_i, _d, _i, _i, C.POINTER(_vp)],
    "kicp_registration_destroy": [_vp],
    "kicp_align_points_to_map": [_vp, _vp, _sz, _vp, _dp, _d, _d, _dp, C.POINTER(IcpStats)],
    "kicp_registration_last_system": [_vp, _dp, _dp, _u64p],
    "kicp_voxel_downsample": [_vp, _sz, _d, _i, _vp, _szp],
    "kicp_preprocess": [_vp, _sz, _vp, _sz, _dp, _d, _d, _i, _i, _vp, _szp],
    "kicp_config_default": [C.POINTER(Config)],
    "kicp_pipeline_create": [C.POINTER(Config), _i, C.POINTER(_vp)],
    "kicp_pipeline_destroy": [_vp],
    "kicp_pipeline_register_frame": [_vp, _vp, _sz, _vp, _sz],
    "kicp_pipeline_register_frame_outputs": [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _szp, _vp, _sz, _szp],
    "kicp_pipeline_register_frame_views": [_vp, _vp, _sz, _vp, _sz, C.POINTER(_vp), _szp, C.POINTER(_vp), _szp],
    "kicp_pipeline_collect_outputs": [_vp, _vp, _sz, _szp, C.POINTER(_vp), _szp],
    "kicp_pipeline_register_frame_async": [_vp, _vp, _sz, _vp, _sz],
    "kicp_pipeline_register_frame_async_f32": [_vp, _vp, _sz, _vp, _sz],
    "kicp_pipeline_register_frame_device": [_vp, _vp, _sz, _vp, _sz],
    "kicp_pipeline_sync": [_vp],
    "kicp_pipeline_synced_poses": [_vp, _vp, _sz, _szp],
    "kicp_pipeline_pose": [_vp, _dp],
    "kicp_pipeline_delta": [_vp, _dp],
    "kicp_pipeline_set_pose": [_vp, _dp],
    "kicp_pipeline_set_delta": [_vp, _dp],
    "kicp_pipeline_map": [_vp, C.POINTER(_vp)],
    "kicp_pipeline_output_size": [_vp, _i, _szp],
    "kicp_pipeline_output": [_vp, _i, _vp, _sz, _szp],
    "kicp_pipeline_voxelize": [_vp, _vp, _sz, _vp, _szp, _vp, _szp],
    "kicp_pipeline_last_stats": [_vp, C.POINTER(FrameStats)],
    "kicp_pipeline_icp_timing": [_vp, _dp, _u64p, _u64p, _u64p, _i],
    "kicp_pipeline_icp_profile": [_vp, _u64p, C.POINTER(_i)],
    "kicp_pipeline_icp_iteration_profile": [_vp, _vp, _i, C.POINTER(_i)],
    "kicp_pipeline_icp_clock": [_vp, _u64p, _u64p],
    "kicp_pipeline_icp_first_iteration": [_vp, _u64p, _u64p, C.POINTER(_i)],
    "kicp_pipeline_icp_group_profile": [_vp, _vp, _sz, C.POINTER(_i), C.POINTER(_i)],
    "kicp_pipeline_host_stats": [_vp, C.POINTER(HostStats), _i],
    "kicp_pipeline_stream": [_vp, C.POINTER(_vp)],
    "kicp_device_alloc": [_i, _sz, C.POINTER(_vp)],
    "kicp_device_free": [_i, _vp],
    "kicp_device_upload": [_i, _vp, _vp, _sz],
    "kicp_device_download": [_i, _vp, _vp, _sz],
    "kicp_device_synchronize": [_i],
    "kicp_selftest_solve": [_i, _vp, _vp, _sz, _vp],
    "kicp_selftest_tile_sort": [_i, _vp, _sz, C.c_double, _sz, _sz, _vp],
    "kicp_selftest_narrow": [_vp, _sz, _vp, C.POINTER(_i)],
    "kicp_batch_unique_id": [_vp],
    "kicp_batch_create": [C.POINTER(Config), C.POINTER(_i), _i, _i, _i, _vp, _vp, _sz, C.POINTER(_vp)],
    "kicp_batch_destroy": [_vp],
    "kicp_batch_register_frames": [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz)],
    "kicp_batch_register_frames_f32": [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz)],
    "kicp_batch_sync": [_vp],
    "kicp_batch_poses": [_vp, _i, _vp, _sz, C.POINTER(_sz)],
    "kicp_batch_pipeline": [_vp, _i, C.POINTER(_vp)],
    "kicp_batch_gather_seconds": [_vp, _dp],
}


class BatchComm(C.Structure):
    """kicp_batch_comm (include/kicp.h): a communicator supplied by the host instead of RCCL"""
    INIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
    ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    FINALIZE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
    _fields_ = [("ctx", C.c_void_p), ("init", INIT), ("all_gather", ALL_GATHER), ("finalize", FINALIZE)]


_STRING_FUNCS = ("kicp_status_string", "kicp_last_error")

_lib = None


def lib():
    """Load libkicp.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C kiss-icp_amd/csrc)"
        )
    L = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = args
    L.kicp_status_string.restype = C.c_char_p
    L.kicp_status_string.argtypes = [C.c_int]
    L.kicp_last_error.restype = C.c_char_p
    L.kicp_last_error.argtypes = []
    _lib = L
    return L


def check(status):
    if status != KICP_OK:
        L = lib()
        msg = L.kicp_last_error().decode() or L.kicp_status_string(status).decode()
        raise KicpError(status, msg)


def points(a):
    """(N,3) float64 C-contiguous view/copy -- what _Vector3dVector(np.ndarray) accepts
    (pybind/stl_vector_eigen.h:67-80: forcecast to double, else cast_error)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise TypeError("expected an (N, 3) array")  # py::cast_error in the reference
    return a


def mat4(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise TypeError("expected a 4x4 matrix")
    return T


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def dptr(a):
    return a.ctypes.data_as(_dp)


def device_count():
    n = C.c_int(0)
    check(lib().kicp_device_count(C.byref(n)))
    return n.value


def device_name(device_id=0):
    buf = C.create_string_buffer(256)
    check(lib().kicp_device_name(device_id, buf, 256))
    return buf.value.decode()


def set_option(name, value):
    check(lib().kicp_set_option(name.encode(), int(value)))


class DeviceArray:
    """a host numpy array staged in HBM through the C-ABI's device-memory helpers"""

    def __init__(self, host, device_id=0):
        host = np.ascontiguousarray(host)
        self.device_id, self.shape, self.dtype, self.nbytes = device_id, host.shape, host.dtype, host.nbytes
        p = C.c_void_p()
        check(lib().kicp_device_alloc(device_id, host.nbytes, C.byref(p)))
        self.ptr = p
        check(lib().kicp_device_upload(device_id, p, ptr(host), host.nbytes))

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().kicp_device_download(self.device_id, ptr(out), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().kicp_device_free(self.device_id, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
