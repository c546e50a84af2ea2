"""KissICP odometry pipelines with the reference's Python interface
(python/kiss_icp/kiss_icp.py:33-80: register_frame(frame, timestamps) -> (frame, source),
last_pose, last_delta, local_map, voxelize).

* ``KissICP``          the fused device pipeline (kicp_pipeline_*, include/kicp.h), a restatement
                       of pipeline::KissICP::RegisterFrame (cpp/kiss_icp/pipeline/KissICP.cpp:35-68)
                       in which every stage is a HIP kernel and all state stays in HBM.
* ``KissICPComposed``  the reference's own Python composition, stage by stage over the standalone
                       C-ABI calls (host round trip between stages) -- slower, used to show that
                       the pieces behave like the reference's pybind classes.
"""
import ctypes as C

import numpy as np

from . import _cabi
from .config import KISSConfig
from .mapping import VoxelHashMap, get_voxel_hash_map
from .preprocess import get_preprocessor
from .registration import get_registration
from .threshold import get_threshold_estimator
from .voxelization import voxel_down_sample


def _c_config(config: KISSConfig) -> _cabi.Config:
    c = _cabi.Config()
    _cabi.check(_cabi.lib().kicp_config_default(C.byref(c)))
    c.voxel_size = config.mapping.voxel_size
    c.max_range = config.data.max_range
    c.min_range = config.data.min_range
    c.max_points_per_voxel = config.mapping.max_points_per_voxel
    c.min_motion_th = config.adaptive_threshold.min_motion_th
    c.initial_threshold = config.adaptive_threshold.initial_threshold
    c.max_num_iterations = config.registration.max_num_iterations
    c.convergence_criterion = config.registration.convergence_criterion
    c.max_num_threads = config.registration.max_num_threads
    c.deskew = int(bool(config.data.deskew))
    return c


class KissICP:
    def __init__(self, config: KISSConfig = None, device_id: int = 0):
        self.config = config if config is not None else KISSConfig()
        if self.config.adaptive_threshold.fixed_threshold is not None:
            raise ValueError("fixed_threshold is a Python-only option of the reference; use KissICPComposed")
        cc = _c_config(self.config)
        h = C.c_void_p()
        _cabi.check(_cabi.lib().kicp_pipeline_create(C.byref(cc), device_id, C.byref(h)))
        self._h = h
        self.device_id = device_id

    def __del__(self):
        if getattr(self, "_h", None):
            _cabi.lib().kicp_pipeline_destroy(self._h)
            self._h = None

    # -- reference interface -------------------------------------------------------------------
    def register_frame(self, frame, timestamps=()):
        pts = _cabi.points(frame)
        ts = np.ascontiguousarray(np.asarray(timestamps, dtype=np.float64).ravel())
        n = len(pts)
        pre, src = np.empty((n, 3)), np.empty((n, 3))
        n_pre, n_src = C.c_size_t(0), C.c_size_t(0)
        st = _cabi.lib().kicp_pipeline_register_frame_outputs(self._h, _cabi.ptr(pts), n, _cabi.ptr(ts) if len(ts) else None, len(ts),
                                                              _cabi.ptr(pre), n, C.byref(n_pre), _cabi.ptr(src), n, C.byref(n_src))
        if st == 8:
            raise IndexError(_cabi.lib().kicp_last_error().decode())
        _cabi.check(st)
        return pre[: n_pre.value], src[: n_src.value].copy()

    def voxelize(self, iframe):
        frame_downsample = voxel_down_sample(iframe, self.config.mapping.voxel_size * 0.5, self.device_id)
        source = voxel_down_sample(frame_downsample, self.config.mapping.voxel_size * 1.5, self.device_id)
        return source, frame_downsample

    @property
    def last_pose(self):
        T = np.empty((4, 4))
        _cabi.check(_cabi.lib().kicp_pipeline_pose(self._h, _cabi.dptr(T)))
        return T

    @last_pose.setter
    def last_pose(self, T):
        _cabi.check(_cabi.lib().kicp_pipeline_set_pose(self._h, _cabi.dptr(_cabi.mat4(T))))

    @property
    def last_delta(self):
        T = np.empty((4, 4))
        _cabi.check(_cabi.lib().kicp_pipeline_delta(self._h, _cabi.dptr(T)))
        return T

    @last_delta.setter
    def last_delta(self, T):
        _cabi.check(_cabi.lib().kicp_pipeline_set_delta(self._h, _cabi.dptr(_cabi.mat4(T))))

    @property
    def local_map(self):
        h = C.c_void_p()
        _cabi.check(_cabi.lib().kicp_pipeline_map(self._h, C.byref(h)))
        return VoxelHashMap(0, 0, 0, _borrowed=h, _owner=self)  # the handle lives as long as this pipeline

    # -- device-side extras ---------------------------------------------------------------------
    def output(self, which):
        n = C.c_size_t(0)
        _cabi.check(_cabi.lib().kicp_pipeline_output_size(self._h, which, C.byref(n)))
        out = np.empty((n.value, 3))
        _cabi.check(_cabi.lib().kicp_pipeline_output(self._h, which, _cabi.ptr(out), n.value, C.byref(n)))
        return out

    def register_frame_async(self, frame, timestamps=()):
        """queue a host scan without waiting (kicp_pipeline_register_frame_async): `frame` is (N,3) float64
        or float32 (the sensor's native precision: no widening on the host); the arrays are free again when
        the call returns.  sync() waits, synced_poses() / last_pose give the results."""
        frame = np.asarray(frame)
        ts = np.ascontiguousarray(np.asarray(timestamps, dtype=np.float64).ravel())
        L = _cabi.lib()
        if frame.dtype == np.float32:
            pts = np.ascontiguousarray(frame)
            if pts.ndim != 2 or pts.shape[1] != 3:
                raise TypeError("expected an (N, 3) array")
            st = L.kicp_pipeline_register_frame_async_f32(self._h, _cabi.ptr(pts), len(pts), _cabi.ptr(ts) if len(ts) else None, len(ts))
        else:
            pts = _cabi.points(frame)
            st = L.kicp_pipeline_register_frame_async(self._h, _cabi.ptr(pts), len(pts), _cabi.ptr(ts) if len(ts) else None, len(ts))
        if st == 8:
            raise IndexError(L.kicp_last_error().decode())
        _cabi.check(st)

    def register_frame_device(self, d_xyz_ptr, n, d_ts_ptr=None, n_ts=0):
        """enqueue a frame whose points already live in this GPU's HBM (raw device pointers);
        returns immediately, call sync() to wait"""
        st = _cabi.lib().kicp_pipeline_register_frame_device(self._h, d_xyz_ptr, n, d_ts_ptr, n_ts)
        if st == 8:
            raise IndexError(_cabi.lib().kicp_last_error().decode())
        _cabi.check(st)

    def sync(self):
        _cabi.check(_cabi.lib().kicp_pipeline_sync(self._h))

    def synced_poses(self):
        n = C.c_size_t(0)
        _cabi.check(_cabi.lib().kicp_pipeline_synced_poses(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 4, 4))
        _cabi.check(_cabi.lib().kicp_pipeline_synced_poses(self._h, _cabi.ptr(out), n.value, C.byref(n)))
        return out

    def last_stats(self):
        s = _cabi.FrameStats()
        _cabi.check(_cabi.lib().kicp_pipeline_last_stats(self._h, C.byref(s)))
        return s.asdict()

    def icp_timing(self, reset=False):
        ms = C.c_double(0)
        launches, iters, nbytes = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _cabi.check(_cabi.lib().kicp_pipeline_icp_timing(self._h, C.byref(ms), C.byref(launches), C.byref(iters), C.byref(nbytes), int(reset)))
        return {"total_ms": ms.value, "launches": launches.value, "iterations": iters.value, "algorithmic_bytes": nbytes.value}

    def icp_profile(self):
        """shader-clock cycles of the last ICP launch's phases (workgroup 0) and workgroups used"""
        cyc = (C.c_uint64 * 4)()
        wg = C.c_int(0)
        _cabi.check(_cabi.lib().kicp_pipeline_icp_profile(self._h, cyc, C.byref(wg)))
        names = ("associate_accumulate", "reduce_publish", "gather", "solve_update")
        return {"workgroups": wg.value, **{n: int(c) for n, c in zip(names, cyc)}}

    def icp_iteration_profile(self):
        """(n_iters, 6) uint32, 10 ns ticks: wg0 {associate, publish, gather, solve}, slowest group's
        associate over all workgroups, gather polling passes"""
        buf = np.zeros((24, 6), dtype=np.uint32)
        n = C.c_int(0)
        _cabi.check(_cabi.lib().kicp_pipeline_icp_iteration_profile(self._h, _cabi.ptr(buf), 24, C.byref(n)))
        return buf[: n.value]

    def icp_clock(self):
        """(shader cycles, 10 ns ticks) of the last ICP launch, workgroup 0"""
        cyc, tk = C.c_uint64(0), C.c_uint64(0)
        _cabi.check(_cabi.lib().kicp_pipeline_icp_clock(self._h, C.byref(cyc), C.byref(tk)))
        return cyc.value, tk.value

    def icp_first_iteration(self):
        """(us to the end of the first iteration, us total, iterations) of the last ICP launch"""
        a, b, n = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        _cabi.check(_cabi.lib().kicp_pipeline_icp_first_iteration(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value / 100.0, b.value / 100.0, n.value

    def icp_group_profile(self):
        """(n_iters, n_groups, 9) int64 of the last ICP launch ("icp_profile" option on): 10 ns ticks
        {wait-in, transform + window test, window fill, scan} of the group's first point, staged points, examined
        points, path, points in the workgroup's run, ticks the group spent searching in the whole iteration"""
        buf = np.zeros(24 * 256 * 16 * 4, dtype=np.uint32)
        ni, ng = C.c_int(0), C.c_int(0)
        _cabi.check(_cabi.lib().kicp_pipeline_icp_group_profile(self._h, _cabi.ptr(buf), buf.size, C.byref(ni), C.byref(ng)))
        raw = buf[: ni.value * ng.value * 4].reshape(ni.value, ng.value, 4).astype(np.int64)
        out = np.zeros((ni.value, ng.value, 9), dtype=np.int64)
        out[..., 0], out[..., 1] = raw[..., 0] & 0xFFFF, raw[..., 0] >> 16
        out[..., 2], out[..., 3] = raw[..., 1] & 0xFFFF, raw[..., 1] >> 16
        out[..., 4], out[..., 5] = raw[..., 2] & 0xFFFF, raw[..., 2] >> 16
        out[..., 6] = raw[..., 3] & 15
        out[..., 7], out[..., 8] = (raw[..., 3] >> 4) & 4095, raw[..., 3] >> 16
        return out

    def host_stats(self, reset=False):
        """kicp_pipeline_host_stats: waits the asynchronous entries took on the host, and where their time went"""
        s = _cabi.HostStats()
        _cabi.check(_cabi.lib().kicp_pipeline_host_stats(self._h, C.byref(s), int(reset)))
        return s.asdict()

    def stream(self):
        s = C.c_void_p()
        _cabi.check(_cabi.lib().kicp_pipeline_stream(self._h, C.byref(s)))
        return s.value


class KissICPComposed:
    """python/kiss_icp/kiss_icp.py:33-80, line for line in behaviour, over our wrappers"""

    def __init__(self, config: KISSConfig = None, device_id: int = 0):
        self.last_pose = np.eye(4)
        self.last_delta = np.eye(4)
        self.config = config if config is not None else KISSConfig()
        self.adaptive_threshold = get_threshold_estimator(self.config)
        self.preprocessor = get_preprocessor(self.config, device_id)
        self.registration = get_registration(self.config, device_id)
        self.local_map = get_voxel_hash_map(self.config, device_id)
        self.device_id = device_id

    def register_frame(self, frame, timestamps=()):
        frame = self.preprocessor.preprocess(frame, np.asarray(timestamps, dtype=np.float64), self.last_delta)
        source, frame_downsample = self.voxelize(frame)
        sigma = self.adaptive_threshold.get_threshold()
        initial_guess = self.last_pose @ self.last_delta
        new_pose = self.registration.align_points_to_map(
            points=source,
            voxel_map=self.local_map,
            initial_guess=initial_guess,
            max_correspondance_distance=3 * sigma,
            kernel=sigma,
        )
        model_deviation = np.linalg.inv(initial_guess) @ new_pose
        self.adaptive_threshold.update_model_deviation(model_deviation)
        self.local_map.update(frame_downsample, new_pose)
        self.last_delta = np.linalg.inv(self.last_pose) @ new_pose
        self.last_pose = new_pose
        return frame, source

    def voxelize(self, iframe):
        frame_downsample = voxel_down_sample(iframe, self.config.mapping.voxel_size * 0.5, self.device_id)
        source = voxel_down_sample(frame_downsample, self.config.mapping.voxel_size * 1.5, self.device_id)
        return source, frame_downsample
