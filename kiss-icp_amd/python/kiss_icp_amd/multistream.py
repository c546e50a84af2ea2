"""Multi-stream batch mode: S independent LiDAR streams, one KissICP pipeline + one local map per
GPU, one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

A single stream cannot be sharded over frames (frame k needs pose k-1 and the map containing
frame k-1: cpp/kiss_icp/pipeline/KissICP.cpp:47,61), so streams are the unit of parallelism and the
data path needs no collective at all.  The only exchange is the "pose-graph sync": after every
batch of frames each rank all-gathers its new poses (128 B per frame and stream) so that every
rank holds all S trajectories.  The reference has no such mode; this is new functionality.

Two hosts of the same mode:
  * the functions below drive it from Python, one process per GPU under torchrun, the exchange through
    torch.distributed (what bench.py --gpus N does; a "pipeline" is any object with register_frame_device /
    sync / synced_poses -- the CPU gloo tests plug a stand-in);
  * StreamBatch is the face of the C-ABI's own batch entry (kicp_batch_*, include/kicp.h): worker threads
    inside libkicp.so, one per stream and bound to its GPU, the exchange by RCCL called directly -- the form a
    C++ / Go / Java host uses, with no Python and no torch in it.
"""
import os

import numpy as np


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def stream_seed(base_seed, rank):
    """sequence id handled by a rank: rank r runs stream r (weak scaling: one stream per GPU)"""
    return int(base_seed) + int(rank)


def init_process_group(backend=None):
    """initialise torch.distributed from the environment when WORLD_SIZE > 1; returns the module
    (or None for a single process).  Rendezvous on 127.0.0.1 unless told otherwise."""
    rank, local_rank, world = dist_env()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def gather_poses(local_poses, dist=None, device=None):
    """all-gather a (K,4,4) float64 block of poses; returns (world, K, 4, 4).
    One collective per batch of frames, never per ICP iteration (a 128-byte message is pure
    latency on xGMI: ~10-20 us per RCCL launch)."""
    import torch

    local = np.ascontiguousarray(local_poses, dtype=np.float64)
    if dist is None:
        return local[None]
    t = torch.from_numpy(local)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def max_over_ranks(value, dist=None, device=None):
    import torch

    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(dist=None):
    if dist is not None:
        dist.barrier()


def run_batch(pipeline, device_frames, dist=None, device=None):
    """enqueue a batch of device-resident frames [(ptr, n, ts_ptr, n_ts), ...] on one pipeline,
    wait for it, and exchange the new poses.  Returns (local_poses (K,4,4), all_poses (S,K,4,4))."""
    for ptr, n, ts_ptr, n_ts in device_frames:
        pipeline.register_frame_device(ptr, n, ts_ptr, n_ts)
    pipeline.sync()
    local = pipeline.synced_poses()
    return local, gather_poses(local, dist, device)


def run_batch_host(pipeline, host_frames, dist=None, device=None):
    """the same for scans in host memory [(points, timestamps), ...]: each is handed to the pipeline's
    asynchronous host-input entry (staged, uploaded under the previous frame's registration, queued)"""
    for pts, ts in host_frames:
        pipeline.register_frame_async(pts, ts)
    pipeline.sync()
    local = pipeline.synced_poses()
    return local, gather_poses(local, dist, device)


class StreamBatch:
    """kicp_batch_* (include/kicp.h): S streams on the GPUs `devices` (one worker thread per stream inside the
    library), poses all-gathered by RCCL at every sync().  One process may own all ranks (the default) or the
    consecutive ranks first_rank .. first_rank + len(devices) - 1 of n_total, with the 128-byte `unique_id` of
    StreamBatch.unique_id() handed to every process by the launcher."""

    def __init__(self, config, devices, first_rank=0, n_total=None, unique_id=None, comm=None, frames_per_gather=0):
        import ctypes as C

        from . import _cabi
        from .kiss_icp import _c_config

        self._lib = _cabi.lib()
        self._C = C
        self.devices = [int(d) for d in devices]
        self.n_local = len(self.devices)
        self.n_total = self.n_local if n_total is None else int(n_total)
        self.first_rank = int(first_rank)
        cfg = _c_config(config)
        dev = (C.c_int * self.n_local)(*self.devices)
        uid = (C.c_ubyte * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        self._comm = comm  # keep the callbacks alive
        h = C.c_void_p()
        _cabi.check(self._lib.kicp_batch_create(C.byref(cfg), dev, self.n_local, self.first_rank, self.n_total, uid,
                                                C.byref(comm) if comm is not None else None, frames_per_gather, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        import ctypes as C

        from . import _cabi

        buf = (C.c_ubyte * 128)()
        _cabi.check(_cabi.lib().kicp_batch_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kicp_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def register_frames(self, scans, timestamps=None):
        """scans[i]: (n,3) float64 or float32 array for local stream i, or None; all of one dtype"""
        C = self._C
        from . import _cabi

        assert len(scans) == self.n_local
        arrs = [None if s is None else np.ascontiguousarray(s) for s in scans]
        kinds = {a.dtype for a in arrs if a is not None}
        f32 = kinds == {np.dtype(np.float32)}
        if not f32:
            arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
        tss = [None] * self.n_local if timestamps is None else [
            None if t is None or len(t) == 0 else np.ascontiguousarray(t, dtype=np.float64).ravel() for t in timestamps]
        xyz = (C.c_void_p * self.n_local)(*[None if a is None else a.ctypes.data for a in arrs])
        n = (C.c_size_t * self.n_local)(*[0 if a is None else len(a) for a in arrs])
        ts = (C.c_void_p * self.n_local)(*[None if t is None else t.ctypes.data for t in tss])
        nts = (C.c_size_t * self.n_local)(*[0 if t is None else len(t) for t in tss])
        fn = self._lib.kicp_batch_register_frames_f32 if f32 else self._lib.kicp_batch_register_frames
        _cabi.check(fn(self._h, xyz, n, ts, nts))

    def sync(self):
        from . import _cabi

        _cabi.check(self._lib.kicp_batch_sync(self._h))

    def poses(self, rank):
        """(K,4,4) poses global rank `rank` completed between the last two syncs"""
        C = self._C
        from . import _cabi

        n = C.c_size_t()
        _cabi.check(self._lib.kicp_batch_poses(self._h, rank, None, 0, C.byref(n)))
        out = np.empty((n.value, 4, 4))
        if n.value:
            _cabi.check(self._lib.kicp_batch_poses(self._h, rank, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def gather_seconds(self):
        C = self._C
        from . import _cabi

        s = C.c_double()
        _cabi.check(self._lib.kicp_batch_gather_seconds(self._h, C.byref(s)))
        return s.value
