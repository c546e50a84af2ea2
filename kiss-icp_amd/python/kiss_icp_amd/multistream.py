"""Multi-stream batch mode: S independent LiDAR streams, one KissICP pipeline + one local map per
GPU, one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

A single stream cannot be sharded over frames (frame k needs pose k-1 and the map containing
frame k-1: cpp/kiss_icp/pipeline/KissICP.cpp:47,61), so streams are the unit of parallelism and the
data path needs no collective at all.  The only exchange is the "pose-graph sync": after every
batch of frames each rank all-gathers its new poses (128 B per frame and stream) so that every
rank holds all S trajectories.  The reference has no such mode; this is new functionality.

Nothing here touches the device directly: a "pipeline" is any object with
register_frame_device / sync / synced_poses (the HIP pipeline) -- the CPU gloo tests plug a
stand-in.
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
