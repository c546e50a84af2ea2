"""Multi-stream batch mode on CPU: world_size 2 over gloo.  The device pipeline is replaced by a
stand-in with the same three methods (register_frame_device / sync / synced_poses) that runs the
CPU oracle, so the test covers everything that is NOT a kernel: stream-to-rank assignment, batch
enqueue/sync, the per-batch pose all-gather ("pose-graph sync"), the max-over-ranks timing
reduction and the barrier."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kiss-icp_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_FRAMES = 5
SCAN = dict(beams=8, azimuth_steps=120)


class OraclePipeline:
    """stand-in for kiss_icp_amd.kiss_icp.KissICP: 'device frames' are indices into a host list"""

    def __init__(self, scans):
        from oracle import oracle as O

        self.k = O.KissICP(deskew=0, max_num_threads=1)
        self.scans = scans
        self.queue = []
        self.done = []

    def register_frame_device(self, ptr, n, ts_ptr=None, n_ts=0):
        assert n == len(self.scans[ptr][0])
        self.queue.append(ptr)

    def sync(self):
        self.done = []
        for i in self.queue:
            self.k.register_frame(*self.scans[i])
            self.done.append(self.k.last_pose)
        self.queue = []

    def synced_poses(self):
        return np.array(self.done).reshape(-1, 4, 4)


def _trajectory(seed):
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=seed, n_frames=N_FRAMES, **SCAN)
    scans = [ds[i] for i in range(N_FRAMES)]
    return scans


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from kiss_icp_amd import multistream

    r, lr, w = multistream.dist_env()
    assert (r, lr, w) == (rank, rank, world)
    dist = multistream.init_process_group("gloo")
    assert dist is not None and dist.get_world_size() == world
    seed = multistream.stream_seed(10, rank)
    scans = _trajectory(seed)
    pipe = OraclePipeline(scans)
    frames = [(i, len(scans[i][0]), None, 0) for i in range(N_FRAMES)]
    # two batches: 2 frames, then 3
    l1, a1 = multistream.run_batch(pipe, frames[:2], dist, None)
    multistream.barrier(dist)
    l2, a2 = multistream.run_batch(pipe, frames[2:], dist, None)
    assert l1.shape == (2, 4, 4) and a1.shape == (world, 2, 4, 4)
    assert l2.shape == (3, 4, 4) and a2.shape == (world, 3, 4, 4)
    assert np.array_equal(a1[rank], l1) and np.array_equal(a2[rank], l2)
    slowest = multistream.max_over_ranks(1.0 + rank, dist, None)
    assert slowest == float(world)
    np.save(os.path.join(out_dir, f"all_{rank}.npy"), np.concatenate([a1, a2], axis=1))
    multistream.barrier(dist)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_streams_over_gloo(tmp_path):
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"all_{r}.npy") for r in range(world)]
    assert np.array_equal(got[0], got[1])  # every rank holds all S trajectories
    # and they are the trajectories of streams seed+0, seed+1 computed independently here
    from oracle import oracle as O

    for r in range(world):
        k = O.KissICP(deskew=0, max_num_threads=1)
        for pts, ts in _trajectory(10 + r):
            k.register_frame(pts, ts)
        np.testing.assert_allclose(got[0][r, -1], k.last_pose, rtol=0, atol=1e-12)
    assert not np.allclose(got[0][0, -1], got[0][1, -1])  # the two streams really differ


def test_single_process_degenerates_cleanly():
    from kiss_icp_amd import multistream

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert multistream.dist_env() == (0, 0, 1)
    assert multistream.init_process_group("gloo") is None
    poses = np.tile(np.eye(4), (3, 1, 1))
    assert multistream.gather_poses(poses).shape == (1, 3, 4, 4)
    assert multistream.max_over_ranks(2.5) == 2.5
    multistream.barrier(None)


def test_cabi_batch_driver_with_stub_pipelines_and_communicator():
    """the C-ABI's own batch entry (kicp_batch_*, thread per stream, direct RCCL on the GPU) shares its
    orchestration with this CPU program, which instantiates the same driver over stand-in pipelines and a
    stand-in communicator: rank-ordered gathers, ragged / empty / multi-block batches, ranks in two processes,
    failures of a pipeline or the communicator (tests/cpp/test_batch_stub.cpp)"""
    import subprocess

    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", d, "test_batch_stub"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(d, "test_batch_stub")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


def test_cabi_batch_entry_fails_loudly_without_a_device():
    """no GPU here: the real entry must refuse (no CPU fallback), with a message naming the stream"""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.config import KISSConfig
    from kiss_icp_amd.multistream import StreamBatch

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")

    comm = _cabi.BatchComm(None, _cabi.BatchComm.INIT(0), _cabi.BatchComm.ALL_GATHER(lambda *a: 0), _cabi.BatchComm.FINALIZE(0))
    with pytest.raises(_cabi.KicpError) as e:
        StreamBatch(KISSConfig(), [0, 0], comm=comm)
    assert e.value.status == 7 and "stream 0" in str(e.value)
