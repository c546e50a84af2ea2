"""Every blocking entry of the C-ABI has a deadline.

The reference cannot hang (core/Registration.cpp:138-167 terminates, always); a library that waits for a device can,
unless every wait is bounded.  libkicp never calls hipStreamSynchronize / hipEventSynchronize: it polls against the
option "wait_timeout_ms" (kicp_internal.hpp: wait_stream / wait_event / wait_device) and returns KICP_ERR_TIMEOUT.  Here
a dependency that is not signalled in time is INJECTED -- option "inject_stall_ms": the next piece of work a handle queues
is preceded by a kernel that spins that long -- with a limit far below it, and every blocking entry must come back with
KICP_ERR_TIMEOUT within the limit (plus slack), leave the handle usable once the device has caught up, and tear down
without hanging.
"""
import time

import numpy as np
import pytest

from helpers import make_pose  # noqa: F401  (conftest puts tests/ on the path)

pytestmark = pytest.mark.gpu

LIMIT_MS, STALL_MS = 300, 2000
SLACK_S = 1.5  # what a call may take beyond the limit (allocation, first-use code loading; still short of the stall: a call that had waited it out would not raise at all)


@pytest.fixture()
def short_deadline(gpu):
    from kiss_icp_amd import _cabi

    _cabi.set_option("wait_timeout_ms", LIMIT_MS)
    yield _cabi
    _cabi.set_option("inject_stall_ms", 0)
    _cabi.set_option("wait_timeout_ms", 120000)
    time.sleep(0.1)


def _expect_timeout(cabi, call):
    cabi.set_option("inject_stall_ms", STALL_MS)
    t0 = time.perf_counter()
    with pytest.raises(cabi.KicpError) as e:
        call()
    took = time.perf_counter() - t0
    assert e.value.status == 6, str(e.value)  # KICP_ERR_TIMEOUT
    assert "wait_timeout_ms" in str(e.value)
    assert took < LIMIT_MS / 1000.0 + SLACK_S, took
    return took


def _scene(seed, n=4000):
    rng = np.random.default_rng(seed)
    floor = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), np.zeros(n)], axis=1)
    wall = np.stack([np.full(n, 10.0), rng.uniform(-20, 20, n), rng.uniform(0, 8, n)], axis=1)
    wall2 = np.stack([rng.uniform(-20, 20, n), np.full(n, -12.0), rng.uniform(0, 8, n)], axis=1)
    return np.concatenate([floor, wall, wall2]) + rng.normal(0.0, 0.01, size=(3 * n, 3))


def test_free_functions_give_up_in_time(short_deadline):
    cabi = short_deadline
    from kiss_icp_amd.preprocess import Preprocessor
    from kiss_icp_amd.voxelization import voxel_down_sample

    pts = _scene(0)
    want = voxel_down_sample(pts, 0.5)
    _expect_timeout(cabi, lambda: voxel_down_sample(pts, 0.5))
    time.sleep(STALL_MS / 1000.0)
    assert np.array_equal(voxel_down_sample(pts, 0.5), want)  # the device has caught up: same answer as before
    pre = Preprocessor(100.0, 0.0, False, 0)
    want = pre.preprocess(pts, np.array([]), np.eye(4))
    _expect_timeout(cabi, lambda: pre.preprocess(pts, np.array([]), np.eye(4)))
    time.sleep(STALL_MS / 1000.0)
    assert np.array_equal(pre.preprocess(pts, np.array([]), np.eye(4)), want)


def test_map_and_registration_give_up_in_time_and_recover(short_deadline):
    cabi = short_deadline
    from kiss_icp_amd.mapping import VoxelHashMap
    from kiss_icp_amd.registration import Registration

    pts = _scene(1)
    ref_map = VoxelHashMap(1.0, 100.0, 20)
    ref_map.add_points(pts)
    m = VoxelHashMap(1.0, 100.0, 20)
    _expect_timeout(cabi, lambda: m.add_points(pts))
    time.sleep(STALL_MS / 1000.0)  # the insert itself was queued behind the stall and has happened by now
    assert m.num_voxels() == ref_map.num_voxels()
    _expect_timeout(cabi, lambda: m.remove_far_away_points(np.zeros(3)))
    time.sleep(STALL_MS / 1000.0)
    _expect_timeout(cabi, lambda: m.closest_neighbor(pts[:100]))
    time.sleep(STALL_MS / 1000.0)
    nn, d = m.closest_neighbor(pts[:100])
    nn_ref, d_ref = ref_map.closest_neighbor(pts[:100])
    assert np.array_equal(nn, nn_ref) and np.array_equal(d, d_ref)
    src = pts[::7] + np.array([0.05, -0.03, 0.0])
    reg = Registration(500, 1e-4)
    want = reg.align_points_to_map(src, ref_map, np.eye(4), 3.0, 1.0)
    _expect_timeout(cabi, lambda: reg.align_points_to_map(src, ref_map, np.eye(4), 3.0, 1.0))
    time.sleep(STALL_MS / 1000.0)
    assert np.array_equal(reg.align_points_to_map(src, ref_map, np.eye(4), 3.0, 1.0), want)
    # teardown with work that does not end in time: returns (the handle's memory is leaked), does not hang
    cabi.set_option("inject_stall_ms", STALL_MS)
    t0 = time.perf_counter()
    with pytest.raises(cabi.KicpError):
        m.add_points(pts)
    del m
    assert time.perf_counter() - t0 < LIMIT_MS / 1000.0 + SLACK_S
    time.sleep(STALL_MS / 1000.0)


def test_pipeline_entries_give_up_in_time_and_the_frame_is_not_lost(short_deadline):
    cabi = short_deadline
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like
    from kiss_icp_amd.kiss_icp import KissICP

    ds = kitti_like(seed=11, n_frames=6, beams=32, azimuth_steps=512)
    cfg = load_config(deskew=False)
    ref, k = KissICP(cfg), KissICP(cfg)
    for i in range(6):
        ref.register_frame(*ds[i])
    ref_poses = []
    r2 = KissICP(cfg)
    for i in range(6):
        r2.register_frame(*ds[i])
        ref_poses.append(r2.last_pose.copy())
    for i in range(3):
        k.register_frame(*ds[i])
    # the blocking entry: gives up in time, the frame stays queued ...
    _expect_timeout(cabi, lambda: k.register_frame(*ds[3]))
    time.sleep(STALL_MS / 1000.0)
    k.sync()  # ... and a later sync collects it
    assert np.array_equal(k.last_pose, ref_poses[3])
    # the asynchronous entry + sync
    cabi.set_option("inject_stall_ms", STALL_MS)
    k.register_frame_async(*ds[4])
    t0 = time.perf_counter()
    with pytest.raises(cabi.KicpError) as e:
        k.sync()
    assert e.value.status == 6 and time.perf_counter() - t0 < LIMIT_MS / 1000.0 + SLACK_S
    with pytest.raises(cabi.KicpError):  # every getter that has to wait says the same (after a limit of its own)
        _ = k.last_pose
    time.sleep(STALL_MS / 1000.0)
    k.sync()
    assert np.array_equal(k.last_pose, ref_poses[4])
    k.register_frame(*ds[5])
    assert np.array_equal(k.last_pose, ref_poses[5]) and np.array_equal(k.last_pose, ref.last_pose)
    # teardown under a stall
    cabi.set_option("inject_stall_ms", STALL_MS)
    k.register_frame_async(*ds[5])
    t0 = time.perf_counter()
    del k
    assert time.perf_counter() - t0 < LIMIT_MS / 1000.0 + SLACK_S
    time.sleep(STALL_MS / 1000.0)


@pytest.mark.cold_libs  # (a batch without a host communicator loads RCCL)
@pytest.mark.timeout(1000)
def test_batch_sync_gives_up_in_time(short_deadline):
    cabi = short_deadline
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like
    from kiss_icp_amd.multistream import StreamBatch

    ds = kitti_like(seed=12, n_frames=3, beams=32, azimuth_steps=512)
    b = StreamBatch(load_config(deskew=False), [0])
    try:
        b.register_frames([ds[0][0]])
        b.sync()
        cabi.set_option("inject_stall_ms", STALL_MS)
        b.register_frames([ds[1][0]])
        t0 = time.perf_counter()
        with pytest.raises(cabi.KicpError) as e:
            b.sync()
        assert e.value.status == 6 and time.perf_counter() - t0 < LIMIT_MS / 1000.0 + SLACK_S
        time.sleep(STALL_MS / 1000.0)
    finally:
        b.close()


def test_batch_gives_up_on_a_peer_that_never_arrives(gpu):
    """The one wait that is not for the device: the pose exchange waits for PEERS.  Rank 0 of a two-rank job whose other
    rank never shows up, through a host communicator (the MPI hook, kicp_batch_comm) whose all-gather blocks until released:
    kicp_batch_sync returns KICP_ERR_TIMEOUT within "collective_timeout_ms", every later call says the batch is broken,
    kicp_batch_destroy returns (the handle is leaked: a thread is still inside the exchange) -- and the late release finds
    valid memory."""
    import threading

    cabi = _batch_cabi()
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like
    from kiss_icp_amd.multistream import StreamBatch

    release = threading.Event()
    entered = []

    def all_gather(ctx, rank, d_send, d_recv, nbytes, stream):
        entered.append(rank)
        release.wait(30.0)  # the peer "arrives" when the test says so
        return 0

    comm = cabi.BatchComm(None, cabi.BatchComm.INIT(0), cabi.BatchComm.ALL_GATHER(all_gather), cabi.BatchComm.FINALIZE(0))
    ds = kitti_like(seed=13, n_frames=2, beams=32, azimuth_steps=512)
    cabi.set_option("collective_timeout_ms", 400)
    try:
        b = StreamBatch(load_config(deskew=False), [0], first_rank=0, n_total=2, comm=comm)
    finally:
        cabi.set_option("collective_timeout_ms", 1800000)
    try:
        b.register_frames([ds[0][0]])
        t0 = time.perf_counter()
        with pytest.raises(cabi.KicpError) as e:
            b.sync()
        took = time.perf_counter() - t0
        assert e.value.status == 6 and "collective_timeout_ms" in str(e.value), str(e.value)
        assert 0.35 < took < 0.4 + SLACK_S, took
        assert entered == [0]
        with pytest.raises(cabi.KicpError) as e2:  # broken: every later call says so at once
            b.register_frames([ds[1][0]])
        assert e2.value.status == 6
        t1 = time.perf_counter()
        assert cabi.lib().kicp_batch_destroy(b._h) == 6  # returns; the handle is leaked on purpose
        b._h = None
        assert time.perf_counter() - t1 < SLACK_S
    finally:
        release.set()  # the abandoned worker comes home into memory that still exists
        time.sleep(0.2)


def _batch_cabi():
    from kiss_icp_amd import _cabi

    return _cabi
