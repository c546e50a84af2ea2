#!/usr/bin/env python
"""Generates the fixtures in tests/golden/ -- TEST INFRASTRUCTURE.

The reference (PRBonn/kiss-icp v1.2.3) holds no golden vectors for the registration path.  The EXPECTED
VALUES stored here (survivors, map content, neighbours, poses) are therefore produced by the reference's
own sources: oracle/_ref/libkiss_ref.so = /root/reference/cpp/kiss_icp/{core,pipeline}/*.cpp compiled
unmodified against stand-in third-party headers (oracle/ref_build/, oracle/ref.py).  While generating,
every value is cross-checked against the CPU oracle (oracle/kiss_oracle.c) and the independent naive
restatement tests/naive_ref.py on the very same inputs (asserted below).  The reference's C++ API does
not report the ICP iteration count, so that one field comes from the oracle (== the naive restatement).
tests/test_oracle.py checks the oracle against these files on CPU, tests/test_gpu_parity.py checks the
HIP path against them on the GPU box (where /root/reference does not exist).  Needs /root/reference (or
an already built oracle/_ref/libkiss_ref.so).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "kiss-icp_amd", "python"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import naive_ref as N  # noqa: E402
from helpers import make_pose, pose_error, random_cloud, sort_rows  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import ref as R  # noqa: E402

SOURCE = "reference sources (cpp/kiss_icp v1.2.3) built by oracle/ref_build"


def downsample():
    rng = np.random.default_rng(101)
    pts = random_cloud(rng, 3000, extent=12.0, z_extent=2.0)
    out = {"points": pts}
    for v, key in ((0.5, "out_050"), (1.5, "out_150")):
        # the reference's order: bucket order of the tsl::robin_map it iterates (VoxelUtils.cpp:17-19)
        o = R.voxel_down_sample(pts, v)
        assert np.array_equal(o, N.voxel_downsample(pts, v)) and np.array_equal(o, O.voxel_down_sample(pts, v))
        out[key] = o
        # ... and ascending original index (option downsample_order = 0 of the HIP path; the reference cannot produce it)
        old = O.set_downsample_order(O.INDEX_ORDER)
        try:
            oi = O.voxel_down_sample(pts, v)
        finally:
            O.set_downsample_order(old)
        assert np.array_equal(oi, N.voxel_downsample(pts, v, order="index")) and np.array_equal(sort_rows(oi), sort_rows(o))
        out[key + "_index"] = oi
    np.savez_compressed(os.path.join(HERE, "downsample.npz"), source=SOURCE, **out)


def map_nn():
    rng = np.random.default_rng(102)
    om, nm, rm = O.VoxelHashMap(1.0, 30.0, 20), N.VoxelHashMap(1.0, 30.0, 20), R.VoxelHashMap(1.0, 30.0, 20)
    out = {"n_updates": 4, "source": SOURCE}
    for k in range(4):
        pts = random_cloud(rng, 1200, extent=14.0, z_extent=2.0)
        T = make_pose((6.0 * k, 0.7 * k, 0.0), (0.0, 0.0, 0.08 * k))
        om.update(pts, T)
        nm.update(pts, T)
        rm.update(pts, T)
        out[f"pts_{k}"], out[f"pose_{k}"] = pts, T
    cloud = sort_rows(rm.point_cloud())
    assert np.array_equal(cloud, sort_rows(om.point_cloud()))
    np.testing.assert_allclose(cloud, sort_rows(nm.point_cloud()), rtol=0, atol=1e-12)
    q = random_cloud(rng, 400, extent=40.0, z_extent=3.0)
    q[:, 0] += 9.0
    nn = np.array([rm.closest_neighbor(x)[0] for x in q])
    dd = np.array([rm.closest_neighbor(x)[1] for x in q])
    assert np.array_equal(nn, np.array([om.closest_neighbor(x)[0] for x in q]))
    assert np.array_equal(dd, np.array([om.closest_neighbor(x)[1] for x in q]))
    for x, a, b in zip(q, nn, dd):
        na, nb = nm.closest_neighbor(x)
        assert abs(nb - b) <= 1e-12 * max(1.0, min(b, 1e3)) and np.allclose(na, a, rtol=0, atol=1e-12)
    out.update(cloud_sorted=cloud, queries=q, nn=nn, dist=dd)
    np.savez_compressed(os.path.join(HERE, "map_nn.npz"), **out)


def align():
    rng = np.random.default_rng(103)
    world = np.concatenate([
        np.stack([rng.uniform(-9, 9, 1200), rng.uniform(-9, 9, 1200), rng.normal(0, 0.01, 1200)], axis=1),
        np.stack([np.full(600, 7.0) + rng.normal(0, 0.01, 600), rng.uniform(-9, 9, 600), rng.uniform(0, 4, 600)], axis=1),
        np.stack([rng.uniform(-9, 9, 600), np.full(600, -6.0) + rng.normal(0, 0.01, 600), rng.uniform(0, 4, 600)], axis=1),
    ])
    om, nm, rm = O.VoxelHashMap(1.0, 100.0, 20), N.VoxelHashMap(1.0, 100.0, 20), R.VoxelHashMap(1.0, 100.0, 20)
    om.add_points(world)
    nm.add_points(world)
    rm.add_points(world)
    T_true = make_pose((0.3, -0.2, 0.04), (0.01, -0.015, 0.04))
    frame = (world[rng.choice(len(world), 300, replace=False)] - T_true[:3, 3]) @ T_true[:3, :3]
    guess = make_pose((0.05, 0.0, 0.0), (0.0, 0.0, 0.005))
    reg = O.Registration(500, 1e-4, 1)
    T = reg.align_points_to_map(frame, om, guess, 3.0, 1.0)
    Tn, it = N.align_points_to_map(frame, nm, guess, 3.0, 1.0)
    assert it == reg.last_stats["iterations"], (it, reg.last_stats)
    dt, dr = pose_error(T, Tn)
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)
    Tr = R.align_points_to_map(frame, rm, guess, 3.0, 1.0, 500, 1e-4)
    dt, dr = pose_error(T, Tr)
    assert dt < 1e-11 and dr < 1e-11, (dt, dr)
    T = Tr
    np.savez_compressed(os.path.join(HERE, "align.npz"), source=SOURCE, world=world, frame=frame, guess=guess, max_dist=3.0, kernel=1.0,
                        T=T, iterations=reg.last_stats["iterations"])


def sequence():
    from kiss_icp_amd.datasets import kitti_like

    seed, n_frames, beams, az = 5, 12, 16, 200
    ds = kitti_like(seed=seed, n_frames=n_frames, beams=beams, azimuth_steps=az)
    ko, kn, kr = O.KissICP(deskew=0, max_num_threads=1), N.KissICP(deskew=False), R.KissICP(deskew=0)
    out = {"source": SOURCE, "seed": seed, "n_frames": n_frames, "beams": beams, "azimuth_steps": az}
    poses, iters = [], []
    for i in range(n_frames):
        pts, ts = ds[i]
        ko.register_frame(pts, ts)
        kn.register_frame(pts, ts)
        dt, dr = pose_error(ko.last_pose, kn.last_pose)
        assert dt < 1e-9 and dr < 1e-9, (i, dt, dr)
        kr.register_frame(pts, ts)
        dt, dr = pose_error(ko.last_pose, kr.last_pose)
        assert dt < 1e-10 and dr < 1e-10, (i, dt, dr)
        out[f"scan_{i}"] = pts
        poses.append(kr.last_pose)
        iters.append(ko.last_stats()["iterations"])
    out["poses"] = np.array(poses)
    out["iterations"] = np.array(iters)
    # the same drive with VoxelDownsample emitting by ascending index (oracle == naive; not a reference behaviour)
    old = O.set_downsample_order(O.INDEX_ORDER)
    try:
        ko, kn = O.KissICP(deskew=0, max_num_threads=1), N.KissICP(deskew=False, downsample_order="index")
        poses_i, iters_i = [], []
        for i in range(n_frames):
            pts, ts = ds[i]
            ko.register_frame(pts, ts)
            kn.register_frame(pts, ts)
            dt, dr = pose_error(ko.last_pose, kn.last_pose)
            assert dt < 1e-9 and dr < 1e-9, (i, dt, dr)
            poses_i.append(ko.last_pose)
            iters_i.append(ko.last_stats()["iterations"])
    finally:
        O.set_downsample_order(old)
    out["poses_index_order"] = np.array(poses_i)
    out["iterations_index_order"] = np.array(iters_i)
    np.savez_compressed(os.path.join(HERE, "sequence.npz"), **out)


if __name__ == "__main__":
    downsample()
    map_nn()
    align()
    sequence()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
