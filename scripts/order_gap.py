#!/usr/bin/env python
"""How far apart are the trajectories that the two VoxelDownsample output orders give?  (CPU only, oracle.)
The reference emits the survivors of VoxelDownsample in tsl::robin_map bucket order; rounds 1-2 of this repository
defined ascending original index.  Same scans, same everything else."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kiss-icp_amd", "python"))
from kiss_icp_amd.datasets import generate_scans, kitti_like, kitti_like_vegetated  # noqa: E402
from oracle import oracle as O  # noqa: E402


def drive(scans, order):
    O.set_downsample_order(order)
    k = O.KissICP(deskew=0)
    poses = []
    for p, t in scans:
        k.register_frame_noout(np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(t, dtype=np.float64))
        poses.append(k.last_pose.copy())
    return poses


def gap(a, b):
    D = np.linalg.inv(a) @ b
    return float(np.linalg.norm(D[:3, 3])), float(np.arccos(min(1.0, max(-1.0, (np.trace(D[:3, :3]) - 1.0) / 2.0))))


for name, factory, kw, n in (("vegetated 64x2048", kitti_like_vegetated, {}, 40), ("bare street 64x2048", kitti_like, {}, 40),
                             ("32x512", kitti_like, dict(beams=32, azimuth_steps=512), 40)):
    scans = generate_scans(factory, dict(kw, seed=0, n_frames=n), range(n))
    ref, idx = drive(scans, O.REFERENCE_ORDER), drive(scans, O.INDEX_ORDER)
    g = np.array([gap(a, b) for a, b in zip(ref, idx)])
    print(f"{name}: {n} frames; trajectories apart (reference order vs index order): after 12 frames {g[11,0]*100:.2f} cm / {g[11,1]*1e3:.2f} mrad, "
          f"max over the drive {g[:,0].max()*100:.2f} cm / {g[:,1].max()*1e3:.2f} mrad, at the end {g[-1,0]*100:.2f} cm / {g[-1,1]*1e3:.2f} mrad")
O.set_downsample_order(O.REFERENCE_ORDER)
