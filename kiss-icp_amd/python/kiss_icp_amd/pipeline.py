"""Pose files of a registered sequence -- the two writers python/kiss_icp/pipeline.py:116-134 offers (KITTI and TUM
formats), so that a trajectory of this library can go through the same evaluation tools as the reference's.  Nothing
else of the reference's OdometryPipeline is mirrored here: the dataset loop, result directories, progress bar and CLI
are application glue outside the registration path; bench.py and the tests drive the
pipeline directly."""
import numpy as np


def rotation_to_quaternion_wxyz(R: np.ndarray) -> np.ndarray:
    """unit quaternion (w, x, y, z), w >= 0, of a rotation matrix (what pyquaternion's Quaternion(matrix=...)
    .elements gives the reference, up to the sign convention w >= 0)"""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0.0:
        s = 2.0 * np.sqrt(1.0 + t)
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 0.0))
        q = np.empty(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q /= np.linalg.norm(q)
    return -q if q[0] < 0.0 else q


def save_poses_kitti_format(filename: str, poses: np.ndarray):
    """one line per pose: the 12 numbers of the upper 3x4 block, row-major (pipeline.py:116-121)"""
    np.savetxt(fname=f"{filename}_kitti.txt", X=np.asarray(poses)[:, :3].reshape(-1, 12))


def save_poses_tum_format(filename, poses, timestamps):
    """one line per pose: timestamp tx ty tz qx qy qz qw, 4 decimals (pipeline.py:123-134)"""
    poses = np.asarray(poses)
    tum = np.zeros((len(poses), 8))
    for i in range(len(poses)):
        qw, qx, qy, qz = rotation_to_quaternion_wxyz(poses[i, :3, :3])
        tum[i] = np.r_[float(timestamps[i]), poses[i, :3, 3], qx, qy, qz, qw]
    np.savetxt(fname=f"{filename}_tum.txt", X=tum, fmt="%.4f")
