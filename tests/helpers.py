"""shared helpers of the test-suite"""
import numpy as np


def pose_error(T_a, T_b):
    """(translation error [m], rotation angle [rad]) of inv(T_a) @ T_b"""
    D = np.linalg.inv(T_a) @ T_b
    c = min(1.0, max(-1.0, (np.trace(D[:3, :3]) - 1.0) / 2.0))
    s = 0.5 * np.sqrt((D[2, 1] - D[1, 2]) ** 2 + (D[0, 2] - D[2, 0]) ** 2 + (D[1, 0] - D[0, 1]) ** 2)
    return float(np.linalg.norm(D[:3, 3])), float(np.arctan2(s, c))


def make_pose(t=(0, 0, 0), rpy=(0, 0, 0)):
    from scipy.spatial.transform import Rotation

    T = np.eye(4)
    T[:3, :3] = Rotation.from_euler("xyz", rpy).as_matrix()
    T[:3, 3] = t
    return T


def sort_rows(a):
    a = np.asarray(a)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def random_cloud(rng, n, extent=30.0, z_extent=4.0):
    return np.stack([rng.uniform(-extent, extent, n), rng.uniform(-extent, extent, n), rng.uniform(-z_extent, z_extent, n)], axis=1)
