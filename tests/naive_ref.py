"""A second, deliberately naive restatement of the KISS-ICP registration path in numpy/scipy with a
Python dict as the voxel map -- TEST INFRASTRUCTURE ONLY.

It exists to pin the C oracle (oracle/kiss_oracle.c): the two are written independently (this one
leans on scipy for SE(3) exp/log and numpy for the 6x6 solve, the oracle restates Sophus/Eigen by
hand), follow the same reference sources, and must agree to rounding.  Pure-Python loops: use
small inputs only.

Reference (PRBonn/kiss-icp v1.2.3):
  VoxelHashMap      cpp/kiss_icp/core/VoxelHashMap.cpp:35-132
  Registration      cpp/kiss_icp/core/Registration.cpp:55-167
  VoxelDownsample   cpp/kiss_icp/core/VoxelUtils.cpp:7-21,  PointToVoxel VoxelUtils.hpp:33-37
  Preprocess        cpp/kiss_icp/core/Preprocessing.cpp:55-95
  AdaptiveThreshold cpp/kiss_icp/core/Threshold.cpp:30-49, Threshold.hpp:38
  KissICP           cpp/kiss_icp/pipeline/KissICP.cpp:35-75
"""
import numpy as np
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation

# VoxelHashMap.cpp:35-41 -- centre, 6 faces, 12 edges, 8 corners, in this order
VOXEL_SHIFTS = [
    (0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1),
    (1, 1, 0), (1, -1, 0), (-1, 1, 0), (-1, -1, 0), (1, 0, 1), (1, 0, -1), (-1, 0, 1),
    (-1, 0, -1), (0, 1, 1), (0, 1, -1), (0, -1, 1), (0, -1, -1), (1, 1, 1), (1, 1, -1),
    (1, -1, 1), (1, -1, -1), (-1, 1, 1), (-1, 1, -1), (-1, -1, 1), (-1, -1, -1),
]


def point_to_voxel(p, voxel_size):
    """VoxelUtils.hpp:33-37: int(floor(p / voxel_size)) per axis"""
    return tuple(int(np.floor(c / voxel_size)) for c in p)


def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(a):
    """Sophus::SE3d::exp, tangent = (translation part, rotation part) -- via the matrix exponential
    of the 4x4 twist (independent of the closed form the oracle uses)"""
    X = np.zeros((4, 4))
    X[:3, :3] = hat(a[3:])
    X[:3, 3] = a[:3]
    return expm(X)


def se3_log(T):
    X = np.real(logm(T))
    return np.array([X[0, 3], X[1, 3], X[2, 3], X[2, 1], X[0, 2], X[1, 0]])


def voxel_hash(v):
    """std::hash<Voxel> (VoxelUtils.hpp:46-50): u32 wrap-around products, xor-ed"""
    M = 0xFFFFFFFF
    return (((v[0] & M) * 73856093) & M) ^ (((v[1] & M) * 19349669) & M) ^ (((v[2] & M) * 83492791) & M)


def robin_bucket_order(hashes, n_reserved):
    """Ids 0..len(hashes)-1, inserted in that order into a tsl::robin_map (v1.4.0, default parameters) on which
    reserve(n_reserved) was called first, listed in ITERATION (= bucket) order.  A second, independent statement of the
    container's published rules (the first is oracle/ref_build/shim/tsl/robin_map.h):
      reserve(n): bucket_count = smallest power of two >= ceil(float32(n) / 0.5);
      insert: walk from hash & (B - 1) while the walker's distance from its home is <= the occupant's (empty = -1);
      there the walker takes the bucket if it is empty, else swaps with the occupant -- which walks on by the same rule
      (so an occupant EQUALLY far from home stays: elements of one home bucket behind an insertion point rotate)."""
    if n_reserved == 0:
        return []
    want = int(np.ceil(np.float32(n_reserved) / np.float32(0.5)))
    B = 1
    while B < want:
        B *= 2
    dist = [-1] * B
    elem = [-1] * B
    for e, h in enumerate(hashes):
        ib, d = h & (B - 1), 0
        while d <= dist[ib]:
            ib, d = (ib + 1) & (B - 1), d + 1
        ce, cd = e, d
        while True:
            if cd > dist[ib]:
                if dist[ib] < 0:
                    dist[ib], elem[ib] = cd, ce
                    break
                (dist[ib], elem[ib]), (cd, ce) = (cd, ce), (dist[ib], elem[ib])
            ib, cd = (ib + 1) & (B - 1), cd + 1
    return [e for e in elem if e >= 0]


def voxel_downsample(frame, voxel_size, order="reference"):
    """VoxelUtils.cpp:7-21: keep the first point of every voxel; the survivors come out in the iteration order of the
    tsl::robin_map they were collected in (order="reference", what the reference does) or by ascending original index
    (order="index", the definition of rounds 1-2 of this repository)."""
    frame = np.asarray(frame, dtype=np.float64).reshape(-1, 3)
    seen = set()
    keep = []
    voxels = []
    for i, p in enumerate(frame):
        v = point_to_voxel(p, voxel_size)
        if v not in seen:
            seen.add(v)
            keep.append(i)
            voxels.append(v)
    if order == "reference":
        keep = [keep[e] for e in robin_bucket_order([voxel_hash(v) for v in voxels], len(frame))]
    return frame[keep]


class VoxelHashMap:
    def __init__(self, voxel_size, max_distance, max_points_per_voxel):
        self.voxel_size = voxel_size
        self.max_distance = max_distance
        self.max_points = max_points_per_voxel
        self.map = {}

    def empty(self):
        return not self.map

    def add_points(self, points):
        """VoxelHashMap.cpp:97-119"""
        res = np.sqrt(self.voxel_size * self.voxel_size / self.max_points)
        for p in np.asarray(points, dtype=np.float64).reshape(-1, 3):
            v = point_to_voxel(p, self.voxel_size)
            if v in self.map:
                pts = self.map[v]
                if len(pts) == self.max_points or any(np.linalg.norm(q - p) < res for q in pts):
                    continue
                pts.append(p.copy())
            else:
                self.map[v] = [p.copy()]

    def remove_far(self, origin):
        """VoxelHashMap.cpp:121-132: a voxel dies iff its FIRST point is >= max_distance away"""
        md2 = self.max_distance * self.max_distance
        for v in [v for v, pts in self.map.items() if np.sum((pts[0] - origin) ** 2) >= md2]:
            del self.map[v]

    def update(self, points, pose):
        """VoxelHashMap.cpp:89-95"""
        pts = np.asarray(points, dtype=np.float64).reshape(-1, 3) @ pose[:3, :3].T + pose[:3, 3]
        self.add_points(pts)
        self.remove_far(pose[:3, 3])

    def point_cloud(self):
        return np.array([p for pts in self.map.values() for p in pts]).reshape(-1, 3)

    def closest_neighbor(self, q):
        """VoxelHashMap.cpp:46-70: strict '<' over the 27 voxels in shift order; inside a voxel
        std::min_element's first minimum"""
        v = point_to_voxel(q, self.voxel_size)
        best, best_d = np.zeros(3), np.finfo(np.float64).max
        for s in VOXEL_SHIFTS:
            pts = self.map.get((v[0] + s[0], v[1] + s[1], v[2] + s[2]))
            if pts is None:
                continue
            d = [np.linalg.norm(p - q) for p in pts]
            k = int(np.argmin(d))  # first minimum
            if d[k] < best_d:
                best, best_d = pts[k], d[k]
        return best, best_d


def align_points_to_map(frame, voxel_map, initial_guess, max_dist, kernel_scale, max_iters=500, conv=1e-4):
    """Registration.cpp:138-167 with DataAssociation (:60-78) and BuildLinearSystem (:80-121)
    written out with explicit 3x6 Jacobians (no closed-form shortcuts)"""
    if voxel_map.empty():
        return initial_guess.copy(), 0
    src = np.asarray(frame, dtype=np.float64).reshape(-1, 3) @ initial_guess[:3, :3].T + initial_guess[:3, 3]
    T_icp = np.eye(4)
    iters = 0
    for _ in range(max_iters):
        JTJ = np.zeros((6, 6))
        JTr = np.zeros(6)
        for s in src:
            nn, d = voxel_map.closest_neighbor(s)
            if not d < max_dist:
                continue
            r = s - nn
            J = np.hstack([np.eye(3), -hat(s)])
            w = kernel_scale**2 / (kernel_scale + r @ r) ** 2
            JTJ += J.T @ (w * J)
            JTr += J.T @ (w * r)
        if np.all(JTJ == 0.0):
            dx = np.zeros(6)  # Eigen's LDLT solve returns 0 for zero pivots
        else:
            dx = np.linalg.lstsq(JTJ, -JTr, rcond=1e-13)[0] if np.linalg.matrix_rank(JTJ) < 6 else np.linalg.solve(JTJ, -JTr)
        est = se3_exp(dx)
        src = src @ est[:3, :3].T + est[:3, 3]
        T_icp = est @ T_icp
        iters += 1
        if np.linalg.norm(dx) < conv:
            break
    return T_icp @ initial_guess, iters


def preprocess(frame, timestamps, relative_motion, max_range, min_range, deskew):
    """Preprocessing.cpp:55-95"""
    frame = np.asarray(frame, dtype=np.float64).reshape(-1, 3)
    ts = np.asarray(timestamps, dtype=np.float64).ravel()
    if deskew and len(ts):
        mn, mx = ts.min(), ts.max()
        omega = se3_log(relative_motion)
        out = np.empty_like(frame)
        for i, p in enumerate(frame):
            stamp = (ts[i] - mn) / (mx - mn)
            T = se3_exp((stamp - 1.0) * omega)
            out[i] = T[:3, :3] @ p + T[:3, 3]
        frame = out
    rng = np.linalg.norm(frame, axis=1)
    return frame[(rng < max_range) & (rng > min_range)]


class AdaptiveThreshold:
    """Threshold.cpp:30-49, Threshold.hpp:38"""

    def __init__(self, initial_threshold, min_motion_th, max_range):
        self.min_motion = min_motion_th
        self.max_range = max_range
        self.sse = initial_threshold * initial_threshold
        self.n = 1

    def compute(self):
        return np.sqrt(self.sse / self.n)

    def update(self, dev):
        theta = Rotation.from_matrix(dev[:3, :3]).magnitude()
        err = np.linalg.norm(dev[:3, 3]) + 2.0 * self.max_range * np.sin(theta / 2.0)
        if err > self.min_motion:
            self.sse += err * err
            self.n += 1


class KissICP:
    """pipeline/KissICP.cpp:35-75 with KISSConfig defaults (KissICP.hpp:36-54)"""

    def __init__(self, voxel_size=1.0, max_range=100.0, min_range=0.0, max_points_per_voxel=20, min_motion_th=0.1,
                 initial_threshold=2.0, max_num_iterations=500, convergence_criterion=1e-4, deskew=True, downsample_order="reference"):
        self.voxel_size, self.max_range, self.min_range, self.deskew = voxel_size, max_range, min_range, deskew
        self.order = downsample_order
        self.max_iters, self.conv = max_num_iterations, convergence_criterion
        self.map = VoxelHashMap(voxel_size, max_range, max_points_per_voxel)
        self.threshold = AdaptiveThreshold(initial_threshold, min_motion_th, max_range)
        self.last_pose = np.eye(4)
        self.last_delta = np.eye(4)

    def register_frame(self, frame, timestamps=()):
        pre = preprocess(frame, timestamps, self.last_delta, self.max_range, self.min_range, self.deskew)
        fd = voxel_downsample(pre, self.voxel_size * 0.5, self.order)
        source = voxel_downsample(fd, self.voxel_size * 1.5, self.order)
        sigma = self.threshold.compute()
        guess = self.last_pose @ self.last_delta
        new_pose, self.iterations = align_points_to_map(source, self.map, guess, 3.0 * sigma, sigma, self.max_iters, self.conv)
        self.threshold.update(np.linalg.inv(guess) @ new_pose)
        self.map.update(fd, new_pose)
        self.last_delta = np.linalg.inv(self.last_pose) @ new_pose
        self.last_pose = new_pose
        return pre, source
