"""Synthetic spinning-LiDAR sequences (no real datasets are available offline).

Stands in for the reference's dataloaders; the *layout* of what they return is kept:
  * KITTI-like   (python/kiss_icp/datasets/kitti.py:56-69): (N,3) float32 values widened to
    float64, timestamps = empty array  -> deskew is skipped (Preprocessing.cpp:59).
  * MulRan-like  (python/kiss_icp/datasets/mulran.py:46-58): per-point timestamps
    floor(arange(H*W)/H)/W (column-major sweep), scans are motion-distorted so that deskewing
    is meaningful.

Scene: ground plane z = 0 plus axis-aligned boxes (buildings, cars, poles, "vegetation" with
large range jitter) scattered along a curved road.  The sensor drives forward `step` metres
and yaws `yaw_deg` degrees per frame.  Everything is seeded and pure numpy, so the same
sequence is produced on any machine.
"""
import numpy as np


def _rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


class SyntheticLidar:
    def __init__(
        self,
        beams=64,
        azimuth_steps=2048,
        elev_deg=(2.0, -24.8),
        sensor_height=1.73,
        step=1.0,
        yaw_deg=0.5,
        n_frames=200,
        seed=0,
        range_noise=0.02,
        sensor_max_range=120.0,
        timestamps=False,
        motion_distortion=False,
        density=1.0,
        clutter=1,
        porosity=0.0,
        volumetric=False,
        terrain_grade=0.0,
    ):
        self.H, self.W = beams, azimuth_steps
        self.height = sensor_height
        self.step, self.yaw = step, np.deg2rad(yaw_deg)
        self.n_frames = n_frames
        self.seed = seed
        self.range_noise = range_noise
        self.sensor_max_range = sensor_max_range
        self.with_timestamps = timestamps
        self.motion_distortion = motion_distortion
        el = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], beams))
        az = -np.arange(azimuth_steps) * (2.0 * np.pi / azimuth_steps)  # clockwise sweep
        # column-major sweep: consecutive H points share one azimuth column
        azg, elg = np.meshgrid(az, el, indexing="ij")
        self.dirs = np.stack(
            [np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], axis=-1
        ).reshape(-1, 3)
        self.col = np.repeat(np.arange(azimuth_steps), beams)
        self.stamps = np.floor(np.arange(beams * azimuth_steps) / beams) / azimuth_steps
        # vegetation lets a fraction `porosity` of the rays through (foliage is not a wall): returns from
        # several depths along one direction, as in real scans; which rays pass is a fixed property of
        # (ray, box), so a tree looks the same from frame to frame
        self.porosity = float(porosity)
        # volumetric: a ray that enters foliage returns from a random depth along its chord through the
        # box (instead of the surface + 0.3 m jitter): crowns are filled volumes, as in real scans
        self.volumetric = bool(volumetric)
        # terrain_grade > 0: the road runs along a shallow valley, the ground rises by this grade with the
        # distance from the road axis (flat within ~8 m of it) -- the near-horizontal beams then reach ground
        # at 40-100 m instead of leaving the scene, as on real roads that are not on an infinite plane
        self.grade = float(terrain_grade)
        self.ray_u = np.random.default_rng(424243 + seed).random(beams * azimuth_steps)
        self._build_scene(density, int(clutter))

    # ---- trajectory ------------------------------------------------------------------
    def _path(self, s):
        """pose (R, t) of the vehicle after driving arc length s (metres)"""
        if abs(self.yaw) < 1e-12:
            return np.eye(3), np.array([s, 0.0, self.height])
        radius = self.step / self.yaw
        th = s / radius
        t = np.array([radius * np.sin(th), radius * (1.0 - np.cos(th)), self.height])
        return _rotz(th), t

    def gt_pose(self, k):
        R, t = self._path(k * self.step)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        return T

    # what the reference's dataloaders offer its OdometryPipeline (python/kiss_icp/pipeline.py:66-73,147-152)
    @property
    def gt_poses(self):
        """(n_frames, 4, 4) ground-truth sensor poses, relative to the first one"""
        T0_inv = np.linalg.inv(self.gt_pose(0))
        return np.array([T0_inv @ self.gt_pose(k) for k in range(self.n_frames)])

    @property
    def sequence_id(self):
        return "synthetic_%dx%d_seed%d" % (self.H, self.W, self.seed)

    def get_frames_timestamps(self):
        return 0.1 * np.arange(self.n_frames)  # a 10 Hz sensor

    # ---- terrain ---------------------------------------------------------------------
    def _lateral(self, x, y):
        """signed distance of (x, y) from the road axis (a line, or a circle of radius step / yaw)"""
        if abs(self.yaw) < 1e-12:
            return y
        rc = self.step / self.yaw
        return rc - np.sign(rc) * np.sqrt(x * x + (y - rc) ** 2)

    def ground(self, x, y):
        if self.grade <= 0.0:
            return np.zeros_like(np.asarray(x, dtype=np.float64))
        lat = self._lateral(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
        return self.grade * (np.sqrt(lat * lat + 64.0) - 8.0)

    def _ground_hit(self, origins, dirs, t_flat):
        """ray / terrain intersection by Newton iterations from the flat-plane hit (the terrain is smooth and
        convex across the valley); inf where a ray escapes"""
        t = np.where(np.isfinite(t_flat), t_flat, 150.0)
        t = np.minimum(t, 400.0)
        h = 0.05
        for _ in range(8):
            x, y = origins[:, 0] + t * dirs[:, 0], origins[:, 1] + t * dirs[:, 1]
            f = origins[:, 2] + t * dirs[:, 2] - self.ground(x, y)
            x2, y2 = x + h * dirs[:, 0], y + h * dirs[:, 1]
            fp = dirs[:, 2] - (self.ground(x2, y2) - self.ground(x, y)) / h
            step = np.where(fp < -1e-9, f / np.where(fp < -1e-9, fp, -1.0), -50.0 * np.sign(f))
            t = np.clip(t - step, 0.0, 1000.0)
        x, y = origins[:, 0] + t * dirs[:, 0], origins[:, 1] + t * dirs[:, 1]
        f = origins[:, 2] + t * dirs[:, 2] - self.ground(x, y)
        return np.where((np.abs(f) < 1e-3) & (t > 0.0) & (t < 999.0), t, np.inf)

    # ---- scene -----------------------------------------------------------------------
    def _build_scene(self, density, clutter=1):
        """Street scene along the road: building facades, parked cars, poles, tree crowns and
        bushes ("vegetation": boxes whose returns get large range jitter)."""
        rng = np.random.default_rng(1000003 * (self.seed + 1))
        total = self.step * self.n_frames
        lo, hi, veg = [], [], []

        def add(s, lateral, half, zc, is_veg):
            R, t = self._path(s)
            c = t + R @ np.array([0.0, lateral, 0.0])
            c[2] = zc + float(self.ground(c[0], c[1]))
            lo.append(c - half)
            hi.append(c + half)
            veg.append(is_veg)

        s = -80.0
        while s < total + 80.0:
            for side in (1.0, -1.0):
                if rng.random() < 0.5 * density:  # facade segment
                    front, depth, h = rng.uniform(4, 7), rng.uniform(3, 6), rng.uniform(2.5, 7)
                    add(s + rng.uniform(-2, 2), side * (rng.uniform(9, 16) + depth), np.array([front, depth, h]), h, False)
                if rng.random() < 0.8 * density:  # second row / back buildings
                    front, depth, h = rng.uniform(4, 8), rng.uniform(4, 8), rng.uniform(4, 10)
                    add(s + rng.uniform(-4, 4), side * rng.uniform(32, 55), np.array([front, depth, h]), h, False)
                for _ in range(3):
                    if rng.random() < 0.8 * density:  # far-field buildings (seen by the near-horizontal beams)
                        front, depth, h = rng.uniform(5, 12), rng.uniform(5, 12), rng.uniform(4, 12)
                        add(s + rng.uniform(-6, 6), side * rng.uniform(55, 100), np.array([front, depth, h]), h, False)
                for ds in (0.0, 6.0):
                    if rng.random() < 0.45 * density:  # parked car
                        add(s + ds + rng.uniform(-1, 1), side * rng.uniform(3.2, 4.5), np.array([2.1, 0.9, 0.75]), 0.75, False)
                if rng.random() < 0.5 * density:  # pole / trunk + crown
                    lat = side * rng.uniform(5.5, 8.0)
                    sp = s + rng.uniform(-5, 5)
                    add(sp, lat, np.array([0.15, 0.15, 3.0]), 3.0, False)
                    if rng.random() < 0.7:
                        r = rng.uniform(1.2, 2.5)
                        add(sp, lat, np.array([r, r, rng.uniform(1.0, 2.0)]), rng.uniform(4.0, 6.0), True)
                for _ in range(5):
                    if rng.random() < 0.6 * density:  # bush / hedge
                        add(s + rng.uniform(-6, 6), side * rng.uniform(6, 60), np.array([rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(0.4, 1.2)]), rng.uniform(0.4, 1.2), True)
                for _ in range(5 * (clutter - 1)):  # clutter > 1: a vegetated scene (real urban scans are ~30 % vegetation)
                    lat = side * rng.uniform(6, 95)
                    if rng.random() < 0.5:  # more bushes
                        add(s + rng.uniform(-6, 6), lat, np.array([rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(0.4, 1.2)]), rng.uniform(0.4, 1.2), True)
                    else:  # tree crowns
                        add(s + rng.uniform(-6, 6), lat, np.array([rng.uniform(1.0, 3.0), rng.uniform(1.0, 3.0), rng.uniform(1.5, 4.0)]), rng.uniform(2.5, 6.0), True)
            s += 12.0
        self.box_lo, self.box_hi = np.array(lo), np.array(hi)
        self.box_veg = np.array(veg)

    # ---- ray casting -----------------------------------------------------------------
    def _cast(self, origins, dirs, R0, t0):
        """nearest hit range along each ray + vegetation flag; inf where nothing is hit.
        Rays are stored column-major (H consecutive rays share an azimuth column), so each box
        is only tested against the contiguous slice(s) of columns it subtends from (R0, t0)."""
        H, W = self.H, self.W
        dz = dirs[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            best = np.where(dz < 0.0, -origins[:, 2] / dz, np.inf)
            inv = 1.0 / dirs
        if self.grade > 0.0:
            best = self._ground_hit(origins, dirs, best)
        veg = np.zeros(len(dirs), dtype=bool)
        self._chord = np.zeros(len(dirs))
        bc = 0.5 * (self.box_lo + self.box_hi)
        near = np.where(np.linalg.norm(bc[:, :2] - t0[:2], axis=1) < self.sensor_max_range + 15.0)[0]
        margin = 3 + (int(0.02 * W) if self.motion_distortion else 0)
        for b in near:
            lo, hi = self.box_lo[b], self.box_hi[b]
            corners = np.array([[lo[0], lo[1]], [lo[0], hi[1]], [hi[0], lo[1]], [hi[0], hi[1]]]) - t0[:2]
            if (lo[0] - 1.0 <= t0[0] <= hi[0] + 1.0) and (lo[1] - 1.0 <= t0[1] <= hi[1] + 1.0):
                spans = [(0, W)]
            else:
                loc = corners @ R0[:2, :2]  # into the sensor frame (R0^T applied to rows)
                ang = np.arctan2(loc[:, 1], loc[:, 0])
                rel = (ang - ang[0] + np.pi) % (2 * np.pi) - np.pi
                a_lo, a_hi = ang[0] + rel.min(), ang[0] + rel.max()
                # column index c <-> azimuth -c*2pi/W
                c_lo = int(np.floor(-a_hi * W / (2 * np.pi))) - margin
                n_col = int(np.ceil(-a_lo * W / (2 * np.pi))) + margin + 1 - c_lo
                if n_col >= W:
                    spans = [(0, W)]
                else:
                    c_lo %= W
                    spans = [(c_lo, min(c_lo + n_col, W))]
                    if c_lo + n_col > W:
                        spans.append((0, c_lo + n_col - W))
            for (c0, c1) in spans:
                sl = slice(c0 * H, c1 * H)
                o, iv = origins[sl], inv[sl]
                with np.errstate(invalid="ignore"):
                    t1 = (lo - o) * iv
                    t2 = (hi - o) * iv
                tn = np.minimum(t1, t2).max(axis=1)
                tf = np.maximum(t1, t2).min(axis=1)
                hit = (tn <= tf) & (tn > 0.0) & (tn < best[sl])
                if self.porosity > 0.0 and self.box_veg[b]:
                    hit &= ((self.ray_u[sl] + 0.6180339887 * b) % 1.0) >= self.porosity
                best[sl] = np.where(hit, tn, best[sl])
                veg[sl] = np.where(hit, self.box_veg[b], veg[sl])
                if self.volumetric and self.box_veg[b]:
                    self._chord[sl] = np.where(hit, tf - tn, self._chord[sl])
        return best, veg

    def scan(self, k):
        """(points (N,3) f64 in the sensor frame, timestamps (N,) or empty) for frame k"""
        rng = np.random.default_rng(7919 * (self.seed + 1) + k)
        if self.motion_distortion:
            # pose of the sensor while column c is fired: between frame k-1 (s=0) and k (s=1)
            frac = (np.arange(self.W) + 0.0) / self.W
            Rs = np.empty((self.W, 3, 3))
            ts = np.empty((self.W, 3))
            for c in range(self.W):
                Rs[c], ts[c] = self._path((k - 1.0 + frac[c]) * self.step)
            Rr = Rs[self.col]
            dirs_w = np.einsum("nij,nj->ni", Rr, self.dirs)
            origins = ts[self.col]
        else:
            R, t = self._path(k * self.step)
            dirs_w = self.dirs @ R.T
            origins = np.broadcast_to(t, dirs_w.shape)
        R0, t0 = self._path(k * self.step)
        rng_hit, veg = self._cast(np.ascontiguousarray(origins), dirs_w, R0, t0)
        ok = np.isfinite(rng_hit) & (rng_hit < self.sensor_max_range)
        noise = rng.normal(0.0, self.range_noise, size=len(rng_hit))
        if self.volumetric:
            noise = np.where(veg, rng.random(len(rng_hit)) * self._chord, noise)
        else:
            noise = np.where(veg, rng.normal(0.0, 0.3, size=len(rng_hit)), noise)
        r = rng_hit + noise
        ok &= r > 0.5
        pts = self.dirs[ok] * r[ok, None]  # sensor frame at firing time
        pts = pts.astype(np.float32).astype(np.float64)  # loaders read float32 files
        if self.with_timestamps:
            return pts, self.stamps[ok].copy()
        return pts, np.array([])

    def __len__(self):
        return self.n_frames

    def __getitem__(self, k):
        return self.scan(k)


def kitti_like(seed=0, n_frames=200, beams=64, azimuth_steps=2048, **kw):
    """BASELINE config 2: 64 x 2048 = 131 072 rays, no timestamps"""
    return SyntheticLidar(beams=beams, azimuth_steps=azimuth_steps, elev_deg=(2.0, -24.8), seed=seed,
                          n_frames=n_frames, timestamps=False, motion_distortion=False, **kw)


def kitti_like_vegetated(seed=0, n_frames=200, **kw):
    """BASELINE config 2 on the scene SURVEY.md section 8(d) asks for: the same 64 x 2048 rays over a street
    lined with foliage (porous, volumetric returns) -- the 1.5 v source cloud lands at 4-4.5 k points
    (the bare street of kitti_like gives 1.7 k), ~125 map points examined per query"""
    kw.setdefault("clutter", 32)
    kw.setdefault("porosity", 0.95)
    kw.setdefault("volumetric", True)
    return kitti_like(seed=seed, n_frames=n_frames, **kw)


def _scan_job(args):
    factory, kw, k = args
    return factory(**kw)[k]


def generate_scans(factory, kw, frames, processes=None):
    """[(points, timestamps)] for the given frame numbers, generated by a pool of processes (a scan of the
    vegetated scene takes ~2 s of numpy ray casting; every frame is seeded on its own, so the result does
    not depend on how the work is split)"""
    import multiprocessing as mp
    import os

    frames = list(frames)
    procs = processes or min(len(frames), max(1, (os.cpu_count() or 2) - 1), 48)
    if procs <= 1 or len(frames) <= 2:
        ds = factory(**kw)
        return [ds[k] for k in frames]
    with mp.get_context("fork").Pool(procs) as pool:
        return pool.map(_scan_job, [(factory, kw, k) for k in frames], chunksize=1)


def mulran_like(seed=1, n_frames=200, beams=64, azimuth_steps=1024, **kw):
    """BASELINE config 3: 64 x 1024 = 65 536 rays, timestamps + motion distortion"""
    return SyntheticLidar(beams=beams, azimuth_steps=azimuth_steps, elev_deg=(16.6, -16.6), seed=seed,
                          n_frames=n_frames, timestamps=True, motion_distortion=True, **kw)


def livox_like(seed=2, n_frames=50, beams=128, azimuth_steps=8192, **kw):
    """BASELINE config 5: 128 x 8192 = 1 048 576 rays (use voxel_size 0.1)"""
    return SyntheticLidar(beams=beams, azimuth_steps=azimuth_steps, elev_deg=(2.0, -24.8), seed=seed,
                          n_frames=n_frames, timestamps=False, motion_distortion=False, **kw)


def plane_pair(seed=42, n=5000, noise=0.01):
    """BASELINE config 1 building block: n points on the floor z=0, (x,y) in [-20,20]^2, and n on
    the wall x = 10, y in [-20,20], z in [0,8]; N(0, noise^2) added."""
    rng = np.random.default_rng(seed)
    floor = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), np.zeros(n)], axis=1)
    wall = np.stack([np.full(n, 10.0), rng.uniform(-20, 20, n), rng.uniform(0, 8, n)], axis=1)
    pts = np.concatenate([floor, wall]) + rng.normal(0.0, noise, size=(2 * n, 3))
    return pts
