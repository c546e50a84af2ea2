"""CPU tests that pin the ORACLE (oracle/kiss_oracle.c).

The reference ships no golden vectors for this path and cannot be built here (SURVEY.md 8c:
"parity unpinned"), so the oracle is anchored three ways:
  1. closed-form / independently computed known answers (scipy, numpy) for the third-party
     arithmetic it restates (Sophus SE3 exp/log, Eigen LDLT, PointToVoxel),
  2. a second, deliberately naive numpy/dict restatement of the same reference sources
     (tests/naive_ref.py) on small inputs,
  3. the committed fixtures under tests/golden/ (made by tests/golden/make_golden.py), so a later
     edit of the oracle cannot drift silently.
"""
import os

import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial.transform import Rotation

import naive_ref as N
from helpers import make_pose, pose_error, random_cloud, sort_rows

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    oracle.lib()
    return oracle


# ---- (1) PointToVoxel: floor(p / voxel_size) with an IEEE divide (VoxelUtils.hpp:33-37) ----------
@pytest.mark.parametrize(
    "p,v,want",
    [
        ((0.0, -0.0, 1e-300), 1.0, (0, 0, 0)),
        ((-1e-12, -1.0, -1.0000000001), 1.0, (-1, -1, -2)),
        ((0.5, 1.0, 1.5), 0.5, (1, 2, 3)),
        ((0.3, 0.6, 0.9), 0.1, (2, 5, 9)),  # 0.3/0.1 = 2.9999999999999996, 0.6/0.1 = 5.999999999999999
        ((-0.3, -0.6, 99.99999999999999), 0.1, (-3, -6, 999)),
        ((123.456, -654.321, 7.0), 1.5, (82, -437, 4)),
    ],
)
def test_point_to_voxel_boundaries(O, p, v, want):
    got = tuple(int(x) for x in O.point_to_voxel(p, v))
    assert got == want
    assert got == tuple(int(np.floor(c / v)) for c in p)


def test_point_to_voxel_random_matches_numpy(O):
    rng = np.random.default_rng(0)
    for v in (0.05, 0.5, 1.0, 1.5):
        pts = rng.uniform(-200, 200, size=(2000, 3))
        want = np.floor(pts / v).astype(np.int64)
        got = np.array([O.point_to_voxel(p, v) for p in pts])
        assert np.array_equal(got, want)


# ---- (2) SE3 exp / log (Sophus 1.24.6 se3.hpp / so3.hpp) vs scipy ----------------------------------
def _twist(a):
    X = np.zeros((4, 4))
    X[:3, :3] = N.hat(a[3:])
    X[:3, 3] = a[:3]
    return X


@pytest.mark.parametrize("scale", [1e-14, 1e-9, 1e-5, 1e-2, 0.5, 2.0, 3.1])
def test_se3_exp_matches_matrix_exponential(O, scale):
    rng = np.random.default_rng(int(-np.log10(scale) * 7) + 100)
    for _ in range(20):
        a = rng.normal(size=6)
        a[3:] *= scale / np.linalg.norm(a[3:])
        got = O.se3_exp(a)
        want = expm(_twist(a))
        # Sophus evaluates V = I + (1-cos t)/t^2 W + (t - sin t)/t^3 W^2 literally for t >= 1e-10, so
        # the translation carries the cancellation error of 1 - cos(t): min(t/2, 2.3e-16/t) * |upsilon|
        th = np.linalg.norm(a[3:])
        slack = min(0.5 * th, 2.3e-16 / th) * np.linalg.norm(a[:3]) * 2.0
        np.testing.assert_allclose(got[:3, :3], want[:3, :3], rtol=0, atol=5e-14)
        np.testing.assert_allclose(got[:3, 3], want[:3, 3], rtol=0, atol=5e-14 + slack)
        np.testing.assert_allclose(got[:3, :3], Rotation.from_rotvec(a[3:]).as_matrix(), rtol=0, atol=5e-15)


def test_se3_exp_zero_and_pure_translation(O):
    assert np.array_equal(O.se3_exp(np.zeros(6)), np.eye(4))
    T = O.se3_exp(np.array([1.0, -2.0, 3.0, 0, 0, 0]))
    assert np.array_equal(T, make_pose((1.0, -2.0, 3.0)))


@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1e-2, 1.0, 3.0])
def test_se3_log_exp_round_trip(O, scale):
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = rng.normal(size=6)
        a[3:] *= scale / np.linalg.norm(a[3:])
        th = np.linalg.norm(a[3:])
        slack = min(0.5 * th, 2.3e-16 / th) * np.linalg.norm(a[:3]) * 4.0  # see the exp test above
        np.testing.assert_allclose(O.se3_log(O.se3_exp(a)), a, rtol=0, atol=1e-12 * max(1.0, np.abs(a).max()) + slack)


def test_se3_group_ops(O):
    rng = np.random.default_rng(6)
    for _ in range(10):
        A = make_pose(rng.uniform(-50, 50, 3), rng.uniform(-3, 3, 3))
        B = make_pose(rng.uniform(-50, 50, 3), rng.uniform(-3, 3, 3))
        np.testing.assert_allclose(O.se3_mul(A, B), A @ B, rtol=0, atol=1e-12)
        np.testing.assert_allclose(O.se3_inverse(A), np.linalg.inv(A), rtol=0, atol=1e-12)
        M, st = O.se3_roundtrip(A)
        assert st == 0
        np.testing.assert_allclose(M, A, rtol=0, atol=1e-14)


def test_se3_from_matrix_rejects_non_rigid(O):
    # SOPHUS_ENSURE(isOrthogonal(R)) / det > 0 in Sophus::SO3(Matrix3) (pybind ctor call sites
    # python/kiss_icp/pybind/kiss_icp_pybind.cpp:68,84,99,117)
    bad = np.eye(4)
    bad[0, 0] = 1.001
    assert O.se3_roundtrip(bad)[1] == -1
    refl = np.diag([1.0, 1.0, -1.0, 1.0])
    assert O.se3_roundtrip(refl)[1] == -1


# ---- (3) Eigen::LDLT<Matrix6d>::solve ---------------------------------------------------------------
def test_ldlt6_matches_numpy_on_spd(O):
    rng = np.random.default_rng(7)
    for _ in range(50):
        J = rng.normal(size=(40, 6)) * rng.uniform(0.1, 30, size=6)
        A = J.T @ J
        b = rng.normal(size=6)
        np.testing.assert_allclose(O.ldlt6_solve(A, b), np.linalg.solve(A, b), rtol=1e-8, atol=0)


def test_ldlt6_zero_matrix_gives_zero(O):
    # empty correspondence set: JTJ = 0, JTr = 0 -> dx = 0 -> converged -> returns the guess
    assert np.array_equal(O.ldlt6_solve(np.zeros((6, 6)), np.zeros(6)), np.zeros(6))
    assert np.array_equal(O.ldlt6_solve(np.zeros((6, 6)), np.ones(6)), np.zeros(6))


def test_ldlt6_rank_deficient_zero_pivot_component(O):
    # a diagonal system with two zero pivots: those components come back as exactly 0
    A = np.diag([4.0, 0.0, 2.0, 0.0, 1.0, 8.0])
    b = np.array([8.0, 5.0, 2.0, -3.0, 1.0, 4.0])
    np.testing.assert_array_equal(O.ldlt6_solve(A, b), np.array([2.0, 0.0, 1.0, 0.0, 1.0, 0.5]))


# ---- (6) GetClosestNeighbor vs brute force over the 27-neighbourhood --------------------------------
def test_closest_neighbor_vs_bruteforce(O):
    rng = np.random.default_rng(8)
    m = O.VoxelHashMap(1.0, 100.0, 20)
    pts = random_cloud(rng, 6000, extent=12.0, z_extent=3.0)
    m.add_points(pts)
    cloud = m.point_cloud()
    vox = np.floor(cloud / 1.0).astype(np.int64)
    queries = random_cloud(rng, 300, extent=14.0, z_extent=4.0)
    for q in queries:
        nn, d = m.closest_neighbor(q)
        qv = np.floor(q).astype(np.int64)
        inside = np.all(np.abs(vox - qv) <= 1, axis=1)
        if not inside.any():
            assert d == np.finfo(np.float64).max and np.array_equal(nn, np.zeros(3))
            continue
        dist = np.linalg.norm(cloud[inside] - q, axis=1)
        assert np.isclose(d, dist.min(), rtol=0, atol=1e-12)
        assert np.isclose(np.linalg.norm(nn - q), d, rtol=0, atol=1e-12)


def test_closest_neighbor_is_not_a_radius_search(O):
    # a point 1.2 m away but two voxels over is invisible; one 1.6 m away in a diagonal voxel is found
    m = O.VoxelHashMap(1.0, 100.0, 20)
    m.add_points(np.array([[2.1, 0.5, 0.5]]))
    nn, d = m.closest_neighbor(np.array([0.9, 0.5, 0.5]))
    assert d == np.finfo(np.float64).max
    m.add_points(np.array([[1.9, 1.9, 0.5]]))
    nn, d = m.closest_neighbor(np.array([0.9, 0.5, 0.5]))
    assert np.allclose(nn, [1.9, 1.9, 0.5]) and np.isclose(d, np.hypot(1.0, 1.4))


def test_closest_neighbor_tie_break_follows_shift_order(O):
    # two points at exactly the same distance: the voxel that comes first in the reference's shift
    # table (VoxelHashMap.cpp:35-41: centre, +x, -x, +y, -y, ...) wins under strict '<'
    m = O.VoxelHashMap(1.0, 100.0, 20)
    q = np.array([0.5, 0.5, 0.5])
    m.add_points(np.array([[-0.25, 0.5, 0.5], [1.25, 0.5, 0.5]]))  # -x voxel inserted first, +x second
    nn, d = m.closest_neighbor(q)
    assert np.array_equal(nn, [1.25, 0.5, 0.5]) and d == 0.75  # +x precedes -x in the table
    m.add_points(np.array([[0.5, 0.5, 0.5 + 0.75 / 2]]))  # closer point in the centre voxel
    nn, d = m.closest_neighbor(q)
    assert np.array_equal(nn, [0.5, 0.5, 0.875])


# ---- (7) Geman-McClure weight and the linear system -------------------------------------------------
def test_linear_system_single_correspondence(O):
    m = O.VoxelHashMap(1.0, 100.0, 20)
    t = np.array([0.5, 0.5, 0.5])
    m.add_points(t[None])
    s = np.array([0.7, 0.4, 0.6])
    sigma = 0.5
    JTJ, JTr, nc = O.build_linear_system(s[None], m, 3 * sigma, sigma)
    r = s - t
    w = sigma**2 / (sigma + r @ r) ** 2  # note sigma + r^2, not sigma^2 + r^2 (Registration.cpp:96-98)
    J = np.hstack([np.eye(3), -N.hat(s)])
    assert nc == 1
    np.testing.assert_allclose(JTJ, J.T @ (w * J), rtol=0, atol=1e-15)
    np.testing.assert_allclose(JTr, J.T @ (w * r), rtol=0, atol=1e-15)


def test_linear_system_threshold_is_strict(O):
    m = O.VoxelHashMap(1.0, 100.0, 20)
    m.add_points(np.array([[0.25, 0.5, 0.5]]))
    s = np.array([[0.75, 0.5, 0.5]])  # distance exactly 0.5
    assert O.build_linear_system(s, m, 0.5, 1.0)[2] == 0  # distance < max is strict (Registration.cpp:72)
    assert O.build_linear_system(s, m, 0.5000001, 1.0)[2] == 1


# ---- (8) AddPoints / RemovePointsFarFromLocation ------------------------------------------------------
def test_add_points_cap_and_spacing(O):
    m = O.VoxelHashMap(1.0, 100.0, 3)
    res = np.sqrt(1.0 / 3.0)  # map_resolution = sqrt(v^2 / max_points) = 0.577
    pts = np.array([[0.1, 0.1, 0.1], [0.2, 0.1, 0.1],  # 2nd is closer than res to the 1st: dropped
                    [0.9, 0.1, 0.1], [0.1, 0.9, 0.1],  # kept, kept -> voxel full (3)
                    [0.9, 0.9, 0.9]])  # full voxel rejects everything
    m.add_points(pts)
    assert np.array_equal(sort_rows(m.point_cloud()), sort_rows(pts[[0, 2, 3]]))
    assert np.linalg.norm(pts[1] - pts[0]) < res
    m2 = O.VoxelHashMap(1.0, 100.0, 3)
    m2.add_points(pts[::-1])  # arrival order matters
    assert np.array_equal(sort_rows(m2.point_cloud()), sort_rows(pts[[4, 3, 2]]))


def test_remove_far_looks_at_first_point_only(O):
    m = O.VoxelHashMap(10.0, 5.0, 20)
    m.add_points(np.array([[1.0, 1.0, 1.0], [9.0, 9.0, 9.0]]))  # one voxel, first point near the origin
    m.add_points(np.array([[19.0, 1.0, 1.0], [11.0, 1.0, 1.0]]))  # one voxel, first point far
    m.remove_far_away_points(np.zeros(3))
    assert np.array_equal(sort_rows(m.point_cloud()), sort_rows(np.array([[1.0, 1.0, 1.0], [9.0, 9.0, 9.0]])))
    m.remove_far_away_points(np.array([1.0, 1.0, 6.0]))  # exactly max_distance away: '>=' removes
    assert m.empty()


# ---- (4) degenerate registrations ---------------------------------------------------------------------
def test_align_empty_map_returns_guess(O):
    reg = O.Registration(500, 1e-4)
    m = O.VoxelHashMap(1.0, 100.0, 20)
    guess = make_pose((1.0, 2.0, 3.0), (0.1, 0.2, 0.3))
    T = reg.align_points_to_map(random_cloud(np.random.default_rng(1), 100), m, guess, 3.0, 1.0)
    np.testing.assert_allclose(T, guess, rtol=0, atol=1e-15)
    assert reg.last_stats["iterations"] == 0


def test_align_no_correspondence_returns_guess(O):
    reg = O.Registration(500, 1e-4)
    m = O.VoxelHashMap(1.0, 100.0, 20)
    m.add_points(np.array([[500.0, 500.0, 500.0]]))
    guess = make_pose((1.0, 2.0, 3.0), (0.1, 0.2, 0.3))
    T = reg.align_points_to_map(random_cloud(np.random.default_rng(1), 100), m, guess, 3.0, 1.0)
    np.testing.assert_allclose(T, guess, rtol=0, atol=1e-15)
    assert reg.last_stats["iterations"] == 1 and reg.last_stats["converged"] == 1


# ---- (5) config 1: plane pair, known offset smaller than a voxel ---------------------------------------
def test_plane_pair_recovers_observable_dof(O):
    from kiss_icp_amd.datasets import plane_pair

    T_true = make_pose((0.30, 0.10, 0.0), (0.0, 0.0, np.deg2rad(1.0)))
    m = O.VoxelHashMap(1.0, 100.0, 20)
    m.add_points(plane_pair(seed=42, noise=0.0))
    # frame 1 = the same two surfaces re-sampled, expressed in a sensor frame displaced by T_true
    world = plane_pair(seed=43, noise=0.0)
    frame = (world - T_true[:3, 3]) @ T_true[:3, :3]
    reg = O.Registration(500, 1e-4)
    T = reg.align_points_to_map(frame, m, np.eye(4), 3 * 2.0, 2.0)
    assert reg.last_stats["converged"] == 1
    # the floor pins z/roll/pitch, the wall pins x/yaw; y slides along both planes (unobservable).
    # Point-to-point ICP against a differently sampled surface stops a few millimetres short (the
    # in-plane components of the residuals never vanish), so the observable part is checked to cm.
    moved = frame @ T[:3, :3].T + T[:3, 3]
    floor, wall = moved[:5000], moved[5000:]
    assert np.abs(floor[:, 2]).max() < 2e-2
    assert np.abs(wall[:, 0] - 10.0).max() < 3e-2
    D = np.linalg.inv(T_true) @ T
    assert abs(D[0, 3]) < 1e-2 and abs(D[2, 3]) < 1e-2
    assert abs(np.arctan2(D[1, 0], D[0, 0])) < 2e-3
    # before: 0.3 m / 1 degree off
    assert np.abs((frame[5000:] @ np.eye(3))[:, 0] - 10.0).max() > 0.25


# ---- (9) oracle vs the naive numpy restatement -------------------------------------------------------
def test_voxel_downsample_vs_naive(O):
    rng = np.random.default_rng(21)
    pts = random_cloud(rng, 4000, extent=10.0, z_extent=2.0)
    for v in (0.5, 1.5):
        assert np.array_equal(O.voxel_down_sample(pts, v), N.voxel_downsample(pts, v))
    assert O.voxel_down_sample(np.zeros((0, 3)), 0.5).shape == (0, 3)


def test_voxel_downsample_order_small_tables_vs_naive(O):
    """the reference emits VoxelDownsample's survivors in the BUCKET ORDER of the tsl::robin_map it collects them in
    (VoxelUtils.cpp:17-19).  Thousands of tiny clouds -- tables of 2..128 buckets, so home-bucket collisions, clusters
    that wrap around the end of the table and equal-home groups rotated by insertions in front of them are the rule --
    through the C oracle and the independent Python statement of the container's rules; and the index order option"""
    rng = np.random.default_rng(23)
    for trial in range(1500):
        n = int(rng.integers(1, 65))
        pts = np.round(rng.normal(0.0, float(rng.choice([2.0, 8.0, 40.0])), (n, 3)), 2)
        v = float(rng.choice([0.5, 1.0, 1.5]))
        a = O.voxel_down_sample(pts, v)
        assert np.array_equal(a, N.voxel_downsample(pts, v)), (trial, n, v)
    pts = random_cloud(rng, 3000, extent=10.0, z_extent=2.0)
    old = O.set_downsample_order(O.INDEX_ORDER)
    try:
        for v in (0.5, 1.5):
            a = O.voxel_down_sample(pts, v)
            assert np.array_equal(a, N.voxel_downsample(pts, v, order="index"))
            O.set_downsample_order(O.REFERENCE_ORDER)
            b = O.voxel_down_sample(pts, v)
            O.set_downsample_order(O.INDEX_ORDER)
            assert not np.array_equal(a, b) and np.array_equal(sort_rows(a), sort_rows(b))  # same survivors, other order
    finally:
        O.set_downsample_order(old)


def test_golden_downsample_both_orders(O):
    g = np.load(os.path.join(GOLDEN, "downsample.npz"))
    for v, key in ((0.5, "out_050"), (1.5, "out_150")):
        assert np.array_equal(O.voxel_down_sample(g["points"], v), g[key])
        old = O.set_downsample_order(O.INDEX_ORDER)
        try:
            assert np.array_equal(O.voxel_down_sample(g["points"], v), g[key + "_index"])
        finally:
            O.set_downsample_order(old)


def test_map_vs_naive(O):
    rng = np.random.default_rng(22)
    om, nm = O.VoxelHashMap(1.0, 20.0, 20), N.VoxelHashMap(1.0, 20.0, 20)
    for k in range(4):
        pts = random_cloud(rng, 1500, extent=15.0, z_extent=2.0)
        T = make_pose((5.0 * k, 1.0, 0.0), (0, 0, 0.1 * k))
        om.update(pts, T)
        nm.update(pts, T)
        assert om.num_voxels() == len(nm.map)
        np.testing.assert_allclose(sort_rows(om.point_cloud()), sort_rows(nm.point_cloud()), rtol=0, atol=1e-12)
    for q in random_cloud(rng, 100, extent=20.0, z_extent=3.0):
        on, od = om.closest_neighbor(q)
        nn, nd = nm.closest_neighbor(q)
        assert od == pytest.approx(nd, abs=1e-12)
        np.testing.assert_allclose(on, nn, rtol=0, atol=1e-12)


def test_preprocess_vs_naive(O):
    rng = np.random.default_rng(23)
    pts = random_cloud(rng, 600, extent=120.0, z_extent=5.0)
    ts = rng.uniform(0, 0.1, 600)
    motion = make_pose((1.0, 0.05, -0.01), (0.002, -0.001, 0.02))
    for deskew in (False, True):
        got = O.Preprocessor(100.0, 3.0, deskew).preprocess(pts, ts, motion)
        want = N.preprocess(pts, ts, motion, 100.0, 3.0, deskew)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    with pytest.raises(IndexError):
        O.Preprocessor(100.0, 0.0, True).preprocess(pts, ts[:10], motion)


def test_adaptive_threshold_vs_naive(O):
    rng = np.random.default_rng(24)
    ot, nt = O.AdaptiveThreshold(2.0, 0.1, 100.0), N.AdaptiveThreshold(2.0, 0.1, 100.0)
    assert ot.get_threshold() == 2.0
    for _ in range(30):
        dev = make_pose(rng.normal(scale=0.2, size=3), rng.normal(scale=0.003, size=3))
        ot.update_model_deviation(dev)
        nt.update(dev)
        assert ot.get_threshold() == pytest.approx(nt.compute(), rel=1e-12)
    small = make_pose((0.01, 0, 0))  # below min_motion_th: ignored
    before = ot.get_threshold()
    ot.update_model_deviation(small)
    assert ot.get_threshold() == before


def test_align_vs_naive_small(O):
    rng = np.random.default_rng(25)
    world = np.concatenate([
        np.stack([rng.uniform(-8, 8, 900), rng.uniform(-8, 8, 900), rng.normal(0, 0.01, 900)], axis=1),
        np.stack([np.full(500, 6.0) + rng.normal(0, 0.01, 500), rng.uniform(-8, 8, 500), rng.uniform(0, 4, 500)], axis=1),
        np.stack([rng.uniform(-8, 8, 500), np.full(500, -5.0) + rng.normal(0, 0.01, 500), rng.uniform(0, 4, 500)], axis=1),
    ])
    om, nm = O.VoxelHashMap(1.0, 100.0, 20), N.VoxelHashMap(1.0, 100.0, 20)
    om.add_points(world)
    nm.add_points(world)
    T_true = make_pose((0.25, -0.15, 0.05), (0.01, -0.02, 0.03))
    sel = rng.choice(len(world), 250, replace=False)
    frame = (world[sel] - T_true[:3, 3]) @ T_true[:3, :3]
    reg = O.Registration(500, 1e-4)
    To = reg.align_points_to_map(frame, om, np.eye(4), 3.0, 1.0)
    Tn, it = N.align_points_to_map(frame, nm, np.eye(4), 3.0, 1.0)
    assert reg.last_stats["iterations"] == it
    dt, dr = pose_error(To, Tn)
    assert dt < 1e-9 and dr < 1e-9
    dt, dr = pose_error(To, T_true)
    assert dt < 2e-2 and dr < 5e-3


def test_pipeline_vs_naive_sequence(O):
    """20 small synthetic frames through RegisterFrame: oracle vs naive, poses within 1e-9"""
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=3, n_frames=20, beams=16, azimuth_steps=180)
    ko, kn = O.KissICP(deskew=0), N.KissICP(deskew=False)
    for i in range(20):
        pts, ts = ds[i]
        ko.register_frame(pts, ts)
        kn.register_frame(pts, ts)
        assert ko.last_stats()["iterations"] == kn.iterations
        dt, dr = pose_error(ko.last_pose, kn.last_pose)
        assert dt < 1e-9 and dr < 1e-9, (i, dt, dr)
    assert np.linalg.norm(ko.last_pose[:3, 3]) > 10.0  # it actually drove somewhere


def test_pipeline_vs_naive_with_deskew(O):
    from kiss_icp_amd.datasets import mulran_like

    ds = mulran_like(seed=4, n_frames=8, beams=16, azimuth_steps=128)
    ko, kn = O.KissICP(deskew=1), N.KissICP(deskew=True)
    for i in range(8):
        pts, ts = ds[i]
        ko.register_frame(pts, ts)
        kn.register_frame(pts, ts)
        dt, dr = pose_error(ko.last_pose, kn.last_pose)
        assert dt < 1e-8 and dr < 1e-8, (i, dt, dr)


# ---- threading: OpenMP in the reference's TBB sites must not change the answer beyond rounding --------
def test_oracle_thread_count_invariance(O):
    rng = np.random.default_rng(26)
    m = O.VoxelHashMap(1.0, 100.0, 20)
    world = random_cloud(rng, 20000, extent=20.0, z_extent=1.0)
    m.add_points(world)
    frame = world[rng.choice(len(world), 3000, replace=False)] + np.array([0.1, -0.05, 0.02])
    T1 = O.Registration(500, 1e-4, 1).align_points_to_map(frame, m, np.eye(4), 3.0, 1.0)
    T4 = O.Registration(500, 1e-4, 4).align_points_to_map(frame, m, np.eye(4), 3.0, 1.0)
    dt, dr = pose_error(T1, T4)
    assert dt < 1e-10 and dr < 1e-10


# ---- (3rd anchor) committed golden fixtures ------------------------------------------------------------
def _golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: run python tests/golden/make_golden.py")
    return np.load(path)


def test_golden_downsample(O):
    g = _golden("downsample.npz")
    for v, key in ((0.5, "out_050"), (1.5, "out_150")):
        assert np.array_equal(O.voxel_down_sample(g["points"], v), g[key])


def test_golden_map_and_neighbors(O):
    g = _golden("map_nn.npz")
    m = O.VoxelHashMap(1.0, 30.0, 20)
    for k in range(int(g["n_updates"])):
        m.update(g[f"pts_{k}"], g[f"pose_{k}"])
    assert np.array_equal(sort_rows(m.point_cloud()), g["cloud_sorted"])
    nn = np.array([m.closest_neighbor(q)[0] for q in g["queries"]])
    dd = np.array([m.closest_neighbor(q)[1] for q in g["queries"]])
    assert np.array_equal(nn, g["nn"])
    assert np.array_equal(dd, g["dist"])


def test_golden_align(O):
    g = _golden("align.npz")
    m = O.VoxelHashMap(1.0, 100.0, 20)
    m.add_points(g["world"])
    reg = O.Registration(500, 1e-4, 1)
    T = reg.align_points_to_map(g["frame"], m, g["guess"], float(g["max_dist"]), float(g["kernel"]))
    assert reg.last_stats["iterations"] == int(g["iterations"])
    np.testing.assert_allclose(T, g["T"], rtol=0, atol=1e-12)


def test_golden_sequence(O):
    g = _golden("sequence.npz")
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=int(g["seed"]), n_frames=int(g["n_frames"]), beams=int(g["beams"]), azimuth_steps=int(g["azimuth_steps"]))
    k = O.KissICP(deskew=0, max_num_threads=1)
    for i in range(int(g["n_frames"])):
        pts, ts = ds[i]
        assert np.array_equal(pts, g[f"scan_{i}"])  # the generator itself is pinned
        k.register_frame(pts, ts)
        np.testing.assert_allclose(k.last_pose, g["poses"][i], rtol=0, atol=1e-11)
        assert k.last_stats()["iterations"] == int(g["iterations"][i])


def test_near_tie_construction_and_the_oracles_norm_comparison(O):
    """tests/norm_ties.py builds candidates whose squared distances differ by one unit in the last place while their norms
    are equal, the LATER one (in the reference's order) the smaller.  The reference compares norms with strict '<'
    (core/VoxelHashMap.cpp:58-63), so it keeps the earlier one: the oracle must (it uses sqrt like the reference), a
    brute-force restatement of the loops must, and an argmin over squared distances must NOT -- else the GPU test built on
    this scene would prove nothing."""
    from norm_ties import brute_reference_choice, d2, make_near_tie_scene

    pts, qs, want = make_near_tie_scene(90, seed=3)
    assert len(qs) == 90
    vox = {}
    for i, p in enumerate(pts):
        vox.setdefault(tuple(np.floor(p).astype(int)), []).append(i)
    m = O.VoxelHashMap(1.0, 1000.0, 20)
    m.add_points(pts)
    assert len(m.point_cloud()) == len(pts)
    differs = 0
    for q, w in zip(qs, want):
        nn, d = m.closest_neighbor(q)
        assert brute_reference_choice(pts, vox, q) == w
        assert np.array_equal(nn, pts[w]) and d == np.sqrt(d2(pts[w], q))
        cands = [i for k, v in vox.items() for i in v if max(abs(np.array(k) - np.floor(q))) <= 1]
        differs += min(cands, key=lambda i: (d2(pts[i], q), i)) != w
    assert differs == len(qs)
