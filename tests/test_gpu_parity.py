"""GPU parity tests: the HIP path, called through the C-ABI (ctypes), against the CPU oracle on the
same seeded inputs.  Index/selection work must agree exactly; poses within the tolerance of
BASELINE.json's north_star (1e-4 m / 1e-4 rad) -- we assert a much tighter 1e-7."""
import numpy as np
import pytest

from helpers import make_pose, pose_error, random_cloud, sort_rows

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4  # north_star tolerance
POSE_TOL_RAD = 1e-4
TIGHT = 1e-7  # what we actually hold


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    oracle.lib()
    return oracle


def test_device_is_gfx950(gpu):
    from kiss_icp_amd import _cabi

    assert "gfx950" in _cabi.device_name(0)


# ---- VoxelDownsample ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,voxel", [(0, 0.5), (1, 0.5), (1000, 0.5), (50000, 0.5), (50000, 1.5), (131072, 0.05)])
def test_voxel_downsample_bit_exact(gpu, O, n, voxel):
    from kiss_icp_amd.voxelization import voxel_down_sample

    rng = np.random.default_rng(n + 7)
    pts = random_cloud(rng, n)
    got = voxel_down_sample(pts, voxel)
    want = O.voxel_down_sample(pts, voxel)
    assert got.shape == want.shape
    assert np.array_equal(got, want)  # same points, in the reference's order (tsl::robin_map bucket order)


def test_voxel_downsample_order_small_tables(gpu, O):
    """the reference's output order on tables of 2..256 buckets -- collisions, clusters wrapping around the end of the
    table, equal-home groups rotated by insertions in front of them -- and a dense cloud whose clusters run long:
    k_ds_arrange against the oracle's statement of tsl::robin_map's rules; then the index-order option"""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.voxelization import voxel_down_sample

    rng = np.random.default_rng(23)
    for trial in range(300):
        n = int(rng.integers(1, 129))
        pts = np.round(rng.normal(0.0, float(rng.choice([2.0, 8.0, 40.0])), (n, 3)), 2)
        v = float(rng.choice([0.5, 1.0, 1.5]))
        assert np.array_equal(voxel_down_sample(pts, v), O.voxel_down_sample(pts, v)), (trial, n, v)
    # every voxel of a small block occupied: the reference's hash maps neighbours to neighbouring buckets -> long clusters
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(6), indexing="ij"), axis=-1).reshape(-1, 3) * 0.5 + 0.25
    g = g[rng.permutation(len(g))]
    assert np.array_equal(voxel_down_sample(g, 0.5), O.voxel_down_sample(g, 0.5))
    gold = _golden("downsample.npz")
    _cabi.set_option("downsample_order", 0)
    old = O.set_downsample_order(O.INDEX_ORDER)
    try:
        for v, key in ((0.5, "out_050"), (1.5, "out_150")):
            assert np.array_equal(voxel_down_sample(gold["points"], v), gold[key + "_index"])
        pts = random_cloud(rng, 20000)
        assert np.array_equal(voxel_down_sample(pts, 0.5), O.voxel_down_sample(pts, 0.5))
    finally:
        _cabi.set_option("downsample_order", 1)
        O.set_downsample_order(old)


def test_voxel_downsample_duplicates_and_boundaries(gpu, O):
    from kiss_icp_amd.voxelization import voxel_down_sample

    pts = np.array([[0.0, 0.0, 0.0], [-0.0, 0.0, 0.0], [-1e-12, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, 0.5, 0.5],
                    [1.0, 1.0, 1.0], [0.999999999, 0.0, 0.0], [-0.5, -0.5, -0.5], [2.0, -2.0, 2.0]])
    for v in (0.5, 1.0, 1.5):
        assert np.array_equal(voxel_down_sample(pts, v), O.voxel_down_sample(pts, v))


def test_voxel_out_of_range_is_loud(gpu):
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.voxelization import voxel_down_sample

    pts = np.array([[0.0, 0.0, 0.0], [3.0e6, 0.0, 0.0]])
    with pytest.raises(_cabi.KicpError) as e:
        voxel_down_sample(pts, 1.0)
    assert e.value.status == 5


# ---- Preprocess ----------------------------------------------------------------------------------
@pytest.mark.parametrize("deskew", [False, True])
def test_preprocess(gpu, O, deskew):
    from kiss_icp_amd.preprocess import Preprocessor

    rng = np.random.default_rng(3)
    n = 40000
    pts = random_cloud(rng, n, extent=130.0, z_extent=10.0)
    ts = rng.uniform(0.0, 0.1, n)
    motion = make_pose((0.9, 0.05, -0.01), (0.002, -0.001, 0.01))
    got = Preprocessor(100.0, 2.0, deskew, 0).preprocess(pts, ts, motion)
    want = O.Preprocessor(100.0, 2.0, deskew, 0).preprocess(pts, ts, motion)
    assert got.shape == want.shape
    if deskew:
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)  # sin/cos/atan2 differ in the last ulp
    else:
        assert np.array_equal(got, want)


def test_preprocess_empty_timestamps_skips_deskew(gpu, O):
    from kiss_icp_amd.preprocess import Preprocessor

    pts = random_cloud(np.random.default_rng(4), 5000, extent=150.0)
    got = Preprocessor(100.0, 0.0, True, 0).preprocess(pts, np.array([]), make_pose((1, 0, 0)))
    want = O.Preprocessor(100.0, 0.0, True, 0).preprocess(pts, np.array([]), make_pose((1, 0, 0)))
    assert np.array_equal(got, want)


def test_preprocess_short_timestamps_raises(gpu):
    from kiss_icp_amd.preprocess import Preprocessor

    pts = random_cloud(np.random.default_rng(5), 100)
    with pytest.raises(IndexError):
        Preprocessor(100.0, 0.0, True, 0).preprocess(pts, np.zeros(10), np.eye(4))


# ---- VoxelHashMap ----------------------------------------------------------------------------------
def _maps(O, voxel=1.0, max_dist=100.0, mp=20):
    from kiss_icp_amd.mapping import VoxelHashMap

    return VoxelHashMap(voxel, max_dist, mp), O.VoxelHashMap(voxel, max_dist, mp)


def test_map_add_points_same_content(gpu, O):
    rng = np.random.default_rng(11)
    g, o = _maps(O)
    assert g.empty() and o.empty()
    for k in range(4):
        pts = random_cloud(rng, 20000, extent=25.0, z_extent=3.0)
        g.add_points(pts)
        o.add_points(pts)
        assert g.num_voxels() == o.num_voxels()
        assert np.array_equal(sort_rows(g.point_cloud()), sort_rows(o.point_cloud()))
    assert not g.empty()


def test_map_full_voxels_and_spacing_rule(gpu, O):
    # many points in few voxels: exercises the 20-point cap and the sqrt(v^2/20) spacing rule,
    # whose outcome depends on the arrival order
    rng = np.random.default_rng(12)
    g, o = _maps(O)
    pts = rng.uniform(0.0, 3.0, size=(30000, 3))
    g.add_points(pts)
    o.add_points(pts)
    assert np.array_equal(sort_rows(g.point_cloud()), sort_rows(o.point_cloud()))
    g2, o2 = _maps(O, voxel=0.5, mp=3)
    g2.add_points(pts)
    o2.add_points(pts)
    assert np.array_equal(sort_rows(g2.point_cloud()), sort_rows(o2.point_cloud()))


def test_map_update_and_prune(gpu, O):
    rng = np.random.default_rng(13)
    g, o = _maps(O, max_dist=30.0)
    for k in range(6):
        pts = random_cloud(rng, 8000, extent=28.0, z_extent=2.0)
        T = make_pose((4.0 * k, 0.5 * k, 0.0), (0, 0, 0.05 * k))
        g.update(pts, T)
        o.update(pts, T)
        assert g.num_voxels() == o.num_voxels()
        gp, op = sort_rows(g.point_cloud()), sort_rows(o.point_cloud())
        assert gp.shape == op.shape
        np.testing.assert_allclose(gp, op, rtol=0, atol=1e-12)
    # origin overload + explicit prune
    g.update(pts, np.array([100.0, 0.0, 0.0]))
    o.update(pts, np.array([100.0, 0.0, 0.0]))
    assert g.num_voxels() == o.num_voxels()
    g.remove_far_away_points(np.array([1e4, 0, 0]))
    assert g.empty()
    g.add_points(pts)  # tombstoned table still works
    o.clear()
    o.add_points(pts)
    assert np.array_equal(sort_rows(g.point_cloud()), sort_rows(o.point_cloud()))
    g.clear()
    assert g.empty() and len(g.point_cloud()) == 0


@pytest.mark.parametrize("voxel,max_dist", [(1.0, 100.0), (0.1, 20.0), (0.5, 37.3)])
def test_prune_decides_the_shell_like_the_reference(gpu, O, voxel, max_dist):
    """RemovePointsFarFromLocation (VoxelHashMap.cpp:121-132): a voxel dies iff its FIRST point is at least max_distance from
    the origin, decided on squared norms.  The device settles most voxels from the first point's x, y and the voxel's
    z-layer alone (k_map_prune) and reads z only in the shell where that cannot decide: points at max_distance exactly, a
    few ulps / 1e-12 / ... either side, steep and flat directions, origins off the lattice -- the same voxels must die."""
    rng = np.random.default_rng(int(voxel * 100 + max_dist))
    g, o = _maps(O, voxel=voxel, max_dist=max_dist)
    n = 6000
    d = rng.normal(size=(n, 3))
    d[:, 2] *= rng.choice([0.02, 0.3, 1.0, 3.0], n)
    d /= np.linalg.norm(d, axis=1)[:, None]
    eps = rng.choice([0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3, 0.02, -0.02, 0.3, -0.3], n)
    origin = rng.uniform(-3.0, 3.0, 3)
    pts = origin + d * (max_dist * (1.0 + eps))[:, None]
    g.add_points(pts)
    o.add_points(pts)
    assert g.num_voxels() == o.num_voxels() > n // 2
    g.remove_far_away_points(origin)
    o.remove_far_away_points(origin)
    assert 0 < o.num_voxels() < n and g.num_voxels() == o.num_voxels()
    assert np.array_equal(sort_rows(g.point_cloud()), sort_rows(o.point_cloud()))
    shifted = origin + np.array([0.37 * voxel, -0.11 * voxel, 0.73 * voxel])  # the same shell seen from next door
    g.remove_far_away_points(shifted)
    o.remove_far_away_points(shifted)
    assert np.array_equal(sort_rows(g.point_cloud()), sort_rows(o.point_cloud()))


@pytest.mark.parametrize("voxel,max_dist,mp", [(1.0, 30.0, 20), (0.5, 12.0, 5), (1.0, 25.0, 40)])
def test_fused_map_update_removes_what_the_reference_removes(gpu, O, voxel, max_dist, mp):
    """local_map_.Update = AddPoints, then RemovePointsFarFromLocation (VoxelHashMap.cpp:83-132).  The fused update takes
    the verdicts beside AddPoints' first kernel and carries them out in its second ("map_fused_update", kicp_map.hip): a
    moving origin whose shell sweeps through voxels that (a) receive nothing, (b) receive points of the very frame that
    removes them -- the reference appends and then drops the whole voxel; creating it anew would be wrong --, (c) are
    created by that frame beyond max_distance (judged by their first point), and voxels re-created later in recycled
    blocks.  Every frame: the oracle's map, and the three-kernel form's, voxel for voxel and point for point."""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.mapping import VoxelHashMap

    rng = np.random.default_rng(int(10 * voxel + max_dist + mp))
    maps = {fused: VoxelHashMap(voxel, max_dist, mp) for fused in (1, 0)}

    def update_all(points, where):  # (the option is read by the call)
        try:
            for fused, m in maps.items():
                _cabi.set_option("map_fused_update", fused)
                m.update(points, where)
        finally:
            _cabi.set_option("map_fused_update", 1)

    o = O.VoxelHashMap(voxel, max_dist, mp)
    died_touched = created_dead = 0
    for k in range(14):
        centre = np.array([0.45 * max_dist * k, 0.1 * max_dist * np.sin(k), 0.0])
        # a disc larger than max_distance around the new origin (points beyond it create voxels that die at once), plus a
        # dense patch on the far side of the OLD origin's disc: voxels that exist, receive points now, and are out of range now
        n = 9000
        r = max_dist * 1.15 * np.sqrt(rng.uniform(0.0, 1.0, n))
        a = rng.uniform(0.0, 2.0 * np.pi, n)
        pts = np.stack([r * np.cos(a), r * np.sin(a), rng.normal(0.0, 0.6, n)], axis=1)
        back = np.array([-0.9 * max_dist, 0.0, 0.0]) + rng.normal(0.0, [0.12 * max_dist, 0.3 * max_dist, 0.5], (3000, 3))
        pts = np.concatenate([pts, back])[rng.permutation(n + 3000)]
        if k % 3 == 2:  # origin overload: world points, an origin
            world = pts + centre
            before = {tuple(v) for v in np.floor(o.point_cloud() / voxel).astype(np.int64)} if o.num_voxels() else set()
            update_all(world, centre)
            o.update(world, centre)
        else:
            T = make_pose(tuple(centre), (0.0, 0.0, 0.3 * k))
            before = {tuple(v) for v in np.floor(o.point_cloud() / voxel).astype(np.int64)} if o.num_voxels() else set()
            world = pts @ T[:3, :3].T + T[:3, 3]
            update_all(pts, T)
            o.update(pts, T)
        after = {tuple(v) for v in np.floor(o.point_cloud() / voxel).astype(np.int64)}
        touched = {tuple(v) for v in np.floor(world / voxel).astype(np.int64)}
        died_touched += len((before & touched) - after)
        created_dead += len((touched - before) - after)
        want = sort_rows(o.point_cloud())
        for fused, m in maps.items():
            assert m.num_voxels() == o.num_voxels(), (k, fused)
            got = sort_rows(m.point_cloud())
            assert got.shape == want.shape, (k, fused)
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, err_msg=str((k, fused)))
        assert np.array_equal(sort_rows(maps[1].point_cloud()), sort_rows(maps[0].point_cloud())), k
    assert died_touched > 50 and created_dead > 50, (died_touched, created_dead)  # the cases were there
    # the tables and the free ring are sound afterwards: a plain insert, an explicit prune
    more = random_cloud(rng, 5000, extent=0.5 * max_dist, z_extent=1.0) + centre
    for m in maps.values():
        m.add_points(more)
        m.remove_far_away_points(centre)
    o.add_points(more)
    o.remove_far_away_points(centre)
    for m in maps.values():
        np.testing.assert_allclose(sort_rows(m.point_cloud()), sort_rows(o.point_cloud()), rtol=0, atol=1e-12)


def test_map_copy_is_an_independent_map(gpu, O):
    """the reference's VoxelHashMap is a copyable value type (VoxelHashMap.hpp:38-57): kicp_map_clone / VoxelHashMap.copy()
    give a second device map with the same voxels and points, which neither follows nor disturbs the original -- also
    when the original is the local map a running pipeline owns"""
    import copy

    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.kiss_icp import KissICP

    rng = np.random.default_rng(14)
    g, o = _maps(O)
    pts = random_cloud(rng, 20000, extent=30.0, z_extent=3.0)
    g.add_points(pts)
    o.add_points(pts)
    c = g.copy()
    assert c.num_voxels() == g.num_voxels() == o.num_voxels()
    assert np.array_equal(sort_rows(c.point_cloud()), sort_rows(o.point_cloud()))
    q = random_cloud(rng, 500, extent=32.0, z_extent=4.0)
    (na, da), (nb, db) = g.closest_neighbor(q), c.closest_neighbor(q)
    assert np.array_equal(na, nb) and np.array_equal(da, db)
    more = random_cloud(rng, 5000, extent=30.0, z_extent=3.0) + np.array([80.0, 0.0, 0.0])
    c.add_points(more)  # the copy grows (tables rehash, pools grow) ...
    o2 = O.VoxelHashMap(1.0, 100.0, 20)
    o2.add_points(pts)
    o2.add_points(more)
    assert c.num_voxels() == o2.num_voxels() > g.num_voxels() == o.num_voxels()  # ... the original does not
    assert np.array_equal(sort_rows(c.point_cloud()), sort_rows(o2.point_cloud()))
    g.clear()
    assert g.empty() and not c.empty()
    d = copy.deepcopy(c)
    assert d.num_voxels() == c.num_voxels()
    # a snapshot of a pipeline's local map, taken between two frames
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=4, n_frames=3, beams=32, azimuth_steps=512)
    k = KissICP(load_config(deskew=False))
    k.register_frame(*ds[0])
    k.register_frame(*ds[1])
    snap = k.local_map.copy()
    n_then = snap.num_voxels()
    assert n_then == k.local_map.num_voxels()
    k.register_frame(*ds[2])
    assert snap.num_voxels() == n_then


def test_closest_neighbor_exact(gpu, O):
    rng = np.random.default_rng(14)
    g, o = _maps(O)
    pts = random_cloud(rng, 60000, extent=20.0, z_extent=3.0)
    g.add_points(pts)
    o.add_points(pts)
    q = random_cloud(rng, 3000, extent=24.0, z_extent=5.0)
    nn, dist = g.closest_neighbor(q)
    for i in range(len(q)):
        onn, od = o.closest_neighbor(q[i])
        assert np.array_equal(nn[i], onn), i
        assert dist[i] == od, i
    # empty neighbourhood convention: zero vector, DBL_MAX
    nn, dist = g.closest_neighbor(np.array([[500.0, 500.0, 500.0]]))
    assert np.array_equal(nn[0], np.zeros(3)) and dist[0] == np.finfo(np.float64).max


def test_closest_neighbor_compares_norms_like_the_reference(gpu, O):
    """The reference's strict '<' is on (p - q).norm() (core/VoxelHashMap.cpp:58-63): among candidates whose ROUNDED ROOTS are
    equal the earliest in (shift, index) order wins -- also when its squared distance is the larger one.  tests/norm_ties.py
    constructs exactly that (squared distances one unit in the last place apart, equal roots, the later candidate the
    smaller): in every cluster a comparison of squared distances would return the other point.  GetClosestNeighbor must
    return the reference's, bit for bit (the searches detect such near ties and settle them with the reference's own
    comparison: kicp_search.hpp, kNormTie)."""
    from norm_ties import make_near_tie_scene

    pts, qs, want = make_near_tie_scene(150, seed=1)
    g, o = _maps(O, max_dist=1000.0)
    g.add_points(pts)
    o.add_points(pts)
    assert np.array_equal(sort_rows(g.point_cloud()), sort_rows(pts))  # (nothing fell to the spacing rule: the construction stands)
    nn, dist = g.closest_neighbor(qs)
    for i in range(len(qs)):
        onn, od = o.closest_neighbor(qs[i])
        assert np.array_equal(onn, pts[want[i]]), i  # the construction: the oracle keeps the EARLY candidate
        assert np.array_equal(nn[i], onn), (i, nn[i], onn)
        assert dist[i] == od, i


@pytest.mark.parametrize("form", ["group", "group_one_workgroup", "thread_per_query"])
def test_registration_compares_norms_like_the_reference(gpu, O, form):
    """the same near ties inside AlignPointsToMap, in every form of the association: queries = source points under the
    identity guess.  The two candidates of a cluster lie in different directions, so the other choice flips residuals: one
    iteration's pose would be millimetres off.  Pose, correspondences and examined counts must be the oracle's."""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.registration import Registration
    from norm_ties import make_near_tie_scene

    pts, qs, want = make_near_tie_scene(150, seed=2)
    g, o = _maps(O, max_dist=1000.0)
    g.add_points(pts)
    o.add_points(pts)
    try:
        _cabi.set_option("icp_wide", 1 if form == "thread_per_query" else 0)
        _cabi.set_option("icp_blocks", 1 if form == "group_one_workgroup" else 0)  # one workgroup: no scan lists, a lane per cell
        for iters in (1, 3):
            rg, ro = Registration(iters, 1e-12), O.Registration(iters, 1e-12)
            Tg = rg.align_points_to_map(qs, g, np.eye(4), 3.0, 1.0)
            To = ro.align_points_to_map(qs, o, np.eye(4), 3.0, 1.0)
            dt, dr = pose_error(To, Tg)
            assert dt < 1e-11 and dr < 1e-11, (form, iters, dt, dr)
            assert rg.last_stats["n_corr_last"] == ro.last_stats["n_corr_last"]
            assert rg.last_stats["points_examined"] == ro.last_stats["points_examined"]
    finally:
        _cabi.set_option("icp_wide", -1)
        _cabi.set_option("icp_blocks", 0)


# ---- Registration ------------------------------------------------------------------------------------
def _scene(rng, n=12000):
    floor = np.stack([rng.uniform(-25, 25, n), rng.uniform(-25, 25, n), rng.normal(0, 0.01, n)], axis=1)
    wall1 = np.stack([np.full(n // 2, 12.0) + rng.normal(0, 0.01, n // 2), rng.uniform(-25, 25, n // 2), rng.uniform(0, 6, n // 2)], axis=1)
    wall2 = np.stack([rng.uniform(-25, 25, n // 2), np.full(n // 2, -9.0) + rng.normal(0, 0.01, n // 2), rng.uniform(0, 6, n // 2)], axis=1)
    return np.concatenate([floor, wall1, wall2])


@pytest.mark.parametrize("blocks", [0, 1, 7, 64, 256])
def test_align_points_to_map_matches_oracle(gpu, O, blocks):
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.registration import Registration

    _cabi.set_option("icp_blocks", blocks)
    try:
        rng = np.random.default_rng(21)
        g, o = _maps(O)
        world = _scene(rng)
        g.add_points(world)
        o.add_points(world)
        T_true = make_pose((0.35, -0.2, 0.05), (0.004, -0.003, 0.02))
        src_world = _scene(np.random.default_rng(22), 3000)
        src = (np.linalg.inv(T_true) @ np.c_[src_world, np.ones(len(src_world))].T).T[:, :3]
        guess = make_pose((0.1, 0.0, 0.0))
        reg_g, reg_o = Registration(500, 1e-4), O.Registration(500, 1e-4)
        Tg = reg_g.align_points_to_map(src, g, guess, 3.0, 1.0)
        To = reg_o.align_points_to_map(src, o, guess, 3.0, 1.0)
        dt, dr = pose_error(To, Tg)
        assert dt < TIGHT and dr < TIGHT, (dt, dr)
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD
        assert reg_g.last_stats["iterations"] == reg_o.last_stats["iterations"]
        assert reg_g.last_stats["n_corr_last"] == reg_o.last_stats["n_corr_last"]
        assert reg_g.last_stats["points_examined"] == reg_o.last_stats["points_examined"]
        # it actually registered: close to the true pose in the observable directions
        assert pose_error(T_true, Tg)[0] < 0.05
        # determinism: a second run is bitwise identical
        Tg2 = reg_g.align_points_to_map(src, g, guess, 3.0, 1.0)
        assert np.array_equal(Tg, Tg2)
    finally:
        _cabi.set_option("icp_blocks", 0)


def test_registration_with_the_references_own_solve(gpu, O):
    """Registration.cpp:156 solves every Gauss-Newton step with Eigen's pivoted 6 x 6 LDLT.  The library's default goes through the
    3 x 3 Schur complement where that is well conditioned -- a DELIBERATE departure (same step within rounding, a third of the
    dependent chain; include/kicp.h "icp_schur_solve") -- so this pins the LDLT path, option 0, against the oracle: same
    iterations, same correspondences, pose at rounding level; and holds the default to it."""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.registration import Registration

    rng = np.random.default_rng(21)
    g, o = _maps(O)
    world = _scene(rng)
    g.add_points(world)
    o.add_points(world)
    T_true = make_pose((0.35, -0.2, 0.05), (0.004, -0.003, 0.02))
    src_world = _scene(np.random.default_rng(22), 3000)
    src = (np.linalg.inv(T_true) @ np.c_[src_world, np.ones(len(src_world))].T).T[:, :3]
    guess = make_pose((0.1, 0.0, 0.0))
    reg_o = O.Registration(500, 1e-4)
    To = reg_o.align_points_to_map(src, o, guess, 3.0, 1.0)
    got = {}
    try:
        for flag in (0, 1):
            _cabi.set_option("icp_schur_solve", flag)
            reg_g = Registration(500, 1e-4)
            got[flag] = reg_g.align_points_to_map(src, g, guess, 3.0, 1.0)
            dt, dr = pose_error(To, got[flag])
            assert dt < TIGHT and dr < TIGHT, (flag, dt, dr)
            assert reg_g.last_stats["iterations"] == reg_o.last_stats["iterations"], flag
            assert reg_g.last_stats["n_corr_last"] == reg_o.last_stats["n_corr_last"], flag
    finally:
        _cabi.set_option("icp_schur_solve", 1)
    dt, dr = pose_error(got[0], got[1])
    assert dt < 1e-11 and dr < 1e-11, (dt, dr)  # the two solves differ by rounding only


@pytest.mark.parametrize("offset", [(0.125, 0.125, 0.125), (0.0625, 0.125, 0.125), (0.9375, 0.125, 0.0625)])
def test_align_exact_ties_follow_the_reference_order(gpu, O, offset):
    """lattice map, queries exactly between lattice points: several candidates at EXACTLY the same
    distance, in different voxels.  The reference keeps the first one in (shift-table, in-voxel)
    order (strict '<', VoxelHashMap.cpp:55-63); a different choice changes the residuals and the
    pose.  Offsets inside 1/8 voxel of a face exercise the widened (filtered) staged windows."""
    from kiss_icp_amd.registration import Registration

    g, o = _maps(O)
    ax = np.arange(-6.0, 6.0, 0.25)
    lattice = np.stack(np.meshgrid(ax, ax, np.arange(-1.0, 1.0, 0.25), indexing="ij"), axis=-1).reshape(-1, 3)
    lattice = lattice[np.random.default_rng(5).permutation(len(lattice))]  # which 20 of a voxel's 64 survive
    g.add_points(lattice)
    o.add_points(lattice)
    qa = np.arange(-4.0, 4.0, 0.5)
    src = np.stack(np.meshgrid(qa, qa, np.array([-0.5, 0.0]), indexing="ij"), axis=-1).reshape(-1, 3) + np.array(offset)
    for guess in (np.eye(4), make_pose((0.5, -0.25, 0.0))):  # both keep the queries on exact binary fractions
        for iters in (1, 4):
            rg, ro = Registration(iters, 1e-12), O.Registration(iters, 1e-12)
            Tg = rg.align_points_to_map(src, g, guess, 3.0, 1.0)
            To = ro.align_points_to_map(src, o, guess, 3.0, 1.0)
            dt, dr = pose_error(To, Tg)
            assert dt < 1e-10 and dr < 1e-10, (offset, iters, dt, dr)
            assert rg.last_stats["n_corr_last"] == ro.last_stats["n_corr_last"]
            assert rg.last_stats["points_examined"] == ro.last_stats["points_examined"]


def test_align_degenerate_cases(gpu, O):
    from kiss_icp_amd.registration import Registration

    g, o = _maps(O)
    reg = Registration(500, 1e-4)
    guess = make_pose((1.0, 2.0, 3.0), (0.1, 0.2, 0.3))
    src = random_cloud(np.random.default_rng(1), 500)
    # empty map -> initial guess (Registration.cpp:143)
    T = reg.align_points_to_map(src, g, guess, 3.0, 1.0)
    np.testing.assert_allclose(T, guess, atol=1e-15)
    assert reg.last_stats["iterations"] == 0
    # no correspondence within the threshold -> dx = 0 -> guess, one iteration
    g.add_points(np.array([[1000.0, 1000.0, 1000.0]]))
    o.add_points(np.array([[1000.0, 1000.0, 1000.0]]))
    T = reg.align_points_to_map(src, g, guess, 3.0, 1.0)
    To = O.Registration(500, 1e-4).align_points_to_map(src, o, guess, 3.0, 1.0)
    np.testing.assert_allclose(T, To, atol=1e-15)
    assert reg.last_stats["iterations"] == 1
    # empty source
    T = reg.align_points_to_map(np.zeros((0, 3)), g, guess, 3.0, 1.0)
    np.testing.assert_allclose(T, guess, atol=1e-15)
    # non-rigid guess -> loud error (SOPHUS_ENSURE in the reference)
    from kiss_icp_amd import _cabi

    with pytest.raises(_cabi.KicpError):
        reg.align_points_to_map(src, g, np.diag([2.0, 1, 1, 1]), 3.0, 1.0)


def test_max_iterations_respected(gpu, O):
    from kiss_icp_amd.registration import Registration

    rng = np.random.default_rng(31)
    g, o = _maps(O)
    world = _scene(rng, 6000)
    g.add_points(world)
    o.add_points(world)
    src = _scene(np.random.default_rng(32), 1500)
    guess = make_pose((0.8, 0.5, 0.0), (0, 0, 0.05))
    for iters in (1, 3):
        rg, ro = Registration(iters, 1e-9), O.Registration(iters, 1e-9)
        Tg = rg.align_points_to_map(src, g, guess, 3.0, 1.0)
        To = ro.align_points_to_map(src, o, guess, 3.0, 1.0)
        assert rg.last_stats["iterations"] == iters == ro.last_stats["iterations"]
        dt, dr = pose_error(To, Tg)
        assert dt < TIGHT and dr < TIGHT


# ---- pipeline (RegisterFrame) over synthetic LiDAR sequences -------------------------------------------
def _run_sequence(O, ds, n_frames, deskew, composed=False, **cfg):
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.kiss_icp import KissICP, KissICPComposed

    config = load_config(deskew=deskew, **cfg)
    kg = (KissICPComposed if composed else KissICP)(config)
    ko = O.KissICP(deskew=int(deskew), **{("voxel_size" if k == "voxel_size" else k): v for k, v in cfg.items()})
    worst = (0.0, 0.0)
    for i in range(n_frames):
        pts, ts = ds[i]
        fg, sg = kg.register_frame(pts, ts)
        fo, so = ko.register_frame(pts, ts)
        assert fg.shape == fo.shape and sg.shape == so.shape, i
        if deskew and len(ts):
            np.testing.assert_allclose(fg, fo, rtol=0, atol=1e-9)
            np.testing.assert_allclose(sg, so, rtol=0, atol=1e-9)
        else:
            assert np.array_equal(fg, fo) and np.array_equal(sg, so), i
        dt, dr = pose_error(ko.last_pose, kg.last_pose)
        worst = (max(worst[0], dt), max(worst[1], dr))
        assert dt < POSE_TOL_M and dr < POSE_TOL_RAD, (i, dt, dr)
    return kg, ko, worst


def test_pipeline_kitti_like_sequence(gpu, O):
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=0, n_frames=12, beams=32, azimuth_steps=512)
    kg, ko, worst = _run_sequence(O, ds, 12, deskew=False)
    assert worst[0] < TIGHT and worst[1] < TIGHT, worst
    assert kg.local_map.num_voxels() == ko.local_map.num_voxels()
    np.testing.assert_allclose(sort_rows(kg.local_map.point_cloud()), sort_rows(ko.local_map.point_cloud()), rtol=0, atol=1e-9)
    sg, so = kg.last_stats(), ko.last_stats()
    assert sg["icp"]["iterations"] == so["iterations"]
    assert abs(sg["sigma"] - so["sigma"]) < 1e-9
    np.testing.assert_allclose(kg.last_delta, ko.last_delta, atol=1e-9)


def test_pipeline_mulran_like_sequence_with_deskew(gpu, O):
    from kiss_icp_amd.datasets import mulran_like

    ds = mulran_like(seed=1, n_frames=10, beams=32, azimuth_steps=512)
    kg, ko, worst = _run_sequence(O, ds, 10, deskew=True)
    assert worst[0] < TIGHT and worst[1] < TIGHT, worst


def test_pipeline_composed_matches_too(gpu, O):
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=3, n_frames=6, beams=32, azimuth_steps=512)
    kg, ko, worst = _run_sequence(O, ds, 6, deskew=False, composed=True)
    assert worst[0] < 1e-6 and worst[1] < 1e-6, worst  # the Python composition round-trips poses through 4x4 matrices


def test_pipeline_full_size_scan(gpu, O):
    """BASELINE config 2 size: 64 x 2048 rays (~130k points) for a few frames"""
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=0, n_frames=5)
    kg, ko, worst = _run_sequence(O, ds, 5, deskew=False)
    assert worst[0] < TIGHT and worst[1] < TIGHT, worst
    assert kg.last_stats()["n_raw"] > 120000


def test_pipeline_async_device_frames_match_sync(gpu, O):
    """frames already in HBM, enqueued back-to-back without host synchronisation, give the same
    trajectory as the synchronous host-buffer path"""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like
    from kiss_icp_amd.kiss_icp import KissICP

    ds = kitti_like(seed=5, n_frames=8, beams=32, azimuth_steps=512)
    scans = [ds[i][0] for i in range(8)]
    ka, ks = KissICP(load_config(deskew=False)), KissICP(load_config(deskew=False))
    dev = [_cabi.DeviceArray(s) for s in scans]
    for d in dev:
        ka.register_frame_device(d.ptr, d.shape[0])
    ka.sync()
    poses_async = ka.synced_poses()
    for i, s in enumerate(scans):
        ks.register_frame(s)
        assert np.array_equal(ks.last_pose, poses_async[i]), i


@pytest.mark.parametrize("deskew", [False, True])
def test_pipeline_async_long_queue_two_streams(gpu, O, deskew):
    """more frames queued than the pipeline keeps in flight, with and without deskewing: the front
    stages of frame k+1 run on a second stream (under frame k's registration when no pose is needed,
    behind it when deskewing), buffers alternate by frame parity, the host throttles instead of
    synchronising -- and the trajectory is bit for bit the synchronous one"""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like, mulran_like
    from kiss_icp_amd.kiss_icp import KissICP

    n_frames = 14
    ds = (mulran_like if deskew else kitti_like)(seed=9, n_frames=n_frames, beams=32, azimuth_steps=512)
    scans = [ds[i] for i in range(n_frames)]
    ka, ks = KissICP(load_config(deskew=deskew)), KissICP(load_config(deskew=deskew))
    dev = [(_cabi.DeviceArray(p), _cabi.DeviceArray(t) if len(t) else None) for p, t in scans]
    for d, t in dev:
        ka.register_frame_device(d.ptr, d.shape[0], t.ptr if t is not None else None, t.shape[0] if t is not None else 0)
    ka.sync()
    poses_async = ka.synced_poses()
    assert len(poses_async) == n_frames
    for i, (p, t) in enumerate(scans):
        ks.register_frame(p, t)
        assert np.array_equal(ks.last_pose, poses_async[i]), i
    # and the clouds of the last frame come from the right parity buffers
    for which in (0, 1, 2):  # preprocessed frame, source, frame downsample
        assert np.array_equal(ka.output(which), ks.output(which)), which


def test_pipeline_empty_frame_in_a_drive(gpu, O):
    """an empty scan in the middle of a drive: no correspondence, dx = 0 after one iteration, the
    constant-velocity guess becomes the pose (Registration.cpp:156 with the zero-pivot rule); clouds,
    iteration count and pose follow the oracle.
    (A ONE-point scan is deliberately not asserted: its 6x6 system has rank 3, the remaining pivots
    are rounding noise of order 1e-16 |s|^2 rather than exact zeros, and the solution -- in the
    reference as much as here -- is whatever that noise divides out to.)"""
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.datasets import kitti_like
    from kiss_icp_amd.kiss_icp import KissICP

    ds = kitti_like(seed=3, n_frames=5, beams=16, azimuth_steps=256)
    scans = [ds[0][0], ds[1][0], np.zeros((0, 3))]
    kg, ko = KissICP(load_config(deskew=False)), O.KissICP(deskew=0)
    for i, pts in enumerate(scans):
        fg, sg = kg.register_frame(pts)
        fo, so = ko.register_frame(pts, np.array([]))
        assert fg.shape == fo.shape and sg.shape == so.shape, i
        assert np.array_equal(fg, fo) and np.array_equal(sg, so), i
        assert kg.last_stats()["icp"]["iterations"] == ko.last_stats()["iterations"], i
        dt, dr = pose_error(ko.last_pose, kg.last_pose)
        assert dt < TIGHT and dr < TIGHT, (i, dt, dr)
    assert kg.last_stats()["icp"]["iterations"] == 1 and len(kg.output(1)) == 0


# ---- committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py) ----------------------
def _golden(name):
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    return np.load(path)


def test_golden_downsample_gpu(gpu):
    from kiss_icp_amd.voxelization import voxel_down_sample

    g = _golden("downsample.npz")
    assert np.array_equal(voxel_down_sample(g["points"], 0.5), g["out_050"])
    assert np.array_equal(voxel_down_sample(g["points"], 1.5), g["out_150"])


def test_golden_map_and_neighbors_gpu(gpu):
    from kiss_icp_amd.mapping import VoxelHashMap

    g = _golden("map_nn.npz")
    m = VoxelHashMap(1.0, 30.0, 20)
    for k in range(int(g["n_updates"])):
        m.update(g[f"pts_{k}"], g[f"pose_{k}"])
    np.testing.assert_allclose(sort_rows(m.point_cloud()), g["cloud_sorted"], rtol=0, atol=1e-12)
    nn, dist = m.closest_neighbor(g["queries"])
    np.testing.assert_allclose(nn, g["nn"], rtol=0, atol=1e-12)
    found = g["dist"] < 1e300
    np.testing.assert_allclose(dist[found], g["dist"][found], rtol=0, atol=1e-12)
    assert np.array_equal(dist[~found], g["dist"][~found])


def test_golden_align_gpu(gpu):
    from kiss_icp_amd.mapping import VoxelHashMap
    from kiss_icp_amd.registration import Registration

    g = _golden("align.npz")
    m = VoxelHashMap(1.0, 100.0, 20)
    m.add_points(g["world"])
    reg = Registration(500, 1e-4)
    T = reg.align_points_to_map(g["frame"], m, g["guess"], float(g["max_dist"]), float(g["kernel"]))
    assert reg.last_stats["iterations"] == int(g["iterations"])
    dt, dr = pose_error(g["T"], T)
    assert dt < TIGHT and dr < TIGHT, (dt, dr)


def test_golden_sequence_gpu(gpu):
    from kiss_icp_amd.config import load_config
    from kiss_icp_amd.kiss_icp import KissICP

    g = _golden("sequence.npz")
    k = KissICP(load_config(deskew=False))
    for i in range(int(g["n_frames"])):
        k.register_frame(g[f"scan_{i}"])
        dt, dr = pose_error(g["poses"][i], k.last_pose)
        assert dt < TIGHT and dr < TIGHT, (i, dt, dr)
        assert k.last_stats()["icp"]["iterations"] == int(g["iterations"][i])


# ---- full-size, size-independent properties (no oracle needed) --------------------------------------------
def test_downsample_properties_at_full_size(gpu):
    """1M points: one survivor per voxel, each the FIRST input point of its voxel, a fixed point as a set; the output
    order is that of the reference's grid -- checked here through a property that needs no oracle (home buckets ascend
    in iteration order) -- and with the index-order option they come out exactly as keep-first says"""
    from kiss_icp_amd import _cabi
    from kiss_icp_amd.voxelization import voxel_down_sample

    rng = np.random.default_rng(77)
    pts = random_cloud(rng, 1_000_000, extent=80.0, z_extent=6.0)
    for v in (0.5, 1.5):
        out = voxel_down_sample(pts, v)
        vox = np.floor(out / v).astype(np.int64)
        assert len(np.unique(vox, axis=0)) == len(out)  # one point per voxel
        keys = np.floor(pts / v).astype(np.int64)
        _, first = np.unique(keys, axis=0, return_index=True)
        assert len(out) == len(first)  # every voxel kept
        assert np.array_equal(sort_rows(out), sort_rows(pts[first]))  # keep-first: the first input point of each voxel
        assert np.array_equal(sort_rows(voxel_down_sample(out, v)), sort_rows(out))  # a fixed point, as a set
        # bucket order: homes (reference hash & (2^21 - 1): 1M points reserve 2^21 buckets) ascend except at collisions
        u = vox.astype(np.int64) & 0xFFFFFFFF
        home = (((u[:, 0] * 73856093) & 0xFFFFFFFF) ^ ((u[:, 1] * 19349669) & 0xFFFFFFFF) ^ ((u[:, 2] * 83492791) & 0xFFFFFFFF)) & ((1 << 21) - 1)
        # robin-hood invariant: in bucket order the home buckets never descend (but once, where a cluster wraps around)
        assert (np.diff(home) < 0).sum() <= 1
        _cabi.set_option("downsample_order", 0)
        try:
            assert np.array_equal(voxel_down_sample(pts, v), pts[np.sort(first)])  # index order: a subsequence of the input
        finally:
            _cabi.set_option("downsample_order", 1)


def test_map_properties_at_full_size(gpu):
    """a 300k-point insert: per-voxel cap and spacing invariants, idempotent re-insert, NN of a stored
    point is itself at distance 0"""
    from kiss_icp_amd.mapping import VoxelHashMap

    rng = np.random.default_rng(78)
    pts = random_cloud(rng, 300_000, extent=60.0, z_extent=3.0)
    m = VoxelHashMap(1.0, 100.0, 20)
    m.add_points(pts)
    cloud = m.point_cloud()
    keys = np.floor(cloud).astype(np.int64)
    _, counts = np.unique(keys, axis=0, return_counts=True)
    assert counts.max() <= 20 and m.num_voxels() == len(counts)
    nv, npts = m.num_voxels(), len(cloud)
    m.add_points(cloud)  # every stored point is closer than map_resolution to itself: nothing changes
    assert (m.num_voxels(), len(m.point_cloud())) == (nv, npts)
    sel = cloud[rng.choice(len(cloud), 5000, replace=False)]
    nn, dist = m.closest_neighbor(sel)
    assert np.array_equal(nn, sel) and not dist.any()
    # spacing rule inside a voxel: pairwise distances >= sqrt(v^2 / max_points)
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    ck, cp = keys[order], cloud[order]
    same = np.all(ck[1:] == ck[:-1], axis=1)
    d = np.linalg.norm(cp[1:] - cp[:-1], axis=1)[same]
    assert d.min() >= np.sqrt(1.0 / 20.0)
