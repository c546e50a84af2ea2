"""CPU checks of the two inequalities the thread-per-query form of k_icp rests on (kiss-icp_amd/csrc/kicp_icp_wide.hpp), restated
in numpy float64 with the kernel's own order of operations (the device code is compiled -ffp-contract=off: no fused
multiply-add, so numpy's arithmetic IS the kernel's):

* wide_gaps / wide_keep_mask -- a voxel is skipped when the lower bound of the distance from the query to any point STORED in
  it is strictly above what is in hand.  The bound must hold for the distance AS THE SEARCH COMPUTES IT, roundings included, for
  every point the reference's PointToVoxel (VoxelUtils.hpp:33-37: floor of an IEEE division) assigns to that voxel -- also for
  points and queries that sit on voxel faces to the last bit.
* the stability test (WideQuery::Lr) -- a query that stays in its voxel keeps its neighbour without a search while the
  neighbour's new distance is strictly below the runner-up bound minus the accumulated motion.  Whenever the test says "keep",
  the reference's strict-'<' walk over all points (VoxelHashMap.cpp:55-63) must return that very point.

Neither needs a GPU: the GPU tests hold the kernel to the plain search bit for bit; these hold the ARGUMENT."""
import numpy as np
import pytest

DBL_MIN = np.finfo(np.float64).tiny
SHIFTS = [(i, j, k) for i in (0, 1, 2) for j in (0, 1, 2) for k in (0, 1, 2)]  # cell codes: 0 the layer below, 1 the query's, 2 above


def voxel_coord(x, vs):
    return np.floor(x / vs)  # PointToVoxel: an IEEE division, then floor


def wide_gaps(s, v, vs):
    f0, f1 = v * vs, (v + 1.0) * vs
    slack = (np.abs(f0) + np.abs(f1) + np.abs(s)) * 2.0 ** -48 + DBL_MIN
    gm = np.maximum((s - f0) - slack, 0.0)
    gp = np.maximum((f1 - s) - slack, 0.0)
    return gm * gm, gp * gp


def cell_bound(m2, p2, code):
    b = [m2[a] if code[a] == 0 else (p2[a] if code[a] == 2 else 0.0) for a in range(3)]
    return (b[0] + b[1]) + b[2]


def dist2(p, s):
    e = p - s
    return (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]  # the search's own expression


def _face_values(c, vs, rng, k):
    """k coordinates that PointToVoxel assigns to layer c, crowded at both faces of the layer (to the last bits) and a few inside"""
    lo, hi = c * vs, (c + 1.0) * vs
    cand = [lo, hi]
    for base in (lo, hi):
        x = base
        for _ in range(6):
            x = np.nextafter(x, np.inf)
            cand.append(x)
        x = base
        for _ in range(6):
            x = np.nextafter(x, -np.inf)
            cand.append(x)
    cand += list(rng.uniform(lo, hi, k))
    cand = np.array(cand)
    return cand[voxel_coord(cand, vs) == c]


@pytest.mark.parametrize("vs", [1.0, 0.5, 0.1, 0.3, 0.05, 2.5, 1.0 / 3.0])
def test_box_bounds_never_exceed_a_computed_distance(vs):
    rng = np.random.default_rng(int(vs * 1000))
    checked = 0
    for trial in range(110):
        scale = 10.0 ** rng.integers(-1, 4)  # coordinates from decimetres to kilometres, both signs
        centre = rng.uniform(-scale, scale, 3)
        v = voxel_coord(centre, vs)
        # the query: anywhere in its voxel, and on / next to its faces
        qs = [np.array([rng.choice(_face_values(v[a], vs, rng, 4)) for a in range(3)]) for _ in range(6)]
        for s in qs:
            assert np.array_equal(voxel_coord(s, vs), v)
            m2, p2 = wide_gaps(s, v, vs)
            for code in SHIFTS:
                bound = cell_bound(m2, p2, code)
                c = v + np.array(code) - 1.0
                axes = [_face_values(c[a], vs, rng, 3) for a in range(3)]
                for _ in range(8):
                    p = np.array([rng.choice(axes[a]) for a in range(3)])
                    assert bound <= dist2(p, s), (vs, s.tolist(), p.tolist(), code)
                    checked += 1
    assert checked > 100000


def test_the_bound_of_the_own_cell_is_zero_and_bounds_grow_outwards():
    rng = np.random.default_rng(1)
    for _ in range(200):
        vs = rng.choice([0.1, 0.5, 1.0])
        s = rng.uniform(-200, 200, 3)
        v = voxel_coord(s, vs)
        m2, p2 = wide_gaps(s, v, vs)
        assert cell_bound(m2, p2, (1, 1, 1)) == 0.0
        assert np.all(m2 >= 0) and np.all(p2 >= 0) and np.all(np.sqrt(m2) + np.sqrt(p2) <= vs * (1 + 1e-12))
        for code in SHIFTS:  # a corner cell is never bounded below a face cell it shares an axis with
            for a in range(3):
                face = tuple(code[b] if b == a else 1 for b in range(3))
                assert cell_bound(m2, p2, code) >= cell_bound(m2, p2, face)


def _reference_search(points, order, s):
    """VoxelHashMap.cpp:55-63: all points in the reference's visiting order, strict '<' keeps the first minimum"""
    best, arg = np.inf, -1
    for i in order:
        d = dist2(points[i], s)
        if d < best:
            best, arg = d, i
    return arg, best


def test_a_kept_neighbour_is_the_one_a_search_would_find():
    rng = np.random.default_rng(7)
    kept = searched = 0
    for trial in range(1500):
        vs = rng.choice([0.1, 0.5, 1.0])
        v = np.floor(rng.uniform(-300, 300, 3))
        s = (v + rng.uniform(0.05, 0.95, 3)) * vs
        if not np.array_equal(voxel_coord(s, vs), v):
            continue
        n = int(rng.integers(2, 40))
        pts = (v + rng.uniform(-1.0, 2.0, (n, 3))) * vs  # somewhere in the 27 cells
        if trial % 3 == 0:  # near-ties: a second point almost as close as the closest
            i0 = int(np.argmin(((pts - s) ** 2).sum(axis=1)))
            pts = np.vstack([pts, s + (pts[i0] - s) * (1.0 + rng.choice([1e-15, 1e-12, 1e-9, 1e-6, 1e-3])) * rng.choice([1.0, -1.0])])
        order = rng.permutation(len(pts))
        nn, best = _reference_search(pts, order, s)
        second = min(dist2(pts[i], s) for i in range(len(pts)) if i != nn)
        Lr = np.sqrt(second) * (1.0 - 2.0 ** -30)  # wide_finish: every cell was read
        for step in range(12):
            delta = rng.normal(0.0, 1.0, 3) * vs * 10.0 ** rng.uniform(-6, -1.3)
            s_new = s + delta
            if not np.array_equal(voxel_coord(s_new, vs), v):
                break  # another voxel: the kernel searches again, nothing to check
            m = s_new - s
            moved = np.sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]) * (1.0 + 2.0 ** -30) + DBL_MIN
            s = s_new
            Lr -= moved
            dp = dist2(pts[nn], s)
            if np.sqrt(dp) * (1.0 + 2.0 ** -30) < Lr:  # the kernel keeps nn without a search
                got, d = _reference_search(pts, order, s)
                assert got == nn and d == dp, (trial, step)
                kept += 1
            else:  # a search renews everything
                nn, best = _reference_search(pts, order, s)
                second = min(dist2(pts[i], s) for i in range(len(pts)) if i != nn)
                Lr = np.sqrt(second) * (1.0 - 2.0 ** -30)
                searched += 1
    assert kept > 2000 and searched > 500, (kept, searched)


def test_group_form_keeps_a_neighbour_only_when_a_search_would_find_it():
    """the group form's test (IcpQueryMeta::L2, kicp_icp.hip phase A): no roots -- with a = |nn - s|^2, b = |s - ss|^2 (ss: where the
    last search was made), L2 = the search's second smallest squared distance shaved by 2^-20, R = (L2 - a) - b, the neighbour is
    kept iff R > 0 and 4 (1 + 2^-20) a b < R^2, i.e. sqrt(a) + sqrt(b) < sqrt(L2).  Whenever it says "keep", the reference's walk
    must return that very point with that very distance -- near-ties down to 1e-15 included; the norms of all other points must
    also be strictly larger than the neighbour's (the reference compares norms: no tie in norm can hide behind a kept neighbour)."""
    rng = np.random.default_rng(11)
    kept = searched = 0
    for trial in range(1500):
        vs = rng.choice([0.1, 0.5, 1.0])
        v = np.floor(rng.uniform(-300, 300, 3))
        s = (v + rng.uniform(0.05, 0.95, 3)) * vs
        if not np.array_equal(voxel_coord(s, vs), v):
            continue
        n = int(rng.integers(2, 40))
        pts = (v + rng.uniform(-1.0, 2.0, (n, 3))) * vs
        if trial % 3 == 0:
            i0 = int(np.argmin(((pts - s) ** 2).sum(axis=1)))
            pts = np.vstack([pts, s + (pts[i0] - s) * (1.0 + rng.choice([1e-15, 1e-12, 1e-9, 1e-6, 1e-3])) * rng.choice([1.0, -1.0])])
        order = rng.permutation(len(pts))

        def search(at):
            nn, _ = _reference_search(pts, order, at)
            second = min(dist2(pts[i], at) for i in range(len(pts)) if i != nn)
            return nn, second * (1.0 - 2.0 ** -20), at.copy()

        nn, L2, ss = search(s)
        for step in range(12):
            s = s + rng.normal(0.0, 1.0, 3) * vs * 10.0 ** rng.uniform(-6, -1.3)
            if not np.array_equal(voxel_coord(s, vs), v):
                break
            m = s - ss
            b2 = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]
            dp = dist2(pts[nn], s)
            R = (L2 - dp) - b2
            if R > 0.0 and (4.0 * (1.0 + 2.0 ** -20)) * (dp * b2) < R * R:
                got, d = _reference_search(pts, order, s)
                assert got == nn and d == dp, (trial, step)
                assert all(np.sqrt(dist2(pts[i], s)) > np.sqrt(dp) for i in range(len(pts)) if i != nn), (trial, step)
                kept += 1
            else:
                nn, L2, ss = search(s)
                searched += 1
    assert kept > 2000 and searched > 500, (kept, searched)


def test_flat_service_settles_an_item_like_the_references_walk():
    """wide_serve_flat: unsigned minimum of the distance's bit pattern, then the smallest index among the points at the minimum,
    the others' distances into the runner-up -- against the plain walk, with exact ties"""
    rng = np.random.default_rng(11)
    for _ in range(3000):
        n = int(rng.integers(1, 21))
        s = rng.uniform(-50, 50, 3)
        pts = s + rng.choice([0.25, 0.5, 0.75, 1.0], (n, 3)) * rng.choice([-1.0, 1.0], (n, 3))  # lattice offsets: many exact ties
        d = np.array([dist2(p, s) for p in pts])
        bits = d.view(np.uint64)
        best_bits = bits.min()  # pass 1
        at_min = np.flatnonzero(bits == best_bits)
        k = at_min.min()  # pass 2
        others = np.delete(d, k)
        runner = others.min() if len(others) else np.inf  # passes 2 + 3 (a tie with the winner counts)
        arg, best = _reference_search(pts, range(n), s)
        assert k == arg and d[k] == best
        assert runner >= best and (len(others) == 0 or runner == sorted(d)[1])


def test_prune_verdict_from_the_z_layer_is_the_plain_test():
    """k_map_prune (kicp_map.hip): RemovePointsFarFromLocation's test (dx^2 + dy^2) + dz^2 >= max_distance^2 on a voxel's first
    point, decided from x, y and the LAYER of voxels z lies in wherever the layer's nearest and farthest |dz| give the same
    answer -- restated here; whenever it decides without z, the plain test on the real z must say the same.  Points at
    max_distance to the last bit either side, z on layer faces to the last bit, steep and flat directions."""
    rng = np.random.default_rng(5)
    decided = undecided = 0
    for vs, md in ((1.0, 100.0), (0.1, 20.0), (0.5, 37.3), (0.3, 250.0), (2.5, 100.0)):
        n = 60000
        d = rng.normal(size=(n, 3))
        d[:, 2] *= rng.choice([0.01, 0.1, 1.0, 5.0], n)
        d /= np.linalg.norm(d, axis=1)[:, None]
        eps = rng.choice([0.0, 1e-16, -1e-16, 2e-16, -2e-16, 1e-15, -1e-15, 1e-12, -1e-12, 1e-8, -1e-8, 1e-4, -1e-4, 0.01, -0.01, 0.5, -0.5], n)
        origin = rng.uniform(-500.0, 500.0, 3) * rng.choice([0.0, 0.01, 1.0])
        p = origin + d * (md * (1.0 + eps))[:, None]
        # a third of the points: z moved onto a face of its layer, or a few ulps beside it
        on_face = rng.random(n) < 0.33
        face = (np.floor(p[:, 2] / vs) + rng.integers(0, 2, n)) * vs
        for _ in range(3):
            face = np.where(rng.random(n) < 0.5, np.nextafter(face, np.inf), np.nextafter(face, -np.inf))
        p[:, 2] = np.where(on_face, face, p[:, 2])
        vz = np.floor(p[:, 2] / vs)  # the layer PointToVoxel puts z in (the key's third component)
        dx, dy = p[:, 0] - origin[0], p[:, 1] - origin[1]
        a = dx * dx + dy * dy
        md2 = md * md
        f0, f1 = vz * vs, (vz + 1.0) * vs
        slack = (np.abs(f0) + np.abs(f1)) * 2.0 ** -48 + DBL_MIN
        lo, hi = (f0 - slack) - origin[2], (f1 + slack) - origin[2]
        d_far = np.maximum(np.abs(lo), np.abs(hi))
        d_near = np.where((lo <= 0.0) & (hi >= 0.0), 0.0, np.minimum(np.abs(lo), np.abs(hi)))
        keeps = a + d_far * d_far < md2
        dies = ~keeps & (a + d_near * d_near >= md2)
        dz = p[:, 2] - origin[2]
        plain = a + dz * dz >= md2
        assert np.all((lo <= dz) & (dz <= hi))  # the interval really contains fl(z - oz)
        assert not np.any(plain[keeps]) and np.all(plain[dies])
        decided += int(keeps.sum() + dies.sum())
        undecided += int((~keeps & ~dies).sum())
    assert decided > 50000 and undecided > 50000, (decided, undecided)  # (both roads taken: most of these points sit in the shell on purpose)


def test_flat_service_index_arithmetic():
    """wide_serve_flat's integer part, restated: prefix sums of the items' point counts, passes of whole items with at most
    2048 points found by descending powers of two, and each point's item found the same way (the last item that starts at
    or before the point; items without points are passed over).  Every point of every item is visited exactly once, by the
    (item, index) pair the kernel would compute."""
    rng = np.random.default_rng(3)
    per_pass = 4 * 512
    for trial in range(300):
        n = int(rng.integers(1, 487))
        cnt = rng.integers(0, 21, n) if trial % 3 else rng.choice([0, 1, 20, 63], n)
        start = np.concatenate([[0], np.cumsum(cnt)])
        seen = [np.zeros(c, dtype=int) for c in cnt]
        lo = 0
        passes = 0
        while lo < n:
            p_lo = start[lo]
            hi = lo + 1
            step = 256
            while step:
                c = hi + step
                if c <= n and start[c] - p_lo <= per_pass:
                    hi = c
                step >>= 1
            assert hi == n or start[hi + 1] - p_lo > per_pass  # as many whole items as fit
            assert start[hi] - p_lo <= per_pass or hi == lo + 1
            for r in range(start[hi] - p_lo):
                p = p_lo + r
                x = lo
                step = 256
                while step:
                    c = x + step
                    if c < hi and start[c] <= p:
                        x = c
                    step >>= 1
                idx = p - start[x]
                assert 0 <= idx < cnt[x], (trial, p, x)
                seen[x][idx] += 1
            lo = hi
            passes += 1
        assert all((s == 1).all() for s in seen)
        assert passes <= (start[-1] + per_pass - 1) // per_pass + n // 32 + 1
