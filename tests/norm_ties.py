"""Constructed NEAR ties for the nearest-neighbour search: candidates whose SQUARED distances differ (by a unit or two in the
last place) while their norms -- sqrt, correctly rounded -- are equal.  The reference compares norms with strict '<' in
(shift, index) order (core/VoxelHashMap.cpp:58-63), so the EARLIER candidate wins although its squared distance is the
larger one; a search that compares squared distances picks the other.  Used by tests/test_gpu_parity.py (GPU vs oracle) and
tests/test_oracle.py (the construction itself, against the oracle and a brute-force restatement)."""
import numpy as np

# the reference's shift table (core/VoxelHashMap.cpp:35-41)
SHIFTS = np.array([
    [0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0],
    [1, 0, 1], [1, 0, -1], [-1, 0, 1], [-1, 0, -1], [0, 1, 1], [0, 1, -1], [0, -1, 1], [0, -1, -1], [1, 1, 1], [1, 1, -1], [1, -1, 1],
    [1, -1, -1], [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]])


def d2(p, q):
    e = np.asarray(p, dtype=np.float64) - np.asarray(q, dtype=np.float64)
    return (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]  # Eigen's 3-term redux order, no FMA


def _near_tie_partner(q, early, direction, side_axis):
    """a point `late` = q + t * direction + eps * e_side with d2(late) < d2(early), the two one unit in the last place apart,
    and EQUAL square roots -- or None when d2(early)'s predecessor has another root (the caller draws again)"""
    target = d2(early, q)
    below = np.nextafter(target, 0.0)
    if np.sqrt(below) != np.sqrt(target):
        return None
    t = np.sqrt(target)
    late = q + t * direction
    for _ in range(200):  # along the direction: just below `below` (coordinate steps are coarse far from the origin)
        if d2(late, q) < below:
            break
        t = np.nextafter(t, 0.0) - 1e-13
        late = q + t * direction
    else:
        return None
    lo, hi = 0.0, 1e-4  # ... then a small sideways component raises the squared distance in steps far below its last place
    side = np.zeros(3)
    side[side_axis] = 1.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        dl = d2(late + mid * side, q)
        if dl == below:
            return late + mid * side
        if dl < below:
            lo = mid
        else:
            hi = mid
    return None


def make_near_tie_scene(n_clusters=120, seed=0, fillers=6):
    """(map points in insertion order, queries, per query: the index of the map point the reference must pick).
    Clusters sit 5 voxels apart.  In each, a query q and two candidates at the same norm: `early` first in the reference's
    order (the same voxel and inserted first; or the smaller shift position), `late` with the SMALLER squared distance (one
    unit in the last place) -- the reference keeps `early`, a comparison of squared distances takes `late`.  A few fillers
    farther away make voxels and scan lists non-trivial.  Voxel size 1."""
    rng = np.random.default_rng(seed)
    pts, queries, want = [], [], []
    x, y, z = np.eye(3)
    tries = 0
    while len(queries) < n_clusters and tries < 40 * n_clusters:
        tries += 1
        c = len(queries)
        base = np.array([5.0 * (c % 10), 5.0 * ((c // 10) % 10), 5.0 * (c // 100)])
        kind = c % 3
        if kind == 0:  # both in the query's own voxel: insertion order decides
            q = base + 0.5 + rng.uniform(-0.05, 0.05, 3)
            r = rng.uniform(0.30, 0.40)
            early, ldir, side = q + r * x, y, 2
        elif kind == 1:  # cell (+1, 0, 0) is shift 1, cell (-1, 0, 0) shift 2
            q = base + 0.5 + rng.uniform(-0.05, 0.05, 3)
            r = rng.uniform(0.62, 0.80)
            early, ldir, side = q + r * x, -x, 1
        else:  # the own cell (shift 0) against the cell above (shift 5)
            q = base + np.array([0.5, 0.5, 0.8]) + rng.uniform(-0.03, 0.03, 3)
            r = rng.uniform(0.33, 0.40)
            early, ldir, side = q + r * x, z, 1
        late = _near_tie_partner(q, early, ldir, side)
        if late is None:
            continue
        vq, ve, vl = np.floor(q), np.floor(early), np.floor(late)
        se = np.where((SHIFTS == (ve - vq)).all(1))[0]
        sl = np.where((SHIFTS == (vl - vq)).all(1))[0]
        assert len(se) == 1 and len(sl) == 1 and se[0] <= sl[0], (kind, q, early, late)
        assert d2(late, q) < d2(early, q) and np.sqrt(d2(late, q)) == np.sqrt(d2(early, q))
        cluster = [early, late]
        for _ in range(fillers):  # farther than both, anywhere in the 27 cells, clear of AddPoints' spacing rule
            f = q + rng.uniform(-1.4, 1.4, 3)
            if d2(f, q) > 1.3 * d2(early, q) and all(np.linalg.norm(f - p) > 0.3 for p in cluster):
                cluster.append(f)
        want.append(len(pts))  # `early` goes in first
        pts.extend(cluster)
        queries.append(q)
    return np.array(pts), np.array(queries), np.array(want)


def brute_reference_choice(pts, voxel_of, q):
    """the reference's loops restated: voxels in shift order, points in insertion order, strict '<' on the NORM"""
    vq = np.floor(q).astype(int)
    best, best_n = -1, np.inf
    for sh in SHIFTS:
        key = tuple(vq + sh)
        for i in voxel_of.get(key, []):
            n = np.sqrt(d2(pts[i], q))
            if n < best_n:
                best, best_n = i, n
    return best
