"""The inequality the near-tie detection of the searches rests on (kicp_search.hpp, kNormTie = 1 + 2^-50).

The reference chooses by (p - q).norm() with strict '<' (VoxelHashMap.cpp:58-63); the device compares squared distances and
hands a query to the search with the reference's own comparison whenever some other candidate's squared distance lies within
kNormTie of the minimum.  That is complete only if squared distances FARTHER apart than that can never have equal rounded
roots: for doubles a < b with b > fl(a * kNormTie), sqrt(a) < sqrt(b) in IEEE arithmetic.  Checked here at the boundary
(b = the first double beyond the limit) for mantissas spread over the binades, both parities of the exponent (the root
halves it) and the edges of a binade, where the spacing of the roots is coarsest relative to the squares'.  CPU only."""
import numpy as np

K_NORM_TIE = float.fromhex("0x1.0000000000004p+0")


def _samples():
    rng = np.random.default_rng(7)
    mant = np.concatenate([
        rng.uniform(1.0, 2.0, 400_000),
        1.0 + rng.uniform(0.0, 1e-9, 50_000),            # just above a power of two
        2.0 - rng.uniform(0.0, 1e-9, 50_000),            # just below the next
        np.nextafter(1.0, 2.0) * np.ones(1), np.ones(1), np.nextafter(2.0, 1.0) * np.ones(1),
    ])
    out = []
    for e in (-40, -21, -20, -1, 0, 1, 2, 7, 8, 13, 14, 40, 41):  # squared distances from (1e-6 m)^2 to (1e6 m)^2, odd and even exponents
        out.append(np.ldexp(mant, e))
    return np.concatenate(out)


def test_squares_beyond_the_detection_limit_have_different_roots():
    a = _samples()
    lim = a * K_NORM_TIE                      # what group_norm_tie computes
    b = np.nextafter(lim, np.inf)             # the closest squared distance that is NOT flagged
    ra, rb = np.sqrt(a), np.sqrt(b)
    assert np.all(rb > ra), int(np.sum(rb <= ra))
    # ... with room: even half the margin would do (the bound is 1 + 2^-51 + O(2^-104)), a quarter would not
    half = np.nextafter(a * float.fromhex("0x1.0000000000002p+0"), np.inf)
    assert np.all(np.sqrt(half) > ra)
    quarter = np.nextafter(a * float.fromhex("0x1.0000000000001p+0"), np.inf)
    assert np.any(np.sqrt(quarter) == ra)


def test_ties_in_norm_do_exist_within_the_limit():
    """the detection is not vacuous: neighbouring squared distances collide in their roots about half of the time"""
    a = _samples()[:200_000]
    b = np.nextafter(a, np.inf)
    collide = np.sqrt(a) == np.sqrt(b)
    assert 0.3 < collide.mean() < 0.8, collide.mean()
    assert np.all(b <= a * K_NORM_TIE)        # ... and every such pair is inside the limit, i.e. flagged
