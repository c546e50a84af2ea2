"""Anchors for the THIRD-PARTY arithmetic on the registration path that share no code with oracle/kiss_oracle.c.

The reference takes its 6x6 solve from Eigen 3.4.0 (`JTJ.ldlt().solve(-JTr)`, Registration.cpp:156) and its SE(3)
exponential / logarithm / product from Sophus 1.24.6 (Registration.cpp:157,161,166; Preprocessing.cpp:68,78;
Threshold.cpp:40-42).  Neither library is installed here, so the oracle restates them and `oracle/_ref` (built over
stand-in headers) cannot pin that restatement (tests/test_ref_pins_oracle.py::test_third_party_arithmetic says which
mode it ran in).  What CAN be held independently:

* exp / log against 80-digit arithmetic (mpmath), across Sophus' small-angle branch at theta = 1e-10 and up to pi --
  a slip in a coefficient or a branch condition shows as an error far above the cancellation slack of the formulas;
* the LDLT against exact rational arithmetic (fractions): the solution of a non-singular system is unique whatever
  the pivot order, so ties on the diagonal may not move it beyond rounding; for singular systems Eigen's algorithm as
  published -- left-looking: the pivot of step k is chosen among diagonal entries no elimination has touched yet; a
  column is divided by its pivot iff that is non-zero; components whose pivot is zero come back 0 -- is replayed with
  exact fractions on integer matrices whose zero pivots are exact in floating point as well.  (A first version of
  this replay eliminated right-looking, i.e. pivoted on Schur complements: the oracle disagreed with it on WHICH of
  two identical unknowns comes back zero, and the oracle was the one following LDLT.h.)
"""
from fractions import Fraction

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    oracle.lib()
    return oracle


# ---- SE3 exp / log vs mpmath ---------------------------------------------------------------------------------
def _mp_exp(a):
    """exp of the twist (upsilon, omega) with 80 digits: closed forms, no branch needed at this precision"""
    mp.mp.dps = 80
    u = mp.matrix([mp.mpf(float(x)) for x in a[:3]])
    w = [mp.mpf(float(x)) for x in a[3:]]
    th = mp.sqrt(w[0] ** 2 + w[1] ** 2 + w[2] ** 2)
    K = mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    eye = mp.eye(3)
    if th == 0:
        return eye, u
    A, B, Cc = mp.sin(th) / th, (1 - mp.cos(th)) / th ** 2, (th - mp.sin(th)) / th ** 3
    R = eye + A * K + B * (K * K)
    V = eye + B * K + Cc * (K * K)
    return R, V * u


THETAS = [1e-14, 1e-12, 5e-11, 9.9e-11, 1.01e-10, 2e-10, 1e-9, 1e-7, 1e-4, 1e-2, 0.3, 1.0, 2.5, 3.0, 3.14159]


@pytest.mark.parametrize("theta", THETAS)
def test_se3_exp_against_80_digit_arithmetic(O, theta):
    rng = np.random.default_rng(int(abs(np.log10(theta)) * 13) + 7)
    for _ in range(12):
        a = rng.normal(size=6)
        a[:3] *= rng.choice([1e-3, 1.0, 50.0])
        a[3:] *= theta / np.linalg.norm(a[3:])
        R, t = _mp_exp(a)
        got = O.se3_exp(a)
        Rf = np.array([[float(R[i, j]) for j in range(3)] for i in range(3)])
        tf = np.array([float(t[i]) for i in range(3)])
        # rotation: a unit quaternion round trip costs a few ulp
        np.testing.assert_allclose(got[:3, :3], Rf, rtol=0, atol=8e-16 * 4)
        # translation: Sophus evaluates (1 - cos t)/t^2 and (t - sin t)/t^3 literally for t >= 1e-10 (cancellation
        # 2.3e-16 / t relative to |upsilon|) and uses V = R below that (error t / 2): min of the two, as a bound
        th = np.linalg.norm(a[3:])
        slack = min(0.5 * th, 4.6e-16 / th) * np.linalg.norm(a[:3]) + 4e-16 * max(1.0, np.abs(tf).max())
        np.testing.assert_allclose(got[:3, 3], tf, rtol=0, atol=slack)
        assert got[3].tolist() == [0.0, 0.0, 0.0, 1.0]


@pytest.mark.parametrize("theta", THETAS)
def test_se3_log_inverts_the_80_digit_exp(O, theta):
    """log of the CORRECTLY ROUNDED exp (built from the 80-digit one, not from the oracle's own exp)"""
    rng = np.random.default_rng(int(abs(np.log10(theta)) * 17) + 3)
    for _ in range(12):
        a = rng.normal(size=6)
        a[3:] *= theta / np.linalg.norm(a[3:])
        R, t = _mp_exp(a)
        M = np.eye(4)
        M[:3, :3] = [[float(R[i, j]) for j in range(3)] for i in range(3)]
        M[:3, 3] = [float(t[i]) for i in range(3)]
        got = O.se3_log(M)
        th = np.linalg.norm(a[3:])
        # the rounded matrix carries ~1e-16 of noise: in the rotation vector that is ~1e-16 absolute (more near pi,
        # where the axis comes from a difference of off-diagonal terms), in upsilon the cancellation slack of V^-1
        tol_w = 4e-16 * max(1.0, th) / max(np.sin(min(th, np.pi - 1e-6)) if th > 1.0 else 1.0, 1e-6) + 1e-15
        np.testing.assert_allclose(got[3:], a[3:], rtol=0, atol=tol_w * 8)
        slack = min(0.5 * th, 4.6e-16 / th) * np.linalg.norm(a[:3]) * 2 + 1e-14 * max(1.0, np.abs(a[:3]).max())
        np.testing.assert_allclose(got[:3], a[:3], rtol=0, atol=slack + np.linalg.norm(a[:3]) * tol_w * 8)


# ---- LDLT vs exact rationals ---------------------------------------------------------------------------------
def _exact_solve(A, b):
    """Gauss-Jordan over the rationals (non-singular A)"""
    n = len(b)
    M = [[Fraction(int(A[i][j])) for j in range(n)] + [Fraction(int(b[i]))] for i in range(n)]
    for c in range(n):
        p = next(r for r in range(c, n) if M[r][c] != 0)
        M[c], M[p] = M[p], M[c]
        M[c] = [x / M[c][c] for x in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0:
                M[r] = [x - M[r][c] * y for x, y in zip(M[r], M[c])]
    return [M[i][n] for i in range(n)]


def _exact_pivoted_ldlt_solve(A, b):
    """Eigen 3.4.0 LDLT<Lower> as published (Eigen/src/Cholesky/LDLT.h: ldlt_inplace<Lower>::unblocked, then
    _solve_impl_transposed), over the rationals.  The factorisation is LEFT-LOOKING: at step k the pivot is the
    largest |diagonal| of the trailing part of the matrix AS STORED -- entries that no elimination step has touched yet,
    i.e. the original diagonal in the current (swapped) layout, the first such on ties --; rows / columns k and pivot
    are swapped; only then column k is brought up to date from the columns before it, and divided by its diagonal iff
    that is non-zero.  Solve: permute, forward-substitute, divide by D where D is non-zero (else 0), back-substitute,
    permute back."""
    n = len(b)
    M = [[Fraction(int(A[i][j])) for j in range(n)] for i in range(n)]  # (kept symmetric: both triangles swapped alike)
    transp = []
    for k in range(n):
        piv = max(range(k, n), key=lambda i: (abs(M[i][i]), -i))  # largest, first on ties
        transp.append(piv)
        if piv != k:
            M[k], M[piv] = M[piv], M[k]
            for r in range(n):
                M[r][k], M[r][piv] = M[r][piv], M[r][k]
        if k > 0:
            temp = [M[j][j] * M[k][j] for j in range(k)]
            M[k][k] -= sum(M[k][j] * temp[j] for j in range(k))
            for i in range(k + 1, n):
                M[i][k] -= sum(M[i][j] * temp[j] for j in range(k))
        if M[k][k] != 0:
            for i in range(k + 1, n):
                M[i][k] /= M[k][k]
        for i in range(k + 1, n):  # (the upper triangle mirrors the lower one for the swaps to come)
            M[k][i] = M[i][k]
    y = [Fraction(int(v)) for v in b]
    for k in range(n):
        y[k], y[transp[k]] = y[transp[k]], y[k]
    for i in range(n):
        for j in range(i):
            y[i] -= M[i][j] * y[j]
    y = [y[i] / M[i][i] if M[i][i] != 0 else Fraction(0) for i in range(n)]
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            y[i] -= M[j][i] * y[j]
    for k in range(n - 1, -1, -1):
        y[k], y[transp[k]] = y[transp[k]], y[k]
    return y


def test_ldlt6_unique_solution_whatever_the_pivot_order(O):
    """non-singular integer systems, some with tied diagonal entries (which pivot comes first must not matter beyond
    rounding): the float solve against the exact rational solution"""
    rng = np.random.default_rng(23)
    for k in range(60):
        J = rng.integers(-6, 7, size=(12, 6))
        A = J.T @ J
        if k % 3 == 0:  # ties on the diagonal
            A[1, 1] = A[3, 3] = A[5, 5] = max(A[1, 1], A[3, 3], A[5, 5]) + 3
        if np.linalg.matrix_rank(A) < 6:
            continue
        b = rng.integers(-20, 21, size=6)
        want = np.array([float(x) for x in _exact_solve(A.tolist(), b.tolist())])
        got = O.ldlt6_solve(A.astype(float), b.astype(float))
        cond = np.linalg.cond(A.astype(float))
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-16 * cond * max(1.0, np.abs(want).max()))


def test_ldlt6_zero_pivot_rule_replayed_with_exact_fractions(O):
    """positive semi-definite integer matrices whose zero pivots are EXACT in floating point too -- an unknown that does
    not occur (zero row and column) or occurs twice (two identical columns of J: after the first of the pair is
    eliminated the other's pivot is a - (a / a) a (a / a) = 0 exactly, and the pair ties on the diagonal, so the test
    also holds the first-largest pivot rule) -- with consistent right-hand sides: the float result equals the exact
    replay of Eigen's algorithm (pivot choice, `divide iff the pivot is non-zero`, pseudo-inverse of D), component for
    component, and it solves the system.  (A zero pivot that floating point turns into rounding residue is divided by;
    the component is then arbitrary -- DESIGN.md section 2, "where parity is not defined" -- which is why the matrices
    are built this way.)"""
    rng = np.random.default_rng(29)
    done = 0
    for k in range(300):
        J = rng.integers(-3, 4, size=(9, 6))
        cols = rng.permutation(6)
        kind = k % 3
        if kind in (0, 2):
            J[:, cols[0]] = 0  # an unknown that does not occur
        if kind in (1, 2):
            J[:, cols[1]] = J[:, cols[2]]  # an unknown that occurs twice
        if kind == 1 and k % 2:
            J[:, cols[3]] = J[:, cols[4]]  # ... two such pairs
        A = J.T @ J
        expect_rank = 6 - {0: 1, 1: 1 + (k % 2), 2: 2}[kind]
        if np.linalg.matrix_rank(A) != expect_rank:
            continue
        b = A @ rng.integers(-4, 5, size=6)  # consistent
        want = np.array([float(x) for x in _exact_pivoted_ldlt_solve(A.tolist(), b.tolist())])
        got = O.ldlt6_solve(A.astype(float), b.astype(float))
        np.testing.assert_allclose(A @ got, b, rtol=0, atol=1e-9 * max(1.0, np.abs(b).max()))
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=str((k, A.tolist())))
        assert (want == 0).sum() >= 6 - expect_rank  # (the dropped components are there)
        done += 1
    assert done >= 150


def test_ldlt6_tiny_pivots_are_divided_not_dropped(O):
    """Eigen drops a component only when |D_ii| <= the smallest NORMAL double (its pseudo-inverse of D uses
    numeric_limits::min()); anything larger is divided, however small"""
    tiny = np.finfo(np.float64).tiny
    A = np.diag([4.0, 8 * tiny, 2.0, tiny / 4, 1.0, 8.0])
    b = np.array([8.0, 16 * tiny, 2.0, tiny / 2, 1.0, 4.0])
    np.testing.assert_array_equal(O.ldlt6_solve(A, b), np.array([2.0, 2.0, 1.0, 0.0, 1.0, 0.5]))
