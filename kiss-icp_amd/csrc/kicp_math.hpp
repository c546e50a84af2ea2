// kicp_math.hpp -- small f64 algebra shared by the HIP kernels and the host side of libkicp.
//
// SE(3) follows the storage and formulas of the reference's dependencies so that results agree
// with the reference CPU path to rounding: Sophus 1.24.6 (so3.hpp / se3.hpp: unit quaternion +
// translation, exp/log/product/inverse) and Eigen 3.4.0 (Quaternion <-> Matrix3, pivoted LDLT of
// Eigen/src/Cholesky/LDLT.h).  Call sites in the reference: core/Registration.cpp:57,86,156-166,
// core/Preprocessing.cpp:68,78, core/Threshold.cpp:40-42, pipeline/KissICP.cpp:47,57,62.
//
// Compile with -ffp-contract=off: the reference is a non-FMA x86-64 build, and keeping
// multiply/add unfused makes voxel assignment and nearest-neighbour decisions agree bit for bit
// whenever the inputs do.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#define KICP_HD __host__ __device__ __forceinline__

namespace kicp {

constexpr double kSophusEps = 1e-10;

struct SE3 {
    double q[4];  // x, y, z, w
    double t[3];
};

KICP_HD void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
KICP_HD double sqnorm3(double x, double y, double z) { return (x * x + y * y) + z * z; }

KICP_HD SE3 se3_identity() {
    SE3 T;
    T.q[0] = T.q[1] = T.q[2] = 0.0;
    T.q[3] = 1.0;
    T.t[0] = T.t[1] = T.t[2] = 0.0;
    return T;
}

// v + w*(2 q x v) + q x (2 q x v)
KICP_HD void quat_rotate(const double q[4], const double v[3], double o[3]) {
    double uv[3], c[3];
    cross3(q, v, uv);
    uv[0] += uv[0];
    uv[1] += uv[1];
    uv[2] += uv[2];
    cross3(q, uv, c);
    o[0] = v[0] + q[3] * uv[0] + c[0];
    o[1] = v[1] + q[3] * uv[1] + c[1];
    o[2] = v[2] + q[3] * uv[2] + c[2];
}

KICP_HD void se3_act(const SE3 &T, const double p[3], double o[3]) {
    double r[3];
    quat_rotate(T.q, p, r);
    o[0] = r[0] + T.t[0];
    o[1] = r[1] + T.t[1];
    o[2] = r[2] + T.t[2];
}

KICP_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = x;
    o[1] = y;
    o[2] = z;
    o[3] = w;
}

// SO3 product with Sophus' first-order renormalisation
KICP_HD void so3_mul(const double a[4], const double b[4], double o[4]) {
    double r[4];
    quat_mul(a, b, r);
    const double sn = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
    if (sn != 1.0) {
        const double scale = 2.0 / (1.0 + sn);
        r[0] *= scale;
        r[1] *= scale;
        r[2] *= scale;
        r[3] *= scale;
    }
    o[0] = r[0];
    o[1] = r[1];
    o[2] = r[2];
    o[3] = r[3];
}

KICP_HD SE3 se3_mul(const SE3 &A, const SE3 &B) {
    SE3 r;
    double rt[3];
    so3_mul(A.q, B.q, r.q);
    quat_rotate(A.q, B.t, rt);
    r.t[0] = A.t[0] + rt[0];
    r.t[1] = A.t[1] + rt[1];
    r.t[2] = A.t[2] + rt[2];
    return r;
}

KICP_HD SE3 se3_inverse(const SE3 &A) {
    SE3 r;
    r.q[0] = -A.q[0];
    r.q[1] = -A.q[1];
    r.q[2] = -A.q[2];
    r.q[3] = A.q[3];
    const double nt[3] = {-A.t[0], -A.t[1], -A.t[2]};
    quat_rotate(r.q, nt, r.t);
    return r;
}

KICP_HD void quat_to_R(const double q[4], double R[9]) {
    const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1.0 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.0 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.0 - (txx + tyy);
}

// Shepperd's method as in Eigen's quaternion-from-matrix assignment (no dynamic indexing, so
// it stays in registers on the device)
KICP_HD void R_to_quat(const double R[9], double q[4]) {
    const double m00 = R[0], m01 = R[1], m02 = R[2];
    const double m10 = R[3], m11 = R[4], m12 = R[5];
    const double m20 = R[6], m21 = R[7], m22 = R[8];
    double t = m00 + m11 + m22;
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m21 - m12) * t;
        q[1] = (m02 - m20) * t;
        q[2] = (m10 - m01) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > (i == 0 ? m00 : m11)) i = 2;
        if (i == 0) {  // j = 1, k = 2
            t = sqrt(m00 - m11 - m22 + 1.0);
            q[0] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m21 - m12) * t;
            q[1] = (m10 + m01) * t;
            q[2] = (m20 + m02) * t;
        } else if (i == 1) {  // j = 2, k = 0
            t = sqrt(m11 - m22 - m00 + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m02 - m20) * t;
            q[2] = (m21 + m12) * t;
            q[0] = (m01 + m10) * t;
        } else {  // i = 2, j = 0, k = 1
            t = sqrt(m22 - m00 - m11 + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m10 - m01) * t;
            q[0] = (m02 + m20) * t;
            q[1] = (m12 + m21) * t;
        }
    }
}

// Sophus::SE3d(Matrix4d) on a row-major 4x4; false where SOPHUS_ENSURE would fire
KICP_HD bool se3_from_matrix(const double M[16], SE3 &T) {
    const double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
    double err2 = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * R[j * 3 + k];
            s -= (i == j) ? 1.0 : 0.0;
            err2 += s * s;
        }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    R_to_quat(R, T.q);
    T.t[0] = M[3];
    T.t[1] = M[7];
    T.t[2] = M[11];
    return (sqrt(err2) < kSophusEps) && (det > 0.0);
}

KICP_HD void se3_matrix(const SE3 &T, double M[16]) {
    double R[9];
    quat_to_R(T.q, R);
    M[0] = R[0];
    M[1] = R[1];
    M[2] = R[2];
    M[3] = T.t[0];
    M[4] = R[3];
    M[5] = R[4];
    M[6] = R[5];
    M[7] = T.t[1];
    M[8] = R[6];
    M[9] = R[7];
    M[10] = R[8];
    M[11] = T.t[2];
    M[12] = M[13] = M[14] = 0.0;
    M[15] = 1.0;
}

KICP_HD void hat3(const double w[3], double O[9]) {
    O[0] = 0.0;
    O[1] = -w[2];
    O[2] = w[1];
    O[3] = w[2];
    O[4] = 0.0;
    O[5] = -w[0];
    O[6] = -w[1];
    O[7] = w[0];
    O[8] = 0.0;
}

KICP_HD void mat3_mul(const double A[9], const double B[9], double C[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

// Sophus SE3::exp, a = (upsilon, omega)
KICP_HD SE3 se3_exp(const double a[6]) {
    SE3 out;
    const double w[3] = {a[3], a[4], a[5]};
    const double theta_sq = sqnorm3(w[0], w[1], w[2]);
    double imag, real, theta, sh = 0.0, ch = 1.0;
    if (theta_sq < kSophusEps * kSophusEps) {
        theta = 0.0;
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        theta = sqrt(theta_sq);
        sincos(0.5 * theta, &sh, &ch);  // one range reduction serves all four trig values below
        imag = sh / theta;
        real = ch;
    }
    out.q[0] = imag * w[0];
    out.q[1] = imag * w[1];
    out.q[2] = imag * w[2];
    out.q[3] = real;
    double Om[9], Om2[9], V[9];
    hat3(w, Om);
    mat3_mul(Om, Om, Om2);
    if (theta < kSophusEps) {
        quat_to_R(out.q, V);
    } else {
        // Sophus: (1 - cos(theta)) / theta^2 and (theta - sin(theta)) / theta^3.  cos(theta) and
        // sin(theta) come from the half angle (cos = 1 - 2 sin^2(theta/2), sin = 2 sin cos) and are
        // rounded to double before the subtraction, like the library calls they replace.
        const double theta2 = theta * theta;
        const double cos_t = 1.0 - 2.0 * sh * sh;
        const double sin_t = 2.0 * sh * ch;
        const double c1 = (1.0 - cos_t) / theta2;
        const double c2 = (theta - sin_t) / (theta2 * theta);
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) out.t[i] = V[i * 3 + 0] * a[0] + V[i * 3 + 1] * a[1] + V[i * 3 + 2] * a[2];
    return out;
}

// Sophus SE3::log
KICP_HD void se3_log(const SE3 &A, double a[6]) {
    const double sqn = sqnorm3(A.q[0], A.q[1], A.q[2]);
    const double qw = A.q[3];
    double two_atan_nbyw_by_n, theta;
    if (sqn < kSophusEps * kSophusEps) {
        const double sqw = qw * qw;
        two_atan_nbyw_by_n = 2.0 / qw - (2.0 / 3.0) * sqn / (qw * sqw);
        theta = 2.0 * sqn / qw;
    } else {
        const double n = sqrt(sqn);
        const double atan_nbyw = (qw < 0.0) ? atan2(-n, -qw) : atan2(n, qw);
        two_atan_nbyw_by_n = 2.0 * atan_nbyw / n;
        theta = two_atan_nbyw_by_n * n;
    }
    const double w[3] = {two_atan_nbyw_by_n * A.q[0], two_atan_nbyw_by_n * A.q[1],
                         two_atan_nbyw_by_n * A.q[2]};
    a[3] = w[0];
    a[4] = w[1];
    a[5] = w[2];
    double Om[9], Om2[9], Vi[9];
    hat3(w, Om);
    mat3_mul(Om, Om, Om2);
    if (fabs(theta) < kSophusEps) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1.0 / 12.0) * Om2[i];
    } else {
        const double half = 0.5 * theta;
        const double c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
#pragma unroll
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = Vi[i * 3 + 0] * A.t[0] + Vi[i * 3 + 1] * A.t[1] + Vi[i * 3 + 2] * A.t[2];
}

// Eigen::AngleAxisd(R).angle() for R = rotationMatrix(q)  (core/Threshold.cpp:40)
KICP_HD double rotation_angle(const double q_in[4]) {
    double R[9], q[4];
    quat_to_R(q_in, R);
    R_to_quat(R, q);
    const double n = sqrt(sqnorm3(q[0], q[1], q[2]));
    return (n != 0.0) ? 2.0 * atan2(n, fabs(q[3])) : 0.0;
}

// Eigen::LDLT<Matrix6d>(A).solve(b): symmetric pivoting on the largest remaining |diagonal|,
// pivots with |D_i| <= DBL_MIN give a zero component.  Fully unrolled with compile-time
// indices (swaps are done by predicated selects) so it stays in registers on the device.  (Measured, same box, round 3:
// making the pivot index wave-uniform (readfirstlane) so that the fifteen candidate swaps become branches the wave
// skips executes fewer instructions and is 1.7 % SLOWER per frame -- 34 taken-or-not branches in a 1 us routine.)
KICP_HD void ldlt6_solve(const double A[36], const double b[6], double x[6]) {
    constexpr int N = 6;
    double m[N][N];
    double d[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        d[i] = b[i];
#pragma unroll
        for (int j = 0; j < N; ++j) m[i][j] = A[i * N + j];
    }
    int transp[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int big = k;
        double bigv = fabs(m[k][k]);
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double v = fabs(m[i][i]);
            if (v > bigv) {
                bigv = v;
                big = i;
            }
        }
        transp[k] = big;
        // symmetric permutation k <-> big on the lower triangle; written as a full symmetric
        // swap of row/col k and row/col `big` of the symmetric completion, by selects.
#pragma unroll
        for (int cand = k + 1; cand < N; ++cand) {
            if (cand == big) {
                // rows/cols k and cand of the lower-stored symmetric matrix
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    const double t = m[k][j];
                    m[k][j] = m[cand][j];
                    m[cand][j] = t;
                }
#pragma unroll
                for (int i = cand + 1; i < N; ++i) {
                    const double t = m[i][k];
                    m[i][k] = m[i][cand];
                    m[i][cand] = t;
                }
                {
                    const double t = m[k][k];
                    m[k][k] = m[cand][cand];
                    m[cand][cand] = t;
                }
#pragma unroll
                for (int i = k + 1; i < cand; ++i) {
                    const double t = m[i][k];
                    m[i][k] = m[cand][i];
                    m[cand][i] = t;
                }
            }
        }
        if (k > 0) {
            double temp[N];
#pragma unroll
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < k; ++j) acc += m[i][j] * temp[j];
                m[i][k] -= acc;
            }
        }
        const double akk = m[k][k];
        if (fabs(akk) > 0.0) {
#pragma unroll
            for (int i = k + 1; i < N; ++i) m[i][k] /= akk;
        }
    }
    // dst = P b
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int cand = k + 1; cand < N; ++cand)
            if (transp[k] == cand) {
                const double t = d[k];
                d[k] = d[cand];
                d[cand] = t;
            }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) d[i] -= m[i][j] * d[j];
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = (fabs(m[i][i]) > DBL_MIN) ? d[i] / m[i][i] : 0.0;
#pragma unroll
    for (int i = N - 1; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) d[i] -= m[j][i] * d[j];
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
#pragma unroll
        for (int cand = k + 1; cand < N; ++cand)
            if (transp[k] == cand) {
                const double t = d[k];
                d[k] = d[cand];
                d[cand] = t;
            }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = d[i];
}

// ---- the normal equations of point-to-point ICP, solved through their structure -----------------------------------------
// BuildLinearSystem's J = [I | -hat(s)] (Registration.cpp:84-86) makes the top-left block of J^T W J the SCALAR matrix
// a I, a = sum w: with m = sum w s, C = sum w (|s|^2 I - s s^T), b1 = -sum w r, b2 = -sum w (s x r) the system
//   [ a I    -[m]x ] [t]   [b1]
//   [ [m]x    C    ] [o] = [b2]
// reduces to the 3 x 3 symmetric one  (C - (|m|^2 I - m m^T) / a) o = b2 - (m x b1) / a,  t = (b1 + m x o) / a --
// some forty operations and four divisions on one dependent chain instead of the ~970 instructions and 21 divisions
// of the pivoted 6 x 6 LDLT (2.7 us of every 15 us iteration, on every workgroup).  It is the same solution in exact
// arithmetic; in floating point it differs from Eigen's LDLT (Registration.cpp:156) by rounding, far inside the
// north star's 1e-4 -- and only well-conditioned systems take this road: S are the kernel's sums (kicp_icp.hip); false
// (x untouched) when a or a pivot of the 3 x 3 elimination is small against its diagonal, and the caller solves the
// 6 x 6 system the reference's way, zero-pivot rule and all.
KICP_HD bool schur3_solve(const double S[16], double x[6]) {
    const double a = S[0];
    if (!(a > 1e-12)) return false;
    const double m0 = S[1], m1 = S[2], m2 = S[3];
    const double b10 = -S[10], b11 = -S[11], b12 = -S[12];
    const double ia = 1.0 / a;
    const double mm = (m0 * m0 + m1 * m1) + m2 * m2;
    // M = C - (|m|^2 I - m m^T) / a, lower triangle
    double M00 = S[4] - (mm - m0 * m0) * ia, M11 = S[7] - (mm - m1 * m1) * ia, M22 = S[9] - (mm - m2 * m2) * ia;
    double M10 = S[5] + (m0 * m1) * ia, M20 = S[6] + (m0 * m2) * ia, M21 = S[8] + (m1 * m2) * ia;
    // rhs = b2 - (m x b1) / a
    double r0 = -S[13] - (m1 * b12 - m2 * b11) * ia, r1 = -S[14] - (m2 * b10 - m0 * b12) * ia, r2 = -S[15] - (m0 * b11 - m1 * b10) * ia;
    const double scale = fmax(fmax(fabs(S[4]), fabs(S[7])), fabs(S[9]));
    const double tiny = scale * 1e-9;
    // LDL^T without pivoting, guarded: a well-conditioned positive definite M has pivots of the order of its diagonal
    if (!(M00 > tiny)) return false;
    const double l10 = M10 / M00, l20 = M20 / M00;
    const double d1 = M11 - l10 * M10;
    if (!(d1 > tiny)) return false;
    const double l21 = (M21 - l20 * M10) / d1;
    const double d2 = M22 - l20 * M20 - l21 * (l21 * d1);
    if (!(d2 > tiny)) return false;
    // forward, diagonal, backward
    const double y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
    const double o2 = y2 / d2;
    const double o1 = y1 / d1 - l21 * o2;
    const double o0 = y0 / M00 - l10 * o1 - l20 * o2;
    x[3] = o0;
    x[4] = o1;
    x[5] = o2;
    // t = (b1 + m x o) / a
    x[0] = (b10 + (m1 * o2 - m2 * o1)) * ia;
    x[1] = (b11 + (m2 * o0 - m0 * o2)) * ia;
    x[2] = (b12 + (m0 * o1 - m1 * o0)) * ia;
    return true;
}

// ---- voxel keys ---------------------------------------------------------------------------
// PointToVoxel (core/VoxelUtils.hpp:33-37): floor(p / voxel_size) per axis, IEEE divide.
// A voxel is packed into one 64-bit word, 21 bits per axis (offset binary); the top bit is
// never set, which leaves room for the EMPTY / TOMBSTONE sentinels of the device hash.
constexpr uint64_t kKeyEmpty = ~0ull;
constexpr uint64_t kKeyTomb = ~0ull - 1ull;
constexpr int kVoxelLimit = 1 << 20;

KICP_HD bool voxel_in_range(int x, int y, int z) {
    return x > -kVoxelLimit && x < kVoxelLimit && y > -kVoxelLimit && y < kVoxelLimit &&
           z > -kVoxelLimit && z < kVoxelLimit;
}
KICP_HD uint64_t pack_voxel(int x, int y, int z) {
    return ((uint64_t)(uint32_t)(x + kVoxelLimit) << 42) | ((uint64_t)(uint32_t)(y + kVoxelLimit) << 21) |
           (uint64_t)(uint32_t)(z + kVoxelLimit);
}
KICP_HD void unpack_voxel(uint64_t key, int &x, int &y, int &z) {
    x = (int)((key >> 42) & 0x1FFFFF) - kVoxelLimit;
    y = (int)((key >> 21) & 0x1FFFFF) - kVoxelLimit;
    z = (int)(key & 0x1FFFFF) - kVoxelLimit;
}
KICP_HD int voxel_coord(double p, double voxel_size) { return (int)floor(p / voxel_size); }
// Same value without the IEEE divide on the common path: p * (1 / voxel_size) is within ~2 ulp of
// the correctly rounded quotient, so the two floors can only differ when the product sits within
// a few ulp of an integer -- then (and only then) the exact divide decides.
KICP_HD int voxel_coord_fast(double p, double voxel_size, double inv_voxel_size) {
    const double q = p * inv_voxel_size;
    const double f = floor(q);
    const double r = q - f;                      // exact (Sterbenz) in [0, 1)
    const double eps = fabs(q) * 0x1p-49 + DBL_MIN;
    if (r < eps || r > 1.0 - eps) return (int)floor(p / voxel_size);
    return (int)f;
}

KICP_HD uint32_t hash_key(uint64_t key, uint32_t mask) {
    uint64_t h = key * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return (uint32_t)(h >> 17) & mask;
}

}  // namespace kicp
