/*
 * kiss_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See kiss_oracle.h.
 *
 * Pinned against the reference's own sources built by oracle/ref_build (tests/test_ref_pins_oracle.py);
 * PARITY UNPINNED for the third-party arithmetic (Eigen LDLT, Sophus SE3) only -- see kiss_oracle.h.
 *
 * Restates, from /root/reference (PRBonn/kiss-icp v1.2.3):
 *   cpp/kiss_icp/core/Registration.cpp:55-167      AlignPointsToMap + helpers
 *   cpp/kiss_icp/core/VoxelHashMap.cpp:35-132      27-voxel NN search, AddPoints, prune
 *   cpp/kiss_icp/core/VoxelUtils.hpp:32-51         PointToVoxel
 *   cpp/kiss_icp/core/VoxelUtils.cpp:7-21          VoxelDownsample
 *   cpp/kiss_icp/core/Preprocessing.cpp:55-95      deskew + range crop
 *   cpp/kiss_icp/core/Threshold.{hpp,cpp}          adaptive threshold
 *   cpp/kiss_icp/pipeline/KissICP.cpp:35-75        RegisterFrame / Voxelize
 * and the third-party arithmetic those call (not under /root/reference):
 *   Eigen 3.4.0   LDLT (pivoted, Eigen/src/Cholesky/LDLT.h), Quaternion<->Matrix3, AngleAxis
 *   Sophus 1.24.6 SO3/SE3 exp, log, product, inverse, action (sophus/so3.hpp, se3.hpp)
 *
 * Parallel ONLY where the reference uses TBB: DataAssociation (Registration.cpp:66),
 * BuildLinearSystem (Registration.cpp:101) and the deskew loop (Preprocessing.cpp:70).
 * Build: gcc -O3 -fopenmp -ffp-contract=off -shared -fPIC (the reference is a generic -O3
 * Release build without FMA contraction on x86-64).
 */
#include "kiss_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SOPHUS_EPS 1e-10 /* Sophus::Constants<double>::epsilon() */

/* CPUs this process may really use: OpenMP's count (the affinity mask), capped by the container's CPU quota
 * (cgroup v2 cpu.max: "<quota> <period>").  A GPU box's container sees 256 CPUs and may use 16: one thread per VISIBLE CPU
 * made an 8-frame drive take 91 s instead of 0.1 (profiles/r05_u_first_rccl_probe.txt). */
int ko_num_procs(void) {
#ifdef _OPENMP
    int n = omp_get_num_procs();
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) {
                const long c = (quota + period - 1) / period;
                if (c >= 1 && c < n) n = (int)c;
            }
        }
        fclose(f);
    }
    return n > 0 ? n : 1;
#else
    return 1;
#endif
}

static int resolve_threads(int max_threads) {
    /* Registration.cpp:130-131: max_num_threads > 0 ? it : tbb max_concurrency().
     * KISS_ORACLE_THREADS caps the "all cores" default: on a 256-core box forking 256 OpenMP
     * threads twice per ICP iteration over ~2 k points costs more than the work (the test-suite
     * sets it; the timed CPU baseline of bench.py passes explicit thread counts and is unaffected). */
    int hw = ko_num_procs();
    if (max_threads > 0) return max_threads;
    const char *cap = getenv("KISS_ORACLE_THREADS");
    if (cap) {
        const int c = atoi(cap);
        if (c > 0 && c < hw) return c;
    }
    return hw;
}

/* ======================================================================================
 * small linear algebra
 * ==================================================================================== */
static inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen: Vector3d::squaredNorm() -- 3-term unrolled redux ((x*x + y*y) + z*z) */
static inline double sqnorm3(const double a[3]) { return (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]; }
static inline double norm3(const double a[3]) { return sqrt(sqnorm3(a)); }

/* Eigen QuaternionBase::_transformVector (Eigen/src/Geometry/Quaternion.h):
 *   uv = q.vec x v; uv += uv; return v + q.w * uv + q.vec x uv                            */
static inline void quat_rotate(const double q[4], const double v[3], double o[3]) {
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

/* Eigen quaternion product a*b (x,y,z,w storage) */
static inline void quat_mul(const double a[4], const double b[4], double o[4]) {
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = x;
    o[1] = y;
    o[2] = z;
    o[3] = w;
}

/* Eigen QuaternionBase::toRotationMatrix; R row-major 3x3 */
static void quat_to_R(const double q[4], double R[9]) {
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

/* Eigen quaternionbase_assign_impl<Matrix3> (Shepperd's method) */
static void R_to_quat(const double R[9], double q[4]) {
#define M(i, j) R[(i)*3 + (j)]
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (M(2, 1) - M(1, 2)) * t;
        q[1] = (M(0, 2) - M(2, 0)) * t;
        q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        t = sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (M(k, j) - M(j, k)) * t;
        q[j] = (M(j, i) + M(i, j)) * t;
        q[k] = (M(k, i) + M(i, k)) * t;
    }
#undef M
}

/* ======================================================================================
 * SE(3)  (Sophus 1.24.6 so3.hpp / se3.hpp)
 * ==================================================================================== */
void ko_se3_identity(ko_se3 *T) {
    T->q[0] = T->q[1] = T->q[2] = 0.0;
    T->q[3] = 1.0;
    T->t[0] = T->t[1] = T->t[2] = 0.0;
}

int ko_se3_from_matrix(const double Mx[16], ko_se3 *T) {
    double R[9] = {Mx[0], Mx[1], Mx[2], Mx[4], Mx[5], Mx[6], Mx[8], Mx[9], Mx[10]};
    /* SOPHUS_ENSURE(isOrthogonal(R)) : (R*R^T - I).norm() < 1e-10 ; det > 0 */
    double err2 = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * R[j * 3 + k];
            s -= (i == j) ? 1.0 : 0.0;
            err2 += s * s;
        }
    double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                 R[2] * (R[3] * R[7] - R[4] * R[6]);
    R_to_quat(R, T->q);
    T->t[0] = Mx[3];
    T->t[1] = Mx[7];
    T->t[2] = Mx[11];
    if (!(sqrt(err2) < SOPHUS_EPS) || !(det > 0.0)) return -1;
    return 0;
}

void ko_se3_matrix(const ko_se3 *T, double Mx[16]) {
    double R[9];
    quat_to_R(T->q, R);
    Mx[0] = R[0];
    Mx[1] = R[1];
    Mx[2] = R[2];
    Mx[3] = T->t[0];
    Mx[4] = R[3];
    Mx[5] = R[4];
    Mx[6] = R[5];
    Mx[7] = T->t[1];
    Mx[8] = R[6];
    Mx[9] = R[7];
    Mx[10] = R[8];
    Mx[11] = T->t[2];
    Mx[12] = Mx[13] = Mx[14] = 0.0;
    Mx[15] = 1.0;
}

/* SO3 product (so3.hpp operator*): quaternion product followed by the first-order
 * renormalisation  q *= 2/(1+|q|^2)  when |q|^2 != 1.                                     */
static void so3_mul(const double a[4], const double b[4], double o[4]) {
    double r[4];
    quat_mul(a, b, r);
    const double sn = ((r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]));
    if (sn != 1.0) {
        const double scale = 2.0 / (1.0 + sn);
        r[0] *= scale;
        r[1] *= scale;
        r[2] *= scale;
        r[3] *= scale;
    }
    memcpy(o, r, sizeof r);
}

void ko_se3_act(const ko_se3 *T, const double p[3], double out[3]) {
    /* se3.hpp operator*(Point): so3()*p + translation() */
    double r[3];
    quat_rotate(T->q, p, r);
    out[0] = r[0] + T->t[0];
    out[1] = r[1] + T->t[1];
    out[2] = r[2] + T->t[2];
}

void ko_se3_mul(const ko_se3 *A, const ko_se3 *B, ko_se3 *out) {
    /* se3.hpp operator*: SE3(so3()*other.so3(), translation() + so3()*other.translation()) */
    ko_se3 r;
    double rt[3];
    so3_mul(A->q, B->q, r.q);
    quat_rotate(A->q, B->t, rt);
    r.t[0] = A->t[0] + rt[0];
    r.t[1] = A->t[1] + rt[1];
    r.t[2] = A->t[2] + rt[2];
    *out = r;
}

void ko_se3_inverse(const ko_se3 *A, ko_se3 *out) {
    /* se3.hpp inverse(): invR = so3().inverse() (conjugate); SE3(invR, invR*(-t)) */
    ko_se3 r;
    r.q[0] = -A->q[0];
    r.q[1] = -A->q[1];
    r.q[2] = -A->q[2];
    r.q[3] = A->q[3];
    double nt[3] = {-A->t[0], -A->t[1], -A->t[2]};
    quat_rotate(r.q, nt, r.t);
    *out = r;
}

/* SO3::expAndTheta */
static void so3_exp(const double w[3], double q[4], double *theta_out) {
    const double theta_sq = sqnorm3(w);
    double imag, real, theta;
    if (theta_sq < SOPHUS_EPS * SOPHUS_EPS) {
        theta = 0.0;
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        theta = sqrt(theta_sq);
        const double half = 0.5 * theta;
        const double sh = sin(half);
        imag = sh / theta;
        real = cos(half);
    }
    q[0] = imag * w[0];
    q[1] = imag * w[1];
    q[2] = imag * w[2];
    q[3] = real;
    *theta_out = theta;
}

static void hat3(const double w[3], double O[9]) {
    /* SO3::hat (so3.hpp): [[0,-z,y],[z,0,-x],[-y,x,0]] ; used at Registration.cpp:86 */
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

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

void ko_se3_exp(const double a[6], ko_se3 *out) {
    /* SE3::exp (se3.hpp): omega = a.tail<3>, so3 = SO3::expAndTheta, V as below,
     * translation = V * a.head<3>                                                        */
    const double *w = a + 3;
    double theta;
    so3_exp(w, out->q, &theta);
    double Om[9], Om2[9], V[9];
    hat3(w, Om);
    mat3_mul(Om, Om, Om2);
    if (theta < SOPHUS_EPS) {
        quat_to_R(out->q, V);
        /* Note: that is an accurate expansion! */
    } else {
        const double theta_sq = theta * theta;
        const double c1 = (1.0 - cos(theta)) / theta_sq;
        const double c2 = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
    }
    for (int i = 0; i < 3; ++i)
        out->t[i] = V[i * 3 + 0] * a[0] + V[i * 3 + 1] * a[1] + V[i * 3 + 2] * a[2];
}

/* SO3::logAndTheta */
static void so3_log(const double q[4], double w[3], double *theta_out) {
    const double sqn = sqnorm3(q);
    const double qw = q[3];
    double two_atan_nbyw_by_n, theta;
    if (sqn < SOPHUS_EPS * SOPHUS_EPS) {
        const double sqw = qw * qw;
        two_atan_nbyw_by_n = 2.0 / qw - (2.0 / 3.0) * sqn / (qw * sqw);
        theta = 2.0 * sqn / qw;
    } else {
        const double n = sqrt(sqn);
        const double atan_nbyw = (qw < 0.0) ? atan2(-n, -qw) : atan2(n, qw);
        two_atan_nbyw_by_n = 2.0 * atan_nbyw / n;
        theta = two_atan_nbyw_by_n * n;
    }
    w[0] = two_atan_nbyw_by_n * q[0];
    w[1] = two_atan_nbyw_by_n * q[1];
    w[2] = two_atan_nbyw_by_n * q[2];
    *theta_out = theta;
}

void ko_se3_log(const ko_se3 *A, double a[6]) {
    /* SE3::log (se3.hpp) */
    double theta, w[3];
    so3_log(A->q, w, &theta);
    a[3] = w[0];
    a[4] = w[1];
    a[5] = w[2];
    double Om[9], Om2[9], Vi[9];
    hat3(w, Om);
    mat3_mul(Om, Om, Om2);
    if (fabs(theta) < SOPHUS_EPS) {
        for (int i = 0; i < 9; ++i)
            Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + (1.0 / 12.0) * Om2[i];
    } else {
        const double half = 0.5 * theta;
        const double c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    }
    for (int i = 0; i < 3; ++i)
        a[i] = Vi[i * 3 + 0] * A->t[0] + Vi[i * 3 + 1] * A->t[1] + Vi[i * 3 + 2] * A->t[2];
}

/* ======================================================================================
 * Eigen::LDLT<Matrix6d>::compute + solve  (Eigen 3.4.0, Eigen/src/Cholesky/LDLT.h:
 * ldlt_inplace<Lower>::unblocked and LDLT::_solve_impl_transposed).
 * Symmetric pivoting on the largest |diagonal| of the trailing block; in solve, pivots
 * with |D_i| <= DBL_MIN give a zero component.
 * ==================================================================================== */
void ko_ldlt6_solve(const double A[36], const double b[6], double x[6]) {
    enum { N = 6 };
    double m[N][N];
    int transp[N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) m[i][j] = A[i * N + j];
    /* only the lower triangle is referenced below */
    for (int k = 0; k < N; ++k) {
        int big = k;
        double bigv = fabs(m[k][k]);
        for (int i = k + 1; i < N; ++i) {
            double v = fabs(m[i][i]);
            if (v > bigv) {
                bigv = v;
                big = i;
            }
        }
        transp[k] = big;
        if (k != big) {
            /* symmetric swap of rows/cols k <-> big acting on the lower triangle */
            int s = N - big - 1;
            for (int j = 0; j < k; ++j) {
                double t = m[k][j];
                m[k][j] = m[big][j];
                m[big][j] = t;
            }
            for (int i = 0; i < s; ++i) {
                double t = m[big + 1 + i][k];
                m[big + 1 + i][k] = m[big + 1 + i][big];
                m[big + 1 + i][big] = t;
            }
            {
                double t = m[k][k];
                m[k][k] = m[big][big];
                m[big][big] = t;
            }
            for (int i = k + 1; i < big; ++i) {
                double t = m[i][k];
                m[i][k] = m[big][i];
                m[big][i] = t;
            }
        }
        int rs = N - k - 1;
        if (k > 0) {
            double temp[N];
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double s = 0.0;
            for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
            for (int i = 0; i < rs; ++i) {
                double acc = 0.0;
                for (int j = 0; j < k; ++j) acc += m[k + 1 + i][j] * temp[j];
                m[k + 1 + i][k] -= acc;
            }
        }
        double akk = m[k][k];
        if (rs > 0 && fabs(akk) > 0.0)
            for (int i = 0; i < rs; ++i) m[k + 1 + i][k] /= akk;
    }
    /* solve: dst = P b */
    double d[N];
    for (int i = 0; i < N; ++i) d[i] = b[i];
    for (int k = 0; k < N; ++k) {
        int p = transp[k];
        if (p != k) {
            double t = d[k];
            d[k] = d[p];
            d[p] = t;
        }
    }
    /* L^-1 (unit lower) */
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < i; ++j) d[i] -= m[i][j] * d[j];
    /* pseudo-inverse of D */
    for (int i = 0; i < N; ++i) {
        if (fabs(m[i][i]) > DBL_MIN)
            d[i] /= m[i][i];
        else
            d[i] = 0.0;
    }
    /* L^-T */
    for (int i = N - 1; i >= 0; --i)
        for (int j = i + 1; j < N; ++j) d[i] -= m[j][i] * d[j];
    /* P^T */
    for (int k = N - 1; k >= 0; --k) {
        int p = transp[k];
        if (p != k) {
            double t = d[k];
            d[k] = d[p];
            d[p] = t;
        }
    }
    for (int i = 0; i < N; ++i) x[i] = d[i];
}

/* ======================================================================================
 * VoxelUtils
 * ==================================================================================== */
void ko_point_to_voxel(const double p[3], double voxel_size, int32_t v[3]) {
    /* VoxelUtils.hpp:33-37: static_cast<int>(std::floor(p / voxel_size)) per axis */
    v[0] = (int32_t)floor(p[0] / voxel_size);
    v[1] = (int32_t)floor(p[1] / voxel_size);
    v[2] = (int32_t)floor(p[2] / voxel_size);
}

static inline uint64_t voxel_hash(const int32_t v[3]) {
    /* VoxelUtils.hpp:46-50 (u32 wrap-around products, xor) */
    uint32_t a = (uint32_t)v[0] * 73856093u;
    uint32_t b = (uint32_t)v[1] * 19349669u;
    uint32_t c = (uint32_t)v[2] * 83492791u;
    return (uint64_t)(a ^ b ^ c);
}

/* A flat open-addressing (linear probing, backward-shift deletion) table from voxel key to
 * a dense voxel index -- the stand-in for tsl::robin_map (also flat, open addressing).   */
typedef struct {
    int32_t *keys; /* 3 per bucket */
    int32_t *vals; /* -1 = empty */
    size_t cap;    /* power of two */
    size_t size;
} vtable;

static void vt_init(vtable *t, size_t min_cap) {
    size_t cap = 16;
    while (cap < min_cap) cap <<= 1;
    t->cap = cap;
    t->size = 0;
    t->keys = (int32_t *)malloc(cap * 3 * sizeof(int32_t));
    t->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    for (size_t i = 0; i < cap; ++i) t->vals[i] = -1;
}
static void vt_free(vtable *t) {
    free(t->keys);
    free(t->vals);
    t->keys = NULL;
    t->vals = NULL;
}
static inline size_t vt_bucket(const vtable *t, const int32_t v[3]) {
    uint64_t h = voxel_hash(v);
    h ^= h >> 15; /* the raw hash is weak in its low bits for a power-of-two table */
    h *= 0x9E3779B97F4A7C15ull;
    return (size_t)(h >> 20) & (t->cap - 1);
}
static inline int32_t vt_find(const vtable *t, const int32_t v[3]) {
    size_t i = vt_bucket(t, v);
    for (;;) {
        if (t->vals[i] < 0) return -1;
        const int32_t *k = t->keys + 3 * i;
        if (k[0] == v[0] && k[1] == v[1] && k[2] == v[2]) return t->vals[i];
        i = (i + 1) & (t->cap - 1);
    }
}
static void vt_insert_nogrow(vtable *t, const int32_t v[3], int32_t val) {
    size_t i = vt_bucket(t, v);
    while (t->vals[i] >= 0) i = (i + 1) & (t->cap - 1);
    t->keys[3 * i] = v[0];
    t->keys[3 * i + 1] = v[1];
    t->keys[3 * i + 2] = v[2];
    t->vals[i] = val;
    t->size++;
}
static void vt_insert(vtable *t, const int32_t v[3], int32_t val) {
    if ((t->size + 1) * 2 > t->cap) {
        vtable n;
        vt_init(&n, t->cap * 2);
        for (size_t i = 0; i < t->cap; ++i)
            if (t->vals[i] >= 0) vt_insert_nogrow(&n, t->keys + 3 * i, t->vals[i]);
        vt_free(t);
        *t = n;
    }
    vt_insert_nogrow(t, v, val);
}
static void vt_set(vtable *t, const int32_t v[3], int32_t val) {
    size_t i = vt_bucket(t, v);
    for (;;) {
        const int32_t *k = t->keys + 3 * i;
        if (t->vals[i] >= 0 && k[0] == v[0] && k[1] == v[1] && k[2] == v[2]) {
            t->vals[i] = val;
            return;
        }
        i = (i + 1) & (t->cap - 1);
    }
}
static void vt_erase(vtable *t, const int32_t v[3]) {
    size_t mask = t->cap - 1;
    size_t i = vt_bucket(t, v);
    for (;;) {
        if (t->vals[i] < 0) return;
        const int32_t *k = t->keys + 3 * i;
        if (k[0] == v[0] && k[1] == v[1] && k[2] == v[2]) break;
        i = (i + 1) & mask;
    }
    /* backward-shift deletion */
    size_t j = i;
    for (;;) {
        j = (j + 1) & mask;
        if (t->vals[j] < 0) break;
        size_t home = vt_bucket(t, t->keys + 3 * j);
        /* can the entry at j move to i?  yes iff home is cyclically not in (i, j] */
        int between = (i <= j) ? (home > i && home <= j) : (home > i || home <= j);
        if (!between) {
            memcpy(t->keys + 3 * i, t->keys + 3 * j, 3 * sizeof(int32_t));
            t->vals[i] = t->vals[j];
            i = j;
        }
    }
    t->vals[i] = -1;
    t->size--;
}

/* Output order of VoxelDownsample.  The reference emits the survivors by iterating a tsl::robin_map
 * (VoxelUtils.cpp:17-19), i.e. in BUCKET order of that container -- and the order matters downstream: it
 * decides which points AddPoints' first-come cap / spacing rule accepts (VoxelHashMap.cpp:98-118) and which
 * point of a 1.5 v voxel the second downsample keeps (KissICP.cpp:72-73).
 *   1 (default)  the bucket order of tsl::robin_map 1.4.0 after grid.reserve(frame.size()) and one insert per
 *                distinct voxel in frame order, restated from the container's published algorithm (see
 *                oracle/ref_build/shim/tsl/robin_map.h for the rules; third-party: parity unpinned);
 *   0            ascending original index (what rounds 1-2 of this repository defined). */
static int g_downsample_order = 1;
void ko_set_downsample_order(int order) { g_downsample_order = order ? 1 : 0; }
int ko_get_downsample_order(void) { return g_downsample_order; }

/* std::hash<Voxel> (VoxelUtils.hpp:46-50): u32 wrap-around products, xor-ed, widened */
static inline uint32_t ref_voxel_hash(const int32_t v[3]) {
    return ((uint32_t)v[0] * 73856093u) ^ ((uint32_t)v[1] * 19349669u) ^ ((uint32_t)v[2] * 83492791u);
}

static size_t downsample_tsl_order(const double *xyz, size_t n, double voxel_size, double *out) {
    if (n == 0) return 0;
    /* grid.reserve(n): rehash(ceil(float(n) / max_load_factor 0.5f)), rounded up to a power of two */
    size_t want = (size_t)ceilf((float)n / 0.5f), B = 1;
    while (B < want) B <<= 1;
    const size_t mask = B - 1;
    int32_t *dist = (int32_t *)malloc(B * sizeof(int32_t)); /* distance from the home bucket, -1 = empty */
    uint32_t *who = (uint32_t *)malloc(B * sizeof(uint32_t)); /* index of the point stored in the bucket */
    int32_t *keys = (int32_t *)malloc(B * 3 * sizeof(int32_t));
    for (size_t b = 0; b < B; ++b) dist[b] = -1;
    size_t count = 0;
    for (size_t i = 0; i < n; ++i) {
        int32_t v[3];
        ko_point_to_voxel(xyz + 3 * i, voxel_size, v);
        size_t ib = ref_voxel_hash(v) & mask;
        int32_t d = 0;
        int found = 0;
        while (d <= dist[ib]) { /* contains() / the search loop of insert */
            if (keys[3 * ib] == v[0] && keys[3 * ib + 1] == v[1] && keys[3 * ib + 2] == v[2]) {
                found = 1;
                break;
            }
            ib = (ib + 1) & mask;
            ++d;
        }
        if (found) continue;
        /* (size() < load_threshold = B / 2 >= n always holds here: the grid never grows.  A probe sequence longer
         * than DIST_FROM_IDEAL_BUCKET_LIMIT would make the container grow; it cannot happen at load <= 1/2 with
         * this hash short of an adversarial cloud, and is refused rather than mis-stated.) */
        uint32_t cw = (uint32_t)i;
        int32_t ck[3] = {v[0], v[1], v[2]};
        for (;;) { /* robin-hood placement: take the bucket of an occupant that is STRICTLY closer to its home */
            if (d > 8192) {
                fprintf(stderr, "ko_voxel_downsample: probe sequence beyond tsl's limit (not restated)\n");
                abort();
            }
            if (d > dist[ib]) {
                if (dist[ib] < 0) {
                    dist[ib] = d;
                    who[ib] = cw;
                    keys[3 * ib] = ck[0];
                    keys[3 * ib + 1] = ck[1];
                    keys[3 * ib + 2] = ck[2];
                    break;
                }
                const int32_t td = dist[ib];
                const uint32_t tw = who[ib];
                const int32_t tk[3] = {keys[3 * ib], keys[3 * ib + 1], keys[3 * ib + 2]};
                dist[ib] = d;
                who[ib] = cw;
                keys[3 * ib] = ck[0];
                keys[3 * ib + 1] = ck[1];
                keys[3 * ib + 2] = ck[2];
                d = td;
                cw = tw;
                ck[0] = tk[0];
                ck[1] = tk[1];
                ck[2] = tk[2];
            }
            ++d;
            ib = (ib + 1) & mask;
        }
        ++count;
    }
    size_t kept = 0;
    for (size_t b = 0; b < B; ++b) /* iteration = bucket order */
        if (dist[b] >= 0) {
            const size_t i = who[b];
            out[3 * kept] = xyz[3 * i];
            out[3 * kept + 1] = xyz[3 * i + 1];
            out[3 * kept + 2] = xyz[3 * i + 2];
            ++kept;
        }
    free(dist);
    free(who);
    free(keys);
    (void)count;
    return kept;
}

size_t ko_voxel_downsample(const double *xyz, size_t n, double voxel_size, double *out) {
    /* VoxelUtils.cpp:7-21: insert the first point seen per voxel; emit one point per voxel. */
    if (g_downsample_order == 1) return downsample_tsl_order(xyz, n, voxel_size, out);
    /* emission order: ascending original index */
    vtable t;
    vt_init(&t, 2 * n + 16);
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) {
        int32_t v[3];
        ko_point_to_voxel(xyz + 3 * i, voxel_size, v);
        if (vt_find(&t, v) < 0) {
            vt_insert_nogrow(&t, v, 0);
            out[3 * kept] = xyz[3 * i];
            out[3 * kept + 1] = xyz[3 * i + 1];
            out[3 * kept + 2] = xyz[3 * i + 2];
            ++kept;
        }
    }
    vt_free(&t);
    return kept;
}

/* ======================================================================================
 * VoxelHashMap
 * ==================================================================================== */
struct ko_map {
    double voxel_size, max_distance;
    unsigned max_points;
    vtable table;
    /* dense voxel storage: contiguous per-voxel blocks (== std::vector reserved to
     * max_points_per_voxel, VoxelHashMap.cpp:113-114) */
    int32_t *vkeys;  /* 3 per voxel */
    unsigned *vcount;
    double *vpts;    /* max_points*3 per voxel */
    size_t nvox, vcap;
};

ko_map *ko_map_create(double voxel_size, double max_distance, unsigned max_points_per_voxel) {
    ko_map *m = (ko_map *)calloc(1, sizeof(ko_map));
    m->voxel_size = voxel_size;
    m->max_distance = max_distance;
    m->max_points = max_points_per_voxel;
    vt_init(&m->table, 1024);
    m->vcap = 1024;
    m->vkeys = (int32_t *)malloc(m->vcap * 3 * sizeof(int32_t));
    m->vcount = (unsigned *)malloc(m->vcap * sizeof(unsigned));
    m->vpts = (double *)malloc(m->vcap * (size_t)m->max_points * 3 * sizeof(double));
    m->nvox = 0;
    return m;
}
void ko_map_destroy(ko_map *m) {
    if (!m) return;
    vt_free(&m->table);
    free(m->vkeys);
    free(m->vcount);
    free(m->vpts);
    free(m);
}
void ko_map_clear(ko_map *m) {
    vt_free(&m->table);
    vt_init(&m->table, 1024);
    m->nvox = 0;
}
int ko_map_empty(const ko_map *m) { return m->nvox == 0; }
size_t ko_map_num_voxels(const ko_map *m) { return m->nvox; }
size_t ko_map_num_points(const ko_map *m) {
    size_t s = 0;
    for (size_t i = 0; i < m->nvox; ++i) s += m->vcount[i];
    return s;
}

void ko_map_add_points(ko_map *m, const double *xyz, size_t n) {
    /* VoxelHashMap.cpp:97-119 */
    const double map_resolution =
        sqrt(m->voxel_size * m->voxel_size / (double)m->max_points); /* :98 */
    const size_t stride = (size_t)m->max_points * 3;
    for (size_t i = 0; i < n; ++i) {
        const double *p = xyz + 3 * i;
        int32_t v[3];
        ko_point_to_voxel(p, m->voxel_size, v);
        int32_t idx = vt_find(&m->table, v);
        if (idx >= 0) {
            unsigned cnt = m->vcount[idx];
            double *pts = m->vpts + (size_t)idx * stride;
            if (cnt == m->max_points) continue; /* :104 */
            int too_close = 0;
            for (unsigned k = 0; k < cnt; ++k) { /* :105-108, strict < on norms */
                double d[3] = {pts[3 * k] - p[0], pts[3 * k + 1] - p[1], pts[3 * k + 2] - p[2]};
                if (norm3(d) < map_resolution) {
                    too_close = 1;
                    break;
                }
            }
            if (too_close) continue;
            pts[3 * cnt] = p[0];
            pts[3 * cnt + 1] = p[1];
            pts[3 * cnt + 2] = p[2];
            m->vcount[idx] = cnt + 1;
        } else {
            if (m->nvox == m->vcap) {
                m->vcap *= 2;
                m->vkeys = (int32_t *)realloc(m->vkeys, m->vcap * 3 * sizeof(int32_t));
                m->vcount = (unsigned *)realloc(m->vcount, m->vcap * sizeof(unsigned));
                m->vpts = (double *)realloc(m->vpts, m->vcap * stride * sizeof(double));
            }
            size_t ni = m->nvox++;
            memcpy(m->vkeys + 3 * ni, v, 3 * sizeof(int32_t));
            m->vcount[ni] = 1;
            double *pts = m->vpts + ni * stride;
            pts[0] = p[0];
            pts[1] = p[1];
            pts[2] = p[2];
            vt_insert(&m->table, v, (int32_t)ni);
        }
    }
}

void ko_map_remove_far(ko_map *m, const double origin[3]) {
    /* VoxelHashMap.cpp:121-132: erase a voxel iff its FIRST point is >= max_distance away */
    const double md2 = m->max_distance * m->max_distance;
    const size_t stride = (size_t)m->max_points * 3;
    size_t i = 0;
    while (i < m->nvox) {
        const double *pt = m->vpts + i * stride;
        double d[3] = {pt[0] - origin[0], pt[1] - origin[1], pt[2] - origin[2]};
        if (sqnorm3(d) >= md2) {
            vt_erase(&m->table, m->vkeys + 3 * i);
            size_t last = m->nvox - 1;
            if (i != last) {
                memcpy(m->vkeys + 3 * i, m->vkeys + 3 * last, 3 * sizeof(int32_t));
                m->vcount[i] = m->vcount[last];
                memcpy(m->vpts + i * stride, m->vpts + last * stride, stride * sizeof(double));
                vt_set(&m->table, m->vkeys + 3 * i, (int32_t)i);
            }
            m->nvox--;
        } else {
            ++i;
        }
    }
}

void ko_map_update_origin(ko_map *m, const double *xyz, size_t n, const double origin[3]) {
    /* VoxelHashMap.cpp:83-87 */
    ko_map_add_points(m, xyz, n);
    ko_map_remove_far(m, origin);
}

void ko_map_update_pose(ko_map *m, const double *xyz, size_t n, const double T[16]) {
    /* VoxelHashMap.cpp:89-95 (pybind lambda kiss_icp_pybind.cpp:65-70 builds SE3d(T)) */
    ko_se3 pose;
    ko_se3_from_matrix(T, &pose);
    double *tmp = (double *)malloc((n ? n : 1) * 3 * sizeof(double));
    for (size_t i = 0; i < n; ++i) ko_se3_act(&pose, xyz + 3 * i, tmp + 3 * i);
    ko_map_update_origin(m, tmp, n, pose.t);
    free(tmp);
}

size_t ko_map_pointcloud(const ko_map *m, double *out) {
    /* VoxelHashMap.cpp:72-81 */
    const size_t stride = (size_t)m->max_points * 3;
    size_t k = 0;
    for (size_t i = 0; i < m->nvox; ++i) {
        memcpy(out + 3 * k, m->vpts + i * stride, (size_t)m->vcount[i] * 3 * sizeof(double));
        k += m->vcount[i];
    }
    return k;
}

static const int8_t voxel_shifts[27][3] = {
    /* VoxelHashMap.cpp:35-41, same order: centre, 6 faces, 12 edges, 8 corners */
    {0, 0, 0},   {1, 0, 0},   {-1, 0, 0},  {0, 1, 0},   {0, -1, 0},  {0, 0, 1},   {0, 0, -1},
    {1, 1, 0},   {1, -1, 0},  {-1, 1, 0},  {-1, -1, 0}, {1, 0, 1},   {1, 0, -1},  {-1, 0, 1},
    {-1, 0, -1}, {0, 1, 1},   {0, 1, -1},  {0, -1, 1},  {0, -1, -1}, {1, 1, 1},   {1, 1, -1},
    {1, -1, 1},  {1, -1, -1}, {-1, 1, 1},  {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};

double ko_map_closest_neighbor_counted(const ko_map *m, const double q[3], double nn[3],
                                       uint64_t *examined) {
    /* VoxelHashMap.cpp:46-70 */
    int32_t voxel[3];
    ko_point_to_voxel(q, m->voxel_size, voxel);
    const size_t stride = (size_t)m->max_points * 3;
    nn[0] = nn[1] = nn[2] = 0.0;
    double closest_distance = DBL_MAX;
    uint64_t ex = 0;
    for (int s = 0; s < 27; ++s) {
        int32_t qv[3] = {voxel[0] + voxel_shifts[s][0], voxel[1] + voxel_shifts[s][1],
                         voxel[2] + voxel_shifts[s][2]};
        int32_t idx = vt_find(&m->table, qv);
        if (idx < 0) continue;
        const double *pts = m->vpts + (size_t)idx * stride;
        unsigned cnt = m->vcount[idx];
        ex += cnt;
        /* std::min_element with comparator (lhs-q).norm() < (rhs-q).norm(): first minimum */
        unsigned best = 0;
        double d0[3] = {pts[0] - q[0], pts[1] - q[1], pts[2] - q[2]};
        double bestn = norm3(d0);
        for (unsigned k = 1; k < cnt; ++k) {
            double d[3] = {pts[3 * k] - q[0], pts[3 * k + 1] - q[1], pts[3 * k + 2] - q[2]};
            double nk = norm3(d);
            if (nk < bestn) {
                bestn = nk;
                best = k;
            }
        }
        /* distance = (neighbor - query).norm() -- same expression, same value as bestn */
        if (bestn < closest_distance) { /* :63 strict */
            nn[0] = pts[3 * best];
            nn[1] = pts[3 * best + 1];
            nn[2] = pts[3 * best + 2];
            closest_distance = bestn;
        }
    }
    if (examined) *examined += ex;
    return closest_distance;
}

double ko_map_closest_neighbor(const ko_map *m, const double q[3], double nn[3]) {
    return ko_map_closest_neighbor_counted(m, q, nn, NULL);
}

/* ======================================================================================
 * Registration
 * ==================================================================================== */
typedef struct {
    double JTJ[36];
    double JTr[6];
} linsys;

static inline void linsys_add_corr(linsys *L, const double s[3], const double t[3],
                                   double kernel_scale) {
    /* Registration.cpp:81-88, 96-98, 109-115 */
    const double r[3] = {s[0] - t[0], s[1] - t[1], s[2] - t[2]};
    /* J_r = [ I | -hat(s) ]  (3x6) */
    double J[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
    J[0][3] = -1.0 * 0.0;
    J[0][4] = -1.0 * -s[2];
    J[0][5] = -1.0 * s[1];
    J[1][3] = -1.0 * s[2];
    J[1][4] = -1.0 * 0.0;
    J[1][5] = -1.0 * -s[0];
    J[2][3] = -1.0 * -s[1];
    J[2][4] = -1.0 * s[0];
    J[2][5] = -1.0 * 0.0;
    const double r2 = sqnorm3(r);
    const double ks = kernel_scale;
    const double w = (ks * ks) / ((ks + r2) * (ks + r2)); /* GM_weight :96-98 */
    /* J^T * w * J  and  J^T * w * r  (Eigen evaluates (J^T*w) then the product) */
    for (int a = 0; a < 6; ++a) {
        const double ja0 = J[0][a] * w, ja1 = J[1][a] * w, ja2 = J[2][a] * w;
        for (int b = 0; b < 6; ++b)
            L->JTJ[a * 6 + b] += (ja0 * J[0][b] + ja1 * J[1][b]) + ja2 * J[2][b];
        L->JTr[a] += (ja0 * r[0] + ja1 * r[1]) + ja2 * r[2];
    }
}

void ko_build_linear_system(const double *src, size_t n, const ko_map *m, double max_dist,
                            double kernel_scale, double JTJ[36], double JTr[6],
                            uint64_t *n_corr) {
    linsys L;
    memset(&L, 0, sizeof L);
    uint64_t nc = 0;
    for (size_t i = 0; i < n; ++i) {
        double nn[3];
        double d = ko_map_closest_neighbor(m, src + 3 * i, nn);
        if (d < max_dist) {
            linsys_add_corr(&L, src + 3 * i, nn, kernel_scale);
            ++nc;
        }
    }
    memcpy(JTJ, L.JTJ, sizeof L.JTJ);
    memcpy(JTr, L.JTr, sizeof L.JTr);
    if (n_corr) *n_corr = nc;
}

static void align_core(const double *frame_xyz, size_t n, const ko_map *m, const ko_se3 *guess_in,
                       double max_dist, double kernel_scale, int max_num_iterations,
                       double convergence_criterion, int max_threads, ko_se3 *result,
                       ko_icp_stats *stats) {
    ko_icp_stats st;
    memset(&st, 0, sizeof st);
    st.n_source = n;
    const ko_se3 guess = *guess_in;
    if (ko_map_empty(m)) { /* Registration.cpp:143 */
        *result = guess;
        if (stats) *stats = st;
        return;
    }
    const int nthreads = resolve_threads(max_threads);
    /* :146-147 */
    double *source = (double *)malloc((n ? n : 1) * 3 * sizeof(double));
    for (size_t i = 0; i < n; ++i) ko_se3_act(&guess, frame_xyz + 3 * i, source + 3 * i);
    double *target = (double *)malloc((n ? n : 1) * 3 * sizeof(double));
    unsigned char *has = (unsigned char *)malloc(n ? n : 1);
    linsys *partial = (linsys *)malloc((size_t)nthreads * sizeof(linsys));

    ko_se3 T_icp;
    ko_se3_identity(&T_icp);
    for (int j = 0; j < max_num_iterations; ++j) { /* :151 */
        /* DataAssociation (:60-78) -- parallel_for over points.  Correspondences are kept in
         * source-index order (the reference's concurrent_vector order is nondeterministic). */
        uint64_t examined = 0, ncorr = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : examined, ncorr)
        for (size_t i = 0; i < n; ++i) {
            uint64_t ex = 0;
            double d = ko_map_closest_neighbor_counted(m, source + 3 * i, target + 3 * i, &ex);
            has[i] = (d < max_dist); /* :72 strict */
            examined += ex;
            ncorr += has[i];
        }
        /* BuildLinearSystem (:80-121) -- parallel_reduce; per-thread partials over contiguous
         * static chunks, summed in thread order (deterministic for a fixed thread count). */
        for (int t = 0; t < nthreads; ++t) memset(&partial[t], 0, sizeof(linsys));
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            int tid = omp_get_thread_num();
            int nt = omp_get_num_threads();
#else
            int tid = 0, nt = 1;
#endif
            size_t lo = n * (size_t)tid / (size_t)nt, hi = n * (size_t)(tid + 1) / (size_t)nt;
            linsys L;
            memset(&L, 0, sizeof L);
            for (size_t i = lo; i < hi; ++i)
                if (has[i]) linsys_add_corr(&L, source + 3 * i, target + 3 * i, kernel_scale);
            partial[tid] = L;
        }
        linsys S;
        memset(&S, 0, sizeof S);
        for (int t = 0; t < nthreads; ++t) {
            for (int k = 0; k < 36; ++k) S.JTJ[k] += partial[t].JTJ[k];
            for (int k = 0; k < 6; ++k) S.JTr[k] += partial[t].JTr[k];
        }
        /* :156 dx = JTJ.ldlt().solve(-JTr) */
        double nb[6], dx[6];
        for (int k = 0; k < 6; ++k) nb[k] = -S.JTr[k];
        ko_ldlt6_solve(S.JTJ, nb, dx);
        /* :157 */
        ko_se3 est;
        ko_se3_exp(dx, &est);
        /* :159 TransformPoints(estimation, source) -- serial std::transform */
        for (size_t i = 0; i < n; ++i) {
            double o[3];
            ko_se3_act(&est, source + 3 * i, o);
            source[3 * i] = o[0];
            source[3 * i + 1] = o[1];
            source[3 * i + 2] = o[2];
        }
        /* :161 */
        ko_se3_mul(&est, &T_icp, &T_icp);
        st.iterations = j + 1;
        st.points_examined += examined;
        st.n_corr_last = ncorr;
        st.n_corr_total += ncorr;
        /* :163 dx.norm() < convergence_criterion */
        double nrm = 0.0;
        for (int k = 0; k < 6; ++k) nrm += dx[k] * dx[k];
        if (sqrt(nrm) < convergence_criterion) {
            st.converged = 1;
            break;
        }
    }
    ko_se3_mul(&T_icp, &guess, result); /* :166 */
    free(source);
    free(target);
    free(has);
    free(partial);
    if (stats) *stats = st;
}

int ko_align_points_to_map(const double *frame_xyz, size_t n, const ko_map *m,
                           const double T_guess[16], double max_dist, double kernel_scale,
                           int max_num_iterations, double convergence_criterion,
                           int max_threads, double T_out[16], ko_icp_stats *stats) {
    /* the pybind lambda (kiss_icp_pybind.cpp:94-106): SE3d(T_guess) in, .matrix() out */
    ko_se3 guess, res;
    ko_se3_from_matrix(T_guess, &guess);
    align_core(frame_xyz, n, m, &guess, max_dist, kernel_scale, max_num_iterations,
               convergence_criterion, max_threads, &res, stats);
    ko_se3_matrix(&res, T_out);
    return 0;
}

/* ======================================================================================
 * Preprocessor
 * ==================================================================================== */
static size_t preprocess_core(const double *xyz, size_t n, const double *timestamps, size_t n_ts,
                              const ko_se3 *motion, double max_range, double min_range,
                              int deskew, int max_threads, double *out) {
    /* Preprocessing.cpp:55-95 */
    const double *frame = xyz;
    double *deskewed = NULL;
    if (deskew && n_ts > 0 && timestamps) { /* :59 */
        if (n_ts < n) return (size_t)-1;    /* timestamps.at(idx) would throw (:77) */
        double mn = timestamps[0], mx = timestamps[0];
        for (size_t i = 1; i < n_ts; ++i) { /* :62 minmax over ALL timestamps */
            if (timestamps[i] < mn) mn = timestamps[i];
            if (timestamps[i] > mx) mx = timestamps[i];
        }
        double omega[6];
        ko_se3_log(motion, omega); /* :68 */
        deskewed = (double *)malloc((n ? n : 1) * 3 * sizeof(double));
        const int nthreads = resolve_threads(max_threads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (size_t i = 0; i < n; ++i) { /* :70-81 */
            const double stamp = (timestamps[i] - mn) / (mx - mn);
            double a[6];
            for (int k = 0; k < 6; ++k) a[k] = (stamp - 1.0) * omega[k];
            ko_se3 pose;
            ko_se3_exp(a, &pose);
            ko_se3_act(&pose, xyz + 3 * i, deskewed + 3 * i);
        }
        frame = deskewed;
    }
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) { /* :86-92 serial range filter, order preserving */
        const double r = norm3(frame + 3 * i);
        if (r < max_range && r > min_range) {
            out[3 * kept] = frame[3 * i];
            out[3 * kept + 1] = frame[3 * i + 1];
            out[3 * kept + 2] = frame[3 * i + 2];
            ++kept;
        }
    }
    free(deskewed);
    return kept;
}

size_t ko_preprocess(const double *xyz, size_t n, const double *timestamps, size_t n_ts,
                     const double relative_motion[16], double max_range, double min_range,
                     int deskew, int max_threads, double *out) {
    /* the pybind lambda (kiss_icp_pybind.cpp:80-87): SE3d(relative_motion) */
    ko_se3 motion;
    ko_se3_from_matrix(relative_motion, &motion);
    return preprocess_core(xyz, n, timestamps, n_ts, &motion, max_range, min_range, deskew,
                           max_threads, out);
}

/* ======================================================================================
 * AdaptiveThreshold
 * ==================================================================================== */
void ko_threshold_init(ko_threshold *t, double initial_threshold, double min_motion_threshold,
                       double max_range) {
    /* Threshold.cpp:30-36 */
    t->min_motion_threshold = min_motion_threshold;
    t->max_range = max_range;
    t->model_sse = initial_threshold * initial_threshold;
    t->num_samples = 1;
}
double ko_threshold_compute(const ko_threshold *t) {
    return sqrt(t->model_sse / t->num_samples); /* Threshold.hpp:38 */
}
static void threshold_update_se3(ko_threshold *t, const ko_se3 *dev) {
    /* Threshold.cpp:38-49.  Eigen::AngleAxisd(R).angle(): R -> quaternion (Shepperd) ->
     * angle = 2*atan2(|vec|, |w|)  (Eigen/src/Geometry/AngleAxis.h operator=(Quaternion)). */
    double R[9], q[4];
    quat_to_R(dev->q, R); /* current_deviation.rotationMatrix() */
    R_to_quat(R, q);
    double n = norm3(q);
    double theta = 0.0;
    if (n != 0.0) theta = 2.0 * atan2(n, fabs(q[3]));
    const double delta_rot = 2.0 * t->max_range * sin(theta / 2.0);
    const double delta_trans = norm3(dev->t);
    const double model_error = delta_trans + delta_rot;
    if (model_error > t->min_motion_threshold) {
        t->model_sse += model_error * model_error;
        t->num_samples++;
    }
}
void ko_threshold_update(ko_threshold *t, const double model_deviation[16]) {
    ko_se3 d;
    ko_se3_from_matrix(model_deviation, &d);
    threshold_update_se3(t, &d);
}

/* ======================================================================================
 * pipeline::KissICP
 * ==================================================================================== */
void ko_config_default(ko_config *c) {
    /* KissICP.hpp:36-54 */
    c->voxel_size = 1.0;
    c->max_range = 100.0;
    c->min_range = 0.0;
    c->max_points_per_voxel = 20;
    c->min_motion_th = 0.1;
    c->initial_threshold = 2.0;
    c->max_num_iterations = 500;
    c->convergence_criterion = 0.0001;
    c->max_num_threads = 0;
    c->deskew = 1;
}

struct ko_pipeline {
    ko_config cfg;
    ko_se3 last_pose, last_delta;
    ko_map *map;
    ko_threshold th;
    double *out[3];
    size_t out_n[3], out_cap[3];
    ko_icp_stats stats;
    double sigma;
};

ko_pipeline *ko_pipeline_create(const ko_config *c) {
    ko_pipeline *p = (ko_pipeline *)calloc(1, sizeof(ko_pipeline));
    p->cfg = *c;
    ko_se3_identity(&p->last_pose);
    ko_se3_identity(&p->last_delta);
    /* KissICP.hpp:62-68 */
    p->map = ko_map_create(c->voxel_size, c->max_range, (unsigned)c->max_points_per_voxel);
    ko_threshold_init(&p->th, c->initial_threshold, c->min_motion_th, c->max_range);
    return p;
}
void ko_pipeline_destroy(ko_pipeline *p) {
    if (!p) return;
    ko_map_destroy(p->map);
    for (int i = 0; i < 3; ++i) free(p->out[i]);
    free(p);
}
static double *pipe_buf(ko_pipeline *p, int which, size_t n) {
    if (p->out_cap[which] < n || !p->out[which]) {
        free(p->out[which]);
        p->out_cap[which] = n ? n : 1;
        p->out[which] = (double *)malloc(p->out_cap[which] * 3 * sizeof(double));
    }
    return p->out[which];
}

int ko_pipeline_register_frame(ko_pipeline *p, const double *xyz, size_t n,
                               const double *timestamps, size_t n_ts) {
    /* KissICP.cpp:35-68.  SE3 objects are kept as objects throughout, as the C++ reference does
     * (no matrix round trips). */
    /* :38 */
    double *pre = pipe_buf(p, 0, n);
    size_t n_pre = preprocess_core(xyz, n, timestamps, n_ts, &p->last_delta, p->cfg.max_range,
                                   p->cfg.min_range, p->cfg.deskew, p->cfg.max_num_threads, pre);
    if (n_pre == (size_t)-1) return -1;
    p->out_n[0] = n_pre;
    /* :41, :70-75 Voxelize */
    double *fd = pipe_buf(p, 2, n_pre);
    size_t n_fd = ko_voxel_downsample(pre, n_pre, p->cfg.voxel_size * 0.5, fd);
    p->out_n[2] = n_fd;
    double *src = pipe_buf(p, 1, n_fd);
    size_t n_src = ko_voxel_downsample(fd, n_fd, p->cfg.voxel_size * 1.5, src);
    p->out_n[1] = n_src;
    /* :44 */
    const double sigma = ko_threshold_compute(&p->th);
    p->sigma = sigma;
    /* :47 */
    ko_se3 guess;
    ko_se3_mul(&p->last_pose, &p->last_delta, &guess);
    /* :50-54 */
    ko_se3 new_pose;
    align_core(src, n_src, p->map, &guess, 3.0 * sigma, sigma, p->cfg.max_num_iterations,
               p->cfg.convergence_criterion, p->cfg.max_num_threads, &new_pose, &p->stats);
    /* :57-60 */
    ko_se3 ginv, dev;
    ko_se3_inverse(&guess, &ginv);
    ko_se3_mul(&ginv, &new_pose, &dev);
    threshold_update_se3(&p->th, &dev);
    /* :61 local_map_.Update(frame_downsample, new_pose) (VoxelHashMap.cpp:89-95) */
    {
        double *tmp = (double *)malloc((n_fd ? n_fd : 1) * 3 * sizeof(double));
        for (size_t i = 0; i < n_fd; ++i) ko_se3_act(&new_pose, fd + 3 * i, tmp + 3 * i);
        ko_map_update_origin(p->map, tmp, n_fd, new_pose.t);
        free(tmp);
    }
    /* :62-63 */
    ko_se3 pinv;
    ko_se3_inverse(&p->last_pose, &pinv);
    ko_se3_mul(&pinv, &new_pose, &p->last_delta);
    p->last_pose = new_pose;
    return 0;
}

void ko_pipeline_pose(const ko_pipeline *p, double T[16]) { ko_se3_matrix(&p->last_pose, T); }
void ko_pipeline_delta(const ko_pipeline *p, double T[16]) { ko_se3_matrix(&p->last_delta, T); }
const ko_map *ko_pipeline_map(const ko_pipeline *p) { return p->map; }
size_t ko_pipeline_output_size(const ko_pipeline *p, int which) { return p->out_n[which]; }
void ko_pipeline_output(const ko_pipeline *p, int which, double *out) {
    memcpy(out, p->out[which], p->out_n[which] * 3 * sizeof(double));
}
void ko_pipeline_last_stats(const ko_pipeline *p, ko_icp_stats *s, double *sigma) {
    if (s) *s = p->stats;
    if (sigma) *sigma = p->sigma;
}
