// metrics.cpp -- kiss_icp::metrics::{SeqError, AbsoluteTrajectoryError}, restated from the
// behaviour of cpp/kiss_icp/metrics/Metrics.cpp:35-189 (itself the KITTI dev-kit's evaluation) with
// a small self-contained 4x4 / 3x3 algebra (no Eigen in this build image): general 4x4 inverse,
// one-sided Jacobi SVD of the 3x3 cross-covariance for the Umeyama alignment (Eigen::umeyama with
// with_scaling = false), Eigen::AngleAxisd's angle of a rotation matrix.
#include "kiss_icp/metrics/Metrics.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <stdexcept>

namespace {

using Mat4 = std::array<double, 16>;  // row-major
using Mat3 = std::array<double, 9>;   // row-major

Mat4 to_rows(const Eigen::Matrix4d &M) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r[i * 4 + j] = M(i, j);
    return r;
}

Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 c{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            c[i * 4 + j] = s;
        }
    return c;
}

// general inverse (the reference calls Eigen's Matrix4d::inverse(), not a rigid shortcut):
// Gauss-Jordan with partial pivoting
Mat4 inverse(const Mat4 &m) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = m[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) throw std::invalid_argument("metrics: singular pose matrix");
        if (p != c)
            for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    Mat4 inv;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
    return inv;
}

double det3(const Mat3 &m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// A = U diag(s) V^T by one-sided (Hestenes) Jacobi: rotate pairs of columns of W = A V until they
// are orthogonal; then s_j = |w_j|, u_j = w_j / s_j.  Singular values sorted descending like Eigen's
// JacobiSVD; null directions of U completed to a right-handed orthonormal basis.
void svd3(const Mat3 &A, Mat3 &U, std::array<double, 3> &S, Mat3 &V) {
    double W[3][3], Vm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W[i][j] = A[i * 3 + j];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += W[i][p] * W[i][p];
                    beta += W[i][q] * W[i][q];
                    gamma += W[i][p] * W[i][q];
                }
                if (gamma == 0.0) continue;
                off = std::max(off, std::fabs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double wp = W[i][p], wq = W[i][q];
                    W[i][p] = c * wp - s * wq;
                    W[i][q] = s * wp + c * wq;
                    const double vp = Vm[i][p], vq = Vm[i][q];
                    Vm[i][p] = c * vp - s * vq;
                    Vm[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    int order[3] = {0, 1, 2};
    double norms[3];
    for (int j = 0; j < 3; ++j) norms[j] = std::sqrt(W[0][j] * W[0][j] + W[1][j] * W[1][j] + W[2][j] * W[2][j]);
    std::sort(order, order + 3, [&](int a, int b) { return norms[a] > norms[b]; });
    double Um[3][3];
    const double tiny = 1e-14 * std::max(norms[order[0]], 1e-300);
    int rank = 0;
    for (int k = 0; k < 3; ++k) {
        const int j = order[k];
        S[k] = norms[j];
        for (int i = 0; i < 3; ++i) V[i * 3 + k] = Vm[i][j];
        if (norms[j] > tiny) {
            for (int i = 0; i < 3; ++i) Um[i][k] = W[i][j] / norms[j];
            rank = k + 1;
        }
    }
    // complete U (rank-deficient cross-covariance: straight or planar trajectories)
    auto cross = [&](int a, int b, int c) {
        Um[0][c] = Um[1][a] * Um[2][b] - Um[2][a] * Um[1][b];
        Um[1][c] = Um[2][a] * Um[0][b] - Um[0][a] * Um[2][b];
        Um[2][c] = Um[0][a] * Um[1][b] - Um[1][a] * Um[0][b];
    };
    if (rank == 0) {
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) Um[i][k] = (i == k) ? 1.0 : 0.0;
    } else if (rank == 1) {
        // any unit vector orthogonal to u0
        int m = 0;
        for (int i = 1; i < 3; ++i)
            if (std::fabs(Um[i][0]) < std::fabs(Um[m][0])) m = i;
        double e[3] = {0, 0, 0};
        e[m] = 1.0;
        const double d = Um[m][0];
        double n = 0;
        for (int i = 0; i < 3; ++i) {
            Um[i][1] = e[i] - d * Um[i][0];
            n += Um[i][1] * Um[i][1];
        }
        n = std::sqrt(n);
        for (int i = 0; i < 3; ++i) Um[i][1] /= n;
        cross(0, 1, 2);
    } else if (rank == 2) {
        cross(0, 1, 2);
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) U[i * 3 + k] = Um[i][k];
}

// Eigen::AngleAxisd(R).angle(): R -> quaternion (Shepperd), angle = 2 atan2(|vec|, |w|)
double rotation_angle(const Mat4 &T) {
    const double m00 = T[0], m11 = T[5], m22 = T[10];
    double q[4];  // x y z w
    double t = m00 + m11 + m22;
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (T[9] - T[6]) * t;
        q[1] = (T[2] - T[8]) * t;
        q[2] = (T[4] - T[1]) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > T[i * 4 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(T[i * 4 + i] - T[j * 4 + j] - T[k * 4 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (T[k * 4 + j] - T[j * 4 + k]) * t;
        q[j] = (T[j * 4 + i] + T[i * 4 + j]) * t;
        q[k] = (T[k * 4 + i] + T[i * 4 + k]) * t;
    }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    return n != 0.0 ? 2.0 * std::atan2(n, std::fabs(q[3])) : 0.0;
}

}  // namespace

namespace kiss_icp::metrics {

std::tuple<float, float> SeqError(const std::vector<Eigen::Matrix4d> &poses_gt,
                                  const std::vector<Eigen::Matrix4d> &poses_result) {
    if (poses_result.size() < poses_gt.size()) throw std::invalid_argument("SeqError: fewer result poses than ground truth");
    static const double kLengths[] = {100, 200, 300, 400, 500, 600, 700, 800};
    const size_t n = poses_gt.size();
    std::vector<Mat4> gt(n), res(n);
    for (size_t i = 0; i < n; ++i) {
        gt[i] = to_rows(poses_gt[i]);
        res[i] = to_rows(poses_result[i]);
    }
    // distance travelled along the ground truth
    std::vector<double> dist(n ? n : 1, 0.0);
    for (size_t i = 1; i < n; ++i) {
        const double dx = gt[i - 1][3] - gt[i][3], dy = gt[i - 1][7] - gt[i][7], dz = gt[i - 1][11] - gt[i][11];
        dist[i] = dist[i - 1] + std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    double t_sum = 0.0, r_sum = 0.0;
    size_t count = 0;
    for (size_t first = 0; first < n; first += 10) {  // "every second" at 10 Hz
        for (double len : kLengths) {
            size_t last = n;
            for (size_t i = first; i < n; ++i)
                if (dist[i] > dist[first] + len) {
                    last = i;
                    break;
                }
            if (last == n) continue;  // sequence not long enough
            const Mat4 delta_gt = mul(inverse(gt[first]), gt[last]);
            const Mat4 delta_res = mul(inverse(res[first]), res[last]);
            const Mat4 err = mul(inverse(delta_res), delta_gt);
            const double d = 0.5 * (err[0] + err[5] + err[10] - 1.0);
            const double r_err = std::acos(std::max(std::min(d, 1.0), -1.0));
            const double t_err = std::sqrt(err[3] * err[3] + err[7] * err[7] + err[11] * err[11]);
            r_sum += r_err / len;
            t_sum += t_err / len;
            ++count;
        }
    }
    // (0 / 0 -> NaN for sequences shorter than 100 m, like the reference; note its 3.14)
    const double avg_t = 100.0 * (t_sum / static_cast<double>(count));
    const double avg_r = (r_sum / static_cast<double>(count)) / 3.14 * 180.0;
    return std::make_tuple(static_cast<float>(avg_t), static_cast<float>(avg_r));
}

std::tuple<float, float> AbsoluteTrajectoryError(const std::vector<Eigen::Matrix4d> &poses_gt,
                                                 const std::vector<Eigen::Matrix4d> &poses_result) {
    if (poses_gt.size() != poses_result.size())
        throw std::invalid_argument("AbsoluteTrajectoryError: different number of poses in ground truth and estimate");
    const size_t n = poses_gt.size();
    std::vector<Mat4> gt(n), res(n);
    double mu_s[3] = {0, 0, 0}, mu_t[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        gt[i] = to_rows(poses_gt[i]);
        res[i] = to_rows(poses_result[i]);
        for (int k = 0; k < 3; ++k) {
            mu_s[k] += res[i][k * 4 + 3];
            mu_t[k] += gt[i][k * 4 + 3];
        }
    }
    for (int k = 0; k < 3; ++k) {
        mu_s[k] /= static_cast<double>(n);
        mu_t[k] /= static_cast<double>(n);
    }
    // Umeyama without scaling: sigma = 1/n sum (y - mu_y)(x - mu_x)^T = U D V^T, R = U S V^T with
    // S = diag(1, 1, sign(det U det V)), t = mu_y - R mu_x
    Mat3 sigma{};
    for (size_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                sigma[r * 3 + c] += (gt[i][r * 4 + 3] - mu_t[r]) * (res[i][c * 4 + 3] - mu_s[c]);
    for (double &v : sigma) v /= static_cast<double>(n);
    Mat3 U, V;
    std::array<double, 3> S;
    svd3(sigma, U, S, V);
    const double sgn = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
    Mat4 align{};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            align[r * 4 + c] = U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1] + sgn * U[r * 3 + 2] * V[c * 3 + 2];
    for (int r = 0; r < 3; ++r)
        align[r * 4 + 3] = mu_t[r] - (align[r * 4 + 0] * mu_s[0] + align[r * 4 + 1] * mu_s[1] + align[r * 4 + 2] * mu_s[2]);
    align[15] = 1.0;
    double rot2 = 0.0, trans2 = 0.0;
    for (size_t j = 0; j < n; ++j) {
        const Mat4 delta = mul(inverse(mul(align, res[j])), gt[j]);
        const double theta = rotation_angle(delta);
        rot2 += theta * theta;
        trans2 += delta[3] * delta[3] + delta[7] * delta[7] + delta[11] * delta[11];
    }
    rot2 /= static_cast<double>(n);
    trans2 /= static_cast<double>(n);
    return std::make_tuple(static_cast<float>(std::sqrt(rot2)), static_cast<float>(std::sqrt(trans2)));
}

}  // namespace kiss_icp::metrics
