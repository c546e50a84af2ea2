// Host-side checks of kicp_math.hpp (the algebra shared by the HIP kernels and the host side of
// libkicp).  Compiled with hipcc, runs without a GPU: no HIP API is called.
//   1. voxel_coord_fast == voxel_coord (PointToVoxel, core/VoxelUtils.hpp:33-37) bit for bit, on
//      multiples of the voxel size +- a few ulp and on random values, for a range of voxel sizes;
//   2. pack_voxel / unpack_voxel round trip at the range limits;
//   3. se3_exp / se3_log round trip and se3_mul / se3_inverse consistency;
//   4. ldlt6_solve on a well-conditioned SPD system, a permuted diagonal (exercises every pivot
//      choice) and the zero matrix (Eigen's zero-pivot rule: zero solution).
#include <cmath>
#include <cstdio>
#include <random>

#include "../../kiss-icp_amd/csrc/kicp_math.hpp"

using namespace kicp;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            ++failures;                                                    \
            std::printf("FAILED %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
        }                                                                  \
    } while (0)

int main() {
    std::mt19937_64 gen(7);
    // 1 ---------------------------------------------------------------------------------------
    long checked = 0, mismatches = 0;
    const double sizes[] = {1.0, 0.5, 0.1, 1.5, 0.05, 0.15, 0.3, 3.0, 0.7, 1e-3, 0.25, 2.0 / 3.0};
    for (double vs : sizes) {
        const double inv = 1.0 / vs;
        for (int k = -200000; k <= 200000; k += 37)
            for (int d = -4; d <= 4; ++d) {
                double p = k * vs;
                for (int t = 0; t < std::abs(d); ++t) p = std::nextafter(p, d > 0 ? INFINITY : -INFINITY);
                ++checked;
                mismatches += voxel_coord(p, vs) != voxel_coord_fast(p, vs, inv);
            }
        std::uniform_real_distribution<double> u(-1000.0, 1000.0);
        for (int i = 0; i < 500000; ++i) {
            const double p = u(gen);
            ++checked;
            mismatches += voxel_coord(p, vs) != voxel_coord_fast(p, vs, inv);
        }
    }
    std::printf("voxel_coord_fast: %ld inputs, %ld mismatches\n", checked, mismatches);
    CHECK(mismatches == 0);
    CHECK(voxel_coord_fast(-0.0, 1.0, 1.0) == 0 && voxel_coord_fast(-1e-12, 1.0, 1.0) == -1);
    // 2 ---------------------------------------------------------------------------------------
    for (int x : {-(kVoxelLimit - 1), -1, 0, 1, kVoxelLimit - 1})
        for (int y : {-(kVoxelLimit - 1), 0, kVoxelLimit - 1})
            for (int z : {-(kVoxelLimit - 1), -7, kVoxelLimit - 1}) {
                int a, b, c;
                const uint64_t key = pack_voxel(x, y, z);
                unpack_voxel(key, a, b, c);
                CHECK(a == x && b == y && c == z);
                CHECK(key != kKeyEmpty && key != kKeyTomb && (key >> 63) == 0);
                CHECK(voxel_in_range(x, y, z));
            }
    CHECK(!voxel_in_range(kVoxelLimit, 0, 0) && !voxel_in_range(0, -kVoxelLimit, 0));
    // 3 ---------------------------------------------------------------------------------------
    std::normal_distribution<double> n01(0.0, 1.0);
    for (int i = 0; i < 2000; ++i) {
        const double scale = (i % 4 == 0) ? 1e-12 : (i % 4 == 1 ? 1e-3 : 0.7);
        double a[6], b[6];
        for (int k = 0; k < 3; ++k) a[k] = 3.0 * n01(gen);
        for (int k = 3; k < 6; ++k) a[k] = scale * n01(gen);
        const SE3 T = se3_exp(a);
        se3_log(T, b);
        for (int k = 0; k < 6; ++k) CHECK(std::fabs(a[k] - b[k]) < 1e-9 * (1.0 + std::fabs(a[k])));
        const SE3 I = se3_mul(T, se3_inverse(T));
        CHECK(std::fabs(I.q[3] - 1.0) < 1e-12 && std::fabs(I.t[0]) + std::fabs(I.t[1]) + std::fabs(I.t[2]) < 1e-9);
        const double p[3] = {n01(gen), n01(gen), n01(gen)};
        double q[3], r[3];
        se3_act(T, p, q);
        se3_act(se3_inverse(T), q, r);
        for (int k = 0; k < 3; ++k) CHECK(std::fabs(r[k] - p[k]) < 1e-9);
    }
    // 4 ---------------------------------------------------------------------------------------
    for (int trial = 0; trial < 200; ++trial) {
        double M[36], A[36], x_true[6], b[6], x[6];
        for (double &m : M) m = n01(gen);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = (i == j) ? 0.5 : 0.0;
                for (int k = 0; k < 6; ++k) s += M[i * 6 + k] * M[j * 6 + k];
                A[i * 6 + j] = s;
            }
        for (int i = 0; i < 6; ++i) x_true[i] = n01(gen);
        for (int i = 0; i < 6; ++i) {
            b[i] = 0.0;
            for (int j = 0; j < 6; ++j) b[i] += A[i * 6 + j] * x_true[j];
        }
        ldlt6_solve(A, b, x);
        for (int i = 0; i < 6; ++i) CHECK(std::fabs(x[i] - x_true[i]) < 1e-8);
    }
    {
        double A[36] = {0}, b[6], x[6];
        const double diag[6] = {3.0, 100.0, 0.5, 7.0, 42.0, 1.0};  // every pivot step has to swap
        for (int i = 0; i < 6; ++i) {
            A[i * 6 + i] = diag[i];
            b[i] = diag[i] * (i + 1);
        }
        ldlt6_solve(A, b, x);
        for (int i = 0; i < 6; ++i) CHECK(std::fabs(x[i] - (i + 1)) < 1e-12);
        double Z[36] = {0}, bz[6] = {1, 2, 3, 4, 5, 6};
        ldlt6_solve(Z, bz, x);  // empty correspondence set: JTJ = 0 -> dx = 0 (Registration.cpp:156)
        for (int i = 0; i < 6; ++i) CHECK(x[i] == 0.0);
    }
    // 5. schur3_solve: the Gauss-Newton step of Registration.cpp:80-121,156 through the 3 x 3 Schur complement of its
    //    normal equations -- against ldlt6_solve on the SAME sixteen sums, formed as k_icp's phase C forms them, for
    //    clouds of every shape an odometry frame takes (room, corridor, slab; tens to thousands of correspondences,
    //    coordinates up to a kilometre from the origin); and the systems it must REFUSE: no correspondence, one point,
    //    points on a line through the origin (rank 4) -- those go to the LDLT and its zero-pivot rule.
    {
        auto sums = [&](int n, const double ext[3], const double off[3], double noise, double S[16]) {
            for (int k = 0; k < 16; ++k) S[k] = 0.0;
            for (int i = 0; i < n; ++i) {
                const double s[3] = {off[0] + ext[0] * n01(gen), off[1] + ext[1] * n01(gen), off[2] + ext[2] * n01(gen)};
                const double r[3] = {noise * n01(gen), noise * n01(gen), noise * n01(gen)};
                const double r2 = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2], ks = 0.3;
                const double w = (ks * ks) / ((ks + r2) * (ks + r2));
                const double T[16] = {w, w * s[0], w * s[1], w * s[2], w * (s[1] * s[1] + s[2] * s[2]), w * (-(s[0] * s[1])), w * (-(s[0] * s[2])),
                                      w * (s[0] * s[0] + s[2] * s[2]), w * (-(s[1] * s[2])), w * (s[0] * s[0] + s[1] * s[1]), w * r[0], w * r[1], w * r[2],
                                      w * (s[1] * r[2] - s[2] * r[1]), w * (s[2] * r[0] - s[0] * r[2]), w * (s[0] * r[1] - s[1] * r[0])};
                for (int k = 0; k < 16; ++k) S[k] += T[k];
            }
        };
        auto by_ldlt = [&](const double S[16], double x[6]) {
            double J[36] = {0}, nb[6];
            J[0] = J[7] = J[14] = S[0];
            J[0 * 6 + 4] = J[4 * 6 + 0] = S[3];
            J[0 * 6 + 5] = J[5 * 6 + 0] = -S[2];
            J[1 * 6 + 3] = J[3 * 6 + 1] = -S[3];
            J[1 * 6 + 5] = J[5 * 6 + 1] = S[1];
            J[2 * 6 + 3] = J[3 * 6 + 2] = S[2];
            J[2 * 6 + 4] = J[4 * 6 + 2] = -S[1];
            J[3 * 6 + 3] = S[4];
            J[3 * 6 + 4] = J[4 * 6 + 3] = S[5];
            J[3 * 6 + 5] = J[5 * 6 + 3] = S[6];
            J[4 * 6 + 4] = S[7];
            J[4 * 6 + 5] = J[5 * 6 + 4] = S[8];
            J[5 * 6 + 5] = S[9];
            for (int i = 0; i < 6; ++i) nb[i] = -S[10 + i];
            ldlt6_solve(J, nb, x);
        };
        const double shapes[4][3] = {{20, 20, 3}, {60, 4, 2}, {30, 30, 0.05}, {2, 2, 2}};
        int taken = 0, refused = 0;
        double worst = 0.0;
        for (int trial = 0; trial < 4000; ++trial) {
            const double *ext = shapes[trial & 3];
            const double far = (trial % 5 == 0) ? 1000.0 : ((trial % 7 == 0) ? 100.0 : 0.0);
            const double off[3] = {far * n01(gen), far * n01(gen), 0.1 * far * n01(gen)};
            const int n = 30 + (trial * 37) % 4000;
            double S[16], xs[6], xl[6];
            sums(n, ext, off, (trial & 8) ? 0.3 : 0.02, S);
            by_ldlt(S, xl);
            if (!schur3_solve(S, xs)) {  // (a slab seen from a kilometre away may be refused: that is what the guard is for)
                ++refused;
                continue;
            }
            ++taken;
            double nl = 0.0, nd = 0.0;
            for (int i = 0; i < 6; ++i) {
                nl += xl[i] * xl[i];
                nd += (xs[i] - xl[i]) * (xs[i] - xl[i]);
            }
            const double rel = std::sqrt(nd) / std::fmax(std::sqrt(nl), 1e-300);
            worst = std::fmax(worst, rel);
            // two stable eliminations of the same SPD system: they differ by rounding amplified by the condition number; the
            // guard (pivots above 1e-9 of the largest diagonal entry) keeps that far below the convergence threshold's 1e-4
            CHECK(rel < 1e-6);
        }
        std::printf("schur3_solve: %d systems taken (largest relative difference to the LDLT %.3g), %d refused\n", taken, worst, refused);
        CHECK(taken > 3000);
        double S[16] = {0}, x[6] = {9, 9, 9, 9, 9, 9};
        CHECK(!schur3_solve(S, x) && x[0] == 9);  // no correspondence
        const double one_ext[3] = {0, 0, 0}, one_off[3] = {3, -2, 1};
        sums(1, one_ext, one_off, 0.1, S);
        CHECK(!schur3_solve(S, x));  // one point: the rotation about the ray to it is free
        for (int k = 0; k < 16; ++k) S[k] = 0.0;
        for (int i = 1; i <= 50; ++i) {  // points on a line through the origin: rank 4
            const double s[3] = {0.3 * i, -0.2 * i, 0.1 * i}, r[3] = {0.01, -0.02, 0.005}, w = 1.0;
            const double T[16] = {w, w * s[0], w * s[1], w * s[2], w * (s[1] * s[1] + s[2] * s[2]), w * (-(s[0] * s[1])), w * (-(s[0] * s[2])),
                                  w * (s[0] * s[0] + s[2] * s[2]), w * (-(s[1] * s[2])), w * (s[0] * s[0] + s[1] * s[1]), w * r[0], w * r[1], w * r[2],
                                  w * (s[1] * r[2] - s[2] * r[1]), w * (s[2] * r[0] - s[0] * r[2]), w * (s[0] * r[1] - s[1] * r[0])};
            for (int k = 0; k < 16; ++k) S[k] += T[k];
        }
        CHECK(!schur3_solve(S, x));
    }
    std::printf(failures ? "%d check(s) FAILED\n" : "all checks passed\n", failures);
    return failures != 0;
}
