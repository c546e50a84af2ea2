// test_cpp_api.cpp -- GPU test of the C++ drop-in surface (kiss-icp_amd/cpp/include/kiss_icp/...):
// code written against the reference's headers -- kiss_icp::pipeline::KissICP::RegisterFrame,
// kiss_icp::Registration::AlignPointsToMap, kiss_icp::VoxelHashMap, VoxelDownsample, Preprocessor --
// compiled unchanged, running on the HIP path, checked against the CPU oracle (oracle/kiss_oracle.h;
// test infrastructure, linked only into this test).  Run by tests/test_cpp_api.py on the GPU box.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "kicp.h"
#include "kiss_icp/pipeline/KissICP.hpp"
#include "kiss_oracle.h"

using Points = std::vector<Eigen::Vector3d>;

static int g_fail = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);    \
            ++g_fail;                                                      \
        }                                                                  \
    } while (0)

// elapsed time at every section (a fresh box pages the ROCm libraries in from its image: the first process can spend
// minutes in one of them -- this says in which)
static void lap(const char *what) {
    static const auto t0 = std::chrono::steady_clock::now();
    std::printf("[%8.2f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what);
    std::fflush(stdout);
}

static unsigned long long g_seed = 88172645463325252ull;
static double urand() {  // xorshift64*, deterministic everywhere
    g_seed ^= g_seed >> 12;
    g_seed ^= g_seed << 25;
    g_seed ^= g_seed >> 27;
    return (double)((g_seed * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}
static double uni(double a, double b) { return a + (b - a) * urand(); }

static Points scene(int n) {  // floor + two walls, a few centimetres of roughness
    Points p;
    for (int i = 0; i < n; ++i) p.emplace_back(uni(-30, 30), uni(-30, 30), uni(-0.02, 0.02));
    for (int i = 0; i < n / 2; ++i) p.emplace_back(14.0 + uni(-0.02, 0.02), uni(-30, 30), uni(0, 6));
    for (int i = 0; i < n / 2; ++i) p.emplace_back(uni(-30, 30), -11.0 + uni(-0.02, 0.02), uni(0, 6));
    return p;
}

static Sophus::SE3d pose_of(double x, double y, double yaw) {
    Eigen::Matrix4d T;
    T(0, 0) = std::cos(yaw);
    T(0, 1) = -std::sin(yaw);
    T(1, 0) = std::sin(yaw);
    T(1, 1) = std::cos(yaw);
    T(0, 3) = x;
    T(1, 3) = y;
    return Sophus::SE3d(T);
}

static void rowmajor(const Sophus::SE3d &T, double out[16]) { kiss_icp::detail::se3_to_rowmajor(T, out); }

static double pose_diff(const double a[16], const double b[16]) {
    double d = 0;
    for (int i = 0; i < 16; ++i) d = std::fmax(d, std::fabs(a[i] - b[i]));
    return d;
}

// everything but the batch entry
static void run_front() {
    const double *X;
    lap("start");
    // ---- VoxelDownsample / VoxelHashMap -------------------------------------------------------------
    Points world = scene(20000);
    X = reinterpret_cast<const double *>(world.data());
    {
        Points ds = kiss_icp::VoxelDownsample(world, 0.5);
        std::vector<double> ref(world.size() * 3);
        const size_t n = ko_voxel_downsample(X, world.size(), 0.5, ref.data());
        CHECK(ds.size() == n);
        bool same = ds.size() == n;
        for (size_t i = 0; same && i < n; ++i)
            same = ds[i][0] == ref[3 * i] && ds[i][1] == ref[3 * i + 1] && ds[i][2] == ref[3 * i + 2];
        CHECK(same);
        const kiss_icp::Voxel v = kiss_icp::PointToVoxel(Eigen::Vector3d(-0.3, 0.6, 99.99999999999999), 0.1);
        CHECK(v[0] == -3 && v[1] == 5 && v[2] == 999);
    }
    kiss_icp::VoxelHashMap map(1.0, 100.0, 20);
    ko_map *omap = ko_map_create(1.0, 100.0, 20);
    CHECK(map.Empty());
    map.AddPoints(world);
    ko_map_add_points(omap, X, world.size());
    CHECK(!map.Empty());
    CHECK(map.NumVoxels() == ko_map_num_voxels(omap));
    CHECK(map.Pointcloud().size() == ko_map_num_points(omap));
    {
        Points q;
        for (int i = 0; i < 500; ++i) q.emplace_back(uni(-32, 32), uni(-32, 32), uni(-1, 7));
        auto res = map.GetClosestNeighbors(q);
        int bad = 0;
        for (size_t i = 0; i < q.size(); ++i) {
            double nn[3];
            const double d = ko_map_closest_neighbor(omap, q[i].data(), nn);
            const auto &[p, dist] = res[i];
            if (!(dist == d && p[0] == nn[0] && p[1] == nn[1] && p[2] == nn[2])) ++bad;
        }
        CHECK(bad == 0);
        const auto [p1, d1] = map.GetClosestNeighbor(Eigen::Vector3d(500, 500, 500));
        CHECK(d1 == 1.7976931348623157e308 && p1[0] == 0.0);
    }
    {
        // value semantics (the reference's VoxelHashMap is copyable, VoxelHashMap.hpp:38-57): a snapshot is a second,
        // independent map -- same content now, untouched by what happens to the original afterwards
        kiss_icp::VoxelHashMap snapshot = map;
        CHECK(snapshot.NumVoxels() == map.NumVoxels());
        CHECK(snapshot.Pointcloud().size() == map.Pointcloud().size());
        const Eigen::Vector3d probe(3.3, -2.1, 0.4);
        const auto [pa, da] = map.GetClosestNeighbor(probe);
        const auto [pb, db] = snapshot.GetClosestNeighbor(probe);
        CHECK(da == db && pa[0] == pb[0] && pa[1] == pb[1] && pa[2] == pb[2]);
        const std::size_t before = snapshot.NumVoxels();
        kiss_icp::VoxelHashMap assigned(2.0, 50.0, 5);
        assigned = map;  // copy assignment replaces parameters and content
        CHECK(assigned.voxel_size_ == map.voxel_size_ && assigned.NumVoxels() == map.NumVoxels());
        Points far;
        for (int i = 0; i < 300; ++i) far.emplace_back(200.0 + uni(0, 20), uni(-5, 5), uni(0, 2));
        snapshot.AddPoints(far);
        CHECK(snapshot.NumVoxels() > before && map.NumVoxels() == before && assigned.NumVoxels() == before);
        snapshot.Clear();
        CHECK(snapshot.Empty() && !map.Empty());
    }

    lap("VoxelDownsample / VoxelHashMap done");
    // ---- Registration::AlignPointsToMap ---------------------------------------------------------------
    {
        const Sophus::SE3d T_true = pose_of(0.3, -0.2, 0.02);
        const Sophus::SE3d inv = T_true.inverse();
        Points frame;
        Points sample = scene(1500);
        for (auto &p : sample) frame.push_back(inv * p);
        kiss_icp::Registration reg(500, 1e-4, 0);
        const Sophus::SE3d guess = pose_of(0.05, 0.0, 0.0);
        const Sophus::SE3d T = reg.AlignPointsToMap(frame, map, guess, 3.0, 1.0);
        double Tg[16], To[16], G[16];
        rowmajor(T, Tg);
        rowmajor(guess, G);
        ko_icp_stats st;
        ko_align_points_to_map(reinterpret_cast<const double *>(frame.data()), frame.size(), omap, G, 3.0, 1.0, 500,
                               1e-4, 0, To, &st);
        CHECK(pose_diff(Tg, To) < 1e-7);
        CHECK(reg.last_iterations_ == st.iterations);
        CHECK(std::fabs(Tg[3] - 0.3) < 0.05 && std::fabs(Tg[7] + 0.2) < 0.05);
        // the reference's Registration is a copyable struct (core/Registration.hpp:33-45): a copy registers like the original
        kiss_icp::Registration reg2 = reg;
        kiss_icp::Registration reg3(1, 1.0, 0);
        reg3 = reg;
        double T2[16], T3[16];
        rowmajor(reg2.AlignPointsToMap(frame, map, guess, 3.0, 1.0), T2);
        rowmajor(reg3.AlignPointsToMap(frame, map, guess, 3.0, 1.0), T3);
        CHECK(pose_diff(T2, Tg) == 0.0 && pose_diff(T3, Tg) == 0.0);
        CHECK(reg2.max_num_iterations_ == 500 && reg3.convergence_criterion_ == 1e-4);
        // empty map -> the guess comes back (Registration.cpp:143)
        kiss_icp::VoxelHashMap empty(1.0, 100.0, 20);
        double Te[16];
        rowmajor(reg.AlignPointsToMap(frame, empty, guess, 3.0, 1.0), Te);
        CHECK(pose_diff(Te, G) < 1e-15);
    }

    lap("AlignPointsToMap done");
    // ---- pipeline::KissICP::RegisterFrame over a short drive ---------------------------------------------
    {
        kiss_icp::pipeline::KISSConfig cfg;
        cfg.deskew = false;
        kiss_icp::pipeline::KissICP odom(cfg);
        ko_config oc;
        ko_config_default(&oc);
        oc.deskew = 0;
        ko_pipeline *op = ko_pipeline_create(&oc);
        double worst = 0;
        for (int k = 0; k < 8; ++k) {
            const Sophus::SE3d inv = pose_of(0.8 * k, 0.05 * k, 0.01 * k).inverse();
            Points frame;
            Points sample = scene(6000);
            for (auto &p : sample) frame.push_back(inv * p);
            const auto &[pre, src] = odom.RegisterFrame(frame, {});
            ko_pipeline_register_frame(op, reinterpret_cast<const double *>(frame.data()), frame.size(), nullptr, 0);
            CHECK(pre.size() == ko_pipeline_output_size(op, 0));
            CHECK(src.size() == ko_pipeline_output_size(op, 1));
            double Tg[16], To[16];
            rowmajor(odom.pose(), Tg);
            ko_pipeline_pose(op, To);
            worst = std::fmax(worst, pose_diff(Tg, To));
        }
        std::printf("pipeline: worst |T_gpu - T_oracle| over 8 frames = %.3e\n", worst);
        CHECK(worst < 1e-7);
        CHECK(odom.LocalMap().size() == ko_map_num_points(ko_pipeline_map(op)));
        CHECK(odom.VoxelMap().NumVoxels() == ko_map_num_voxels(ko_pipeline_map(op)));
        CHECK(odom.pose().translation()[0] > 4.0);  // it drove
        // pose() is a mutable reference: an edit is honoured by the next frame
        const auto [s2, fd2] = odom.Voxelize(scene(3000));
        CHECK(!s2.empty() && fd2.size() >= s2.size());
        ko_pipeline_destroy(op);
    }

    lap("RegisterFrame drive done");
    // ---- error conventions ---------------------------------------------------------------------------------
    {
        bool threw = false;
        try {
            Eigen::Matrix4d bad;
            bad(0, 0) = 2.0;
            Sophus::SE3d T(bad);
            (void)T;
        } catch (const std::exception &) {
            threw = true;
        }
        CHECK(threw || KISS_ICP_HIP_HAVE_EIGEN);  // real Sophus aborts instead (SOPHUS_ENSURE)
        threw = false;
        try {
            kiss_icp::Preprocessor pre(100.0, 0.0, true, 0);
            Points f = scene(100);
            pre.Preprocess(f, std::vector<double>(10, 0.0), Sophus::SE3d());
        } catch (const std::out_of_range &) {
            threw = true;
        }
        CHECK(threw);
        kiss_icp::Preprocessor pre(20.0, 1.0, false, 0);
        Points f = scene(2000);
        std::vector<double> ref(f.size() * 3);
        double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const size_t n =
            ko_preprocess(reinterpret_cast<const double *>(f.data()), f.size(), nullptr, 0, I, 20.0, 1.0, 0, 1, ref.data());
        CHECK(pre.Preprocess(f, {}, Sophus::SE3d()).size() == n);
    }
    ko_map_destroy(omap);
    lap("error conventions done");
}

// the multi-stream batch entry (its first use on a machine loads RCCL: 570 MB of compressed code objects, read in full)
static void run_batch() {
    // ---- multi-stream batch entry of the C-ABI from C++ (no Python, no torch): one stream on GPU 0, poses exchanged by
    // RCCL called directly by the library; frames queued four deep, poses against the oracle's --------------------------
    {
        kicp_config kc;
        kicp_config_default(&kc);
        kc.deskew = 0;
        const int devices[1] = {0};
        kicp_batch *b = nullptr;
        const int rc = kicp_batch_create(&kc, devices, 1, 0, 1, nullptr, nullptr, 4, &b);
        if (rc != KICP_OK) std::printf("kicp_batch_create: %s\n", kicp_last_error());
        CHECK(rc == KICP_OK);
        if (rc == KICP_OK) {
            ko_config oc;
            ko_config_default(&oc);
            oc.deskew = 0;
            ko_pipeline *op = ko_pipeline_create(&oc);
            std::vector<double> want;
            double worst = 0;
            size_t got = 0;
            for (int k = 0; k < 10; ++k) {
                const Sophus::SE3d inv = pose_of(0.8 * k, 0.05 * k, 0.01 * k).inverse();
                Points frame;
                Points sample = scene(6000);
                for (auto &p : sample) frame.push_back(inv * p);
                const double *xyz[1] = {reinterpret_cast<const double *>(frame.data())};
                const size_t n[1] = {frame.size()};
                CHECK(kicp_batch_register_frames(b, xyz, n, nullptr, nullptr) == KICP_OK);  // frame may be freed now
                ko_pipeline_register_frame(op, xyz[0], n[0], nullptr, 0);
                double To[16];
                ko_pipeline_pose(op, To);
                want.insert(want.end(), To, To + 16);
                if (k % 5 == 4) {  // five frames per sync: two blocks of four
                    CHECK(kicp_batch_sync(b) == KICP_OK);
                    std::vector<double> T(16 * 5);
                    size_t nf = 0;
                    CHECK(kicp_batch_poses(b, 0, T.data(), 5, &nf) == KICP_OK && nf == 5);
                    for (size_t i = 0; i < nf; ++i) worst = std::fmax(worst, pose_diff(T.data() + 16 * i, want.data() + 16 * (got + i)));
                    got += nf;
                }
            }
            double sec = 0;
            CHECK(kicp_batch_gather_seconds(b, &sec) == KICP_OK && sec > 0.0);
            std::printf("batch entry: worst |T_gpu - T_oracle| over %zu frames = %.3e, last pose exchange %.1f us\n", got, worst, 1e6 * sec);
            CHECK(got == 10 && worst < 1e-7);
            kicp_pipeline *pp = nullptr;
            CHECK(kicp_batch_pipeline(b, 0, &pp) == KICP_OK && pp != nullptr);
            CHECK(kicp_batch_destroy(b) == KICP_OK);
        }
    }
    lap("batch entry (RCCL) done");
}

// no argument: everything; --no-batch / --batch-only: the two halves (tests/test_cpp_api.py runs the second one LAST: on a
// machine that reads its image at a few MB/s the first use of RCCL alone takes minutes)
int main(int argc, char **argv) {
    const bool no_batch = argc > 1 && std::string(argv[1]) == "--no-batch", batch_only = argc > 1 && std::string(argv[1]) == "--batch-only";
    if (batch_only) lap("start (batch entry only)");
    if (!batch_only) run_front();
    if (!no_batch) run_batch();
    std::printf(g_fail ? "test_cpp_api: %d FAILED\n" : "test_cpp_api: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
