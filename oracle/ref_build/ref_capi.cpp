// ref_capi.cpp -- plain-C face of the REFERENCE's own classes (compiled from /root/reference, see Makefile), so
// that tests can drive them through ctypes next to the oracle.  TEST INFRASTRUCTURE: only tests/ and the golden
// generator load oracle/_ref/libkiss_ref.so.  Points are row-major N x 3 float64; poses row-major 4 x 4.
#include <cstring>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "kiss_icp/core/Preprocessing.hpp"
#include "kiss_icp/core/Registration.hpp"
#include "kiss_icp/core/Threshold.hpp"
#include "kiss_icp/core/VoxelHashMap.hpp"
#include "kiss_icp/core/VoxelUtils.hpp"
#include "kiss_icp/pipeline/KissICP.hpp"

using V3 = Eigen::Vector3d;
using Cloud = std::vector<V3>;

namespace {
Cloud to_cloud(const double *xyz, size_t n) {
    Cloud c(n);
    for (size_t i = 0; i < n; ++i) c[i] = V3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return c;
}
size_t from_cloud(const Cloud &c, double *out) {
    for (size_t i = 0; i < c.size(); ++i) {
        out[3 * i] = c[i].x();
        out[3 * i + 1] = c[i].y();
        out[3 * i + 2] = c[i].z();
    }
    return c.size();
}
#if KICP_REF_THIRDPARTY
// built against the REAL Eigen / Sophus / tsl / oneTBB headers (make THIRDPARTY=...): nothing of the oracle is linked
Sophus::SE3d to_se3(const double M[16]) {
    Eigen::Matrix4d m;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = M[4 * r + c];
    return Sophus::SE3d(m);  // Sophus::SE3d(Matrix4d), as the pybind layer does (kiss_icp_pybind.cpp:68,84,99)
}
void from_se3(const Sophus::SE3d &T, double M[16]) {
    const Eigen::Matrix4d m = T.matrix();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) M[4 * r + c] = m(r, c);
}
#else
Sophus::SE3d to_se3(const double M[16]) {
    ko_se3 T;
    ko_se3_from_matrix(M, &T);  // Sophus::SE3d(Matrix4d), as the pybind layer does (kiss_icp_pybind.cpp:68,84,99)
    return Sophus::SE3d(T);
}
void from_se3(const Sophus::SE3d &T, double M[16]) { ko_se3_matrix(&T.T, M); }
#endif

struct RefPipeline {
    kiss_icp::pipeline::KissICP odom;
    Cloud last_pre, last_src;
    explicit RefPipeline(const kiss_icp::pipeline::KISSConfig &c) : odom(c) {}
};
}  // namespace

extern "C" {
// which third-party code this library was built against: "shim" (oracle/ref_build/shim: Eigen's LDLT, Sophus' exp / log /
// product and tsl::robin_map's bucket order are the oracle's restatements -- the reference's own code is pinned, the
// libraries' arithmetic is not) or "thirdparty" (the real headers: everything is pinned)
const char *kr_build_mode(void) {
#if KICP_REF_THIRDPARTY
    return "thirdparty";
#else
    return "shim";
#endif
}
// the arithmetic the reference takes from its libraries, exposed so that the oracle's restatements can be held against
// the real thing where the real headers are present (tests/test_ref_pins_oracle.py::test_third_party_arithmetic)
void kr_se3_exp(const double a[6], double M[16]) {
    Eigen::Matrix<double, 6, 1> x;
    for (int i = 0; i < 6; ++i) x(i) = a[i];
    from_se3(Sophus::SE3d::exp(x), M);
}
void kr_se3_log(const double M[16], double a[6]) {
    const Eigen::Matrix<double, 6, 1> x = to_se3(M).log();
    for (int i = 0; i < 6; ++i) a[i] = x(i);
}
void kr_se3_mul(const double A[16], const double B[16], double M[16]) { from_se3(to_se3(A) * to_se3(B), M); }
void kr_ldlt6_solve(const double A[36], const double b[6], double x[6]) {
    Eigen::Matrix<double, 6, 6> m;
    Eigen::Matrix<double, 6, 1> r;
    for (int i = 0; i < 6; ++i) {
        r(i) = b[i];
        for (int j = 0; j < 6; ++j) m(i, j) = A[6 * i + j];
    }
    const Eigen::Matrix<double, 6, 1> s = m.ldlt().solve(r);
    for (int i = 0; i < 6; ++i) x[i] = s(i);
}
// ---- VoxelDownsample / Preprocess
size_t kr_voxel_downsample(const double *xyz, size_t n, double voxel_size, double *out) {
    return from_cloud(kiss_icp::VoxelDownsample(to_cloud(xyz, n), voxel_size), out);
}
long kr_preprocess(const double *xyz, size_t n, const double *ts, size_t n_ts, const double motion[16], double max_range,
                   double min_range, int deskew, double *out) {
    kiss_icp::Preprocessor p(max_range, min_range, deskew != 0, 1);
    std::vector<double> t(ts, ts + n_ts);
    try {
        return (long)from_cloud(p.Preprocess(to_cloud(xyz, n), t, to_se3(motion)), out);
    } catch (const std::out_of_range &) {
        return -1;  // timestamps shorter than the frame: std::vector::at throws (Preprocessing.cpp:76-77)
    }
}
// ---- VoxelHashMap
void *kr_map_create(double voxel_size, double max_distance, unsigned max_points) {
    return new kiss_icp::VoxelHashMap(voxel_size, max_distance, max_points);
}
void kr_map_destroy(void *m) { delete static_cast<kiss_icp::VoxelHashMap *>(m); }
void kr_map_clear(void *m) { static_cast<kiss_icp::VoxelHashMap *>(m)->Clear(); }
int kr_map_empty(const void *m) { return static_cast<const kiss_icp::VoxelHashMap *>(m)->Empty() ? 1 : 0; }
size_t kr_map_num_voxels(const void *m) { return static_cast<const kiss_icp::VoxelHashMap *>(m)->map_.size(); }
void kr_map_add_points(void *m, const double *xyz, size_t n) { static_cast<kiss_icp::VoxelHashMap *>(m)->AddPoints(to_cloud(xyz, n)); }
void kr_map_remove_far(void *m, const double o[3]) {
    static_cast<kiss_icp::VoxelHashMap *>(m)->RemovePointsFarFromLocation(V3(o[0], o[1], o[2]));
}
void kr_map_update_origin(void *m, const double *xyz, size_t n, const double o[3]) {
    static_cast<kiss_icp::VoxelHashMap *>(m)->Update(to_cloud(xyz, n), V3(o[0], o[1], o[2]));
}
void kr_map_update_pose(void *m, const double *xyz, size_t n, const double T[16]) {
    static_cast<kiss_icp::VoxelHashMap *>(m)->Update(to_cloud(xyz, n), to_se3(T));
}
size_t kr_map_num_points(const void *m) { return static_cast<const kiss_icp::VoxelHashMap *>(m)->Pointcloud().size(); }
size_t kr_map_pointcloud(const void *m, double *out) { return from_cloud(static_cast<const kiss_icp::VoxelHashMap *>(m)->Pointcloud(), out); }
double kr_map_closest_neighbor(const void *m, const double q[3], double nn[3]) {
    const auto [p, d] = static_cast<const kiss_icp::VoxelHashMap *>(m)->GetClosestNeighbor(V3(q[0], q[1], q[2]));
    nn[0] = p.x();
    nn[1] = p.y();
    nn[2] = p.z();
    return d;
}
// ---- Registration
void kr_align_points_to_map(const double *xyz, size_t n, const void *m, const double guess[16], double max_dist, double kernel,
                            int max_iters, double conv, double T_out[16]) {
    kiss_icp::Registration reg(max_iters, conv, 1);
    from_se3(reg.AlignPointsToMap(to_cloud(xyz, n), *static_cast<const kiss_icp::VoxelHashMap *>(m), to_se3(guess), max_dist, kernel), T_out);
}
// ---- AdaptiveThreshold
void kr_threshold_step(double *model_sse, int *num_samples, double min_motion_th, double max_range, const double dev[16], double *sigma_after) {
    kiss_icp::AdaptiveThreshold t(0.0, min_motion_th, max_range);
    t.model_sse_ = *model_sse;
    t.num_samples_ = *num_samples;
    t.UpdateModelDeviation(to_se3(dev));
    *model_sse = t.model_sse_;
    *num_samples = t.num_samples_;
    *sigma_after = t.ComputeThreshold();
}
// ---- pipeline::KissICP
void *kr_pipeline_create(double voxel_size, double max_range, double min_range, int max_points_per_voxel, double min_motion_th,
                         double initial_threshold, int max_num_iterations, double convergence_criterion, int deskew) {
    kiss_icp::pipeline::KISSConfig c;
    c.voxel_size = voxel_size;
    c.max_range = max_range;
    c.min_range = min_range;
    c.max_points_per_voxel = max_points_per_voxel;
    c.min_motion_th = min_motion_th;
    c.initial_threshold = initial_threshold;
    c.max_num_iterations = max_num_iterations;
    c.convergence_criterion = convergence_criterion;
    c.max_num_threads = 1;
    c.deskew = deskew != 0;
    return new RefPipeline(c);
}
void kr_pipeline_destroy(void *p) { delete static_cast<RefPipeline *>(p); }
void kr_pipeline_register_frame(void *p, const double *xyz, size_t n, const double *ts, size_t n_ts) {
    auto *r = static_cast<RefPipeline *>(p);
    std::vector<double> t(ts, ts + n_ts);
    std::tie(r->last_pre, r->last_src) = r->odom.RegisterFrame(to_cloud(xyz, n), t);
}
void kr_pipeline_pose(const void *p, double T[16]) { from_se3(static_cast<const RefPipeline *>(p)->odom.pose(), T); }
void kr_pipeline_delta(const void *p, double T[16]) { from_se3(static_cast<const RefPipeline *>(p)->odom.delta(), T); }
const void *kr_pipeline_map(const void *p) { return &static_cast<const RefPipeline *>(p)->odom.VoxelMap(); }
size_t kr_pipeline_output_size(const void *p, int which) {
    const auto *r = static_cast<const RefPipeline *>(p);
    return which == 0 ? r->last_pre.size() : r->last_src.size();
}
size_t kr_pipeline_output(const void *p, int which, double *out) {
    const auto *r = static_cast<const RefPipeline *>(p);
    return from_cloud(which == 0 ? r->last_pre : r->last_src, out);
}
}  // extern "C"
