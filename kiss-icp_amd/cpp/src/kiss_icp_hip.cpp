// kiss_icp_hip.cpp -- the reference's C++ classes (namespace kiss_icp, kiss_icp::pipeline) re-hosted
// on the C-ABI of libkicp.so (include/kicp.h).  Plain C++17, compiled with g++: nothing here
// touches HIP directly.  Statuses become exceptions at this layer (the C-ABI never throws):
//   KICP_ERR_TIMESTAMPS -> std::out_of_range  (the reference's std::vector::at)
//   KICP_ERR_INVALID_ARG -> std::invalid_argument
//   everything else      -> std::runtime_error with kicp_last_error()
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>

#include "kicp.h"
#include "kiss_icp/pipeline/KissICP.hpp"

namespace kiss_icp {

namespace {
std::atomic<int> g_default_device{0};

void check(int status, const char *what) {
    if (status == KICP_OK) return;
    const std::string msg = std::string(what) + ": " + kicp_status_string(status) + " (" + kicp_last_error() + ")";
    if (status == KICP_ERR_TIMESTAMPS) throw std::out_of_range(msg);
    if (status == KICP_ERR_INVALID_ARG) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

const double *xyz(const std::vector<Eigen::Vector3d> &v) {
    static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "points must be packed N x 3 doubles");
    return v.empty() ? nullptr : reinterpret_cast<const double *>(v.data());
}
double *xyz(std::vector<Eigen::Vector3d> &v) { return v.empty() ? nullptr : reinterpret_cast<double *>(v.data()); }
}  // namespace

void SetDefaultDevice(int device_id) { g_default_device = device_id; }
int DefaultDevice() { return g_default_device; }

// ---- VoxelUtils ---------------------------------------------------------------------------------
std::vector<Eigen::Vector3d> VoxelDownsample(const std::vector<Eigen::Vector3d> &frame, const double voxel_size) {
    return VoxelDownsample(PointSpan(frame), voxel_size);
}
std::vector<Eigen::Vector3d> VoxelDownsample(PointSpan frame, const double voxel_size) {
    std::vector<Eigen::Vector3d> out(frame.n);
    size_t n = 0;
    check(kicp_voxel_downsample(frame.xyz, frame.n, voxel_size, DefaultDevice(), xyz(out), &n), "VoxelDownsample");
    out.resize(n);
    out.shrink_to_fit();
    return out;
}

// ---- VoxelHashMap ---------------------------------------------------------------------------------
VoxelHashMap::VoxelHashMap(double voxel_size, double max_distance, unsigned int max_points_per_voxel)
    : VoxelHashMap(voxel_size, max_distance, max_points_per_voxel, DefaultDevice()) {}

VoxelHashMap::VoxelHashMap(double voxel_size, double max_distance, unsigned int max_points_per_voxel, int device_id)
    : voxel_size_(voxel_size), max_distance_(max_distance), max_points_per_voxel_(max_points_per_voxel) {
    check(kicp_map_create(voxel_size, max_distance, max_points_per_voxel, device_id, &handle_), "VoxelHashMap");
}

VoxelHashMap VoxelHashMap::Borrow(kicp_map *handle, double voxel_size, double max_distance,
                                  unsigned int max_points_per_voxel) {
    VoxelHashMap m;
    m.voxel_size_ = voxel_size;
    m.max_distance_ = max_distance;
    m.max_points_per_voxel_ = max_points_per_voxel;
    m.handle_ = handle;
    m.owned_ = false;
    return m;
}

VoxelHashMap::VoxelHashMap(VoxelHashMap &&o) noexcept
    : voxel_size_(o.voxel_size_),
      max_distance_(o.max_distance_),
      max_points_per_voxel_(o.max_points_per_voxel_),
      handle_(o.handle_),
      owned_(o.owned_) {
    o.handle_ = nullptr;
}

VoxelHashMap::VoxelHashMap(const VoxelHashMap &o)
    : voxel_size_(o.voxel_size_), max_distance_(o.max_distance_), max_points_per_voxel_(o.max_points_per_voxel_) {
    if (o.handle_) check(kicp_map_clone(o.handle_, &handle_), "VoxelHashMap(const VoxelHashMap &)");
}

VoxelHashMap &VoxelHashMap::operator=(const VoxelHashMap &o) {
    if (this == &o) return *this;
    kicp_map *fresh = nullptr;
    if (o.handle_) check(kicp_map_clone(o.handle_, &fresh), "VoxelHashMap::operator=");
    if (handle_ && owned_) kicp_map_destroy(handle_);
    handle_ = fresh;
    owned_ = true;
    voxel_size_ = o.voxel_size_;
    max_distance_ = o.max_distance_;
    max_points_per_voxel_ = o.max_points_per_voxel_;
    return *this;
}

VoxelHashMap::~VoxelHashMap() {
    if (handle_ && owned_) kicp_map_destroy(handle_);
}

void VoxelHashMap::Clear() { check(kicp_map_clear(handle_), "VoxelHashMap::Clear"); }
bool VoxelHashMap::Empty() const {
    int e = 1;
    check(kicp_map_empty(handle_, &e), "VoxelHashMap::Empty");
    return e != 0;
}
std::size_t VoxelHashMap::NumVoxels() const {
    size_t nv = 0;
    check(kicp_map_size(handle_, &nv, nullptr), "VoxelHashMap::NumVoxels");
    return nv;
}
void VoxelHashMap::Update(const std::vector<Eigen::Vector3d> &points, const Eigen::Vector3d &origin) {
    Update(PointSpan(points), origin);
}
void VoxelHashMap::Update(const std::vector<Eigen::Vector3d> &points, const Sophus::SE3d &pose) {
    Update(PointSpan(points), pose);
}
void VoxelHashMap::AddPoints(const std::vector<Eigen::Vector3d> &points) { AddPoints(PointSpan(points)); }
void VoxelHashMap::Update(PointSpan points, const Eigen::Vector3d &origin) {
    check(kicp_map_update_origin(handle_, points.xyz, points.n, origin.data()), "VoxelHashMap::Update");
}
void VoxelHashMap::Update(PointSpan points, const Sophus::SE3d &pose) {
    double T[16];
    detail::se3_to_rowmajor(pose, T);
    check(kicp_map_update_pose(handle_, points.xyz, points.n, T), "VoxelHashMap::Update");
}
void VoxelHashMap::AddPoints(PointSpan points) {
    check(kicp_map_add_points(handle_, points.xyz, points.n), "VoxelHashMap::AddPoints");
}
void VoxelHashMap::RemovePointsFarFromLocation(const Eigen::Vector3d &origin) {
    check(kicp_map_remove_far(handle_, origin.data()), "VoxelHashMap::RemovePointsFarFromLocation");
}
std::vector<Eigen::Vector3d> VoxelHashMap::Pointcloud() const {
    size_t n = 0;
    check(kicp_map_size(handle_, nullptr, &n), "VoxelHashMap::Pointcloud");
    std::vector<Eigen::Vector3d> out(n);
    size_t got = 0;
    check(kicp_map_pointcloud(handle_, xyz(out), n, &got), "VoxelHashMap::Pointcloud");
    out.resize(got < n ? got : n);
    return out;
}
std::vector<std::tuple<Eigen::Vector3d, double>> VoxelHashMap::GetClosestNeighbors(
    const std::vector<Eigen::Vector3d> &queries) const {
    std::vector<Eigen::Vector3d> nn(queries.size());
    std::vector<double> dist(queries.size());
    check(kicp_map_closest_neighbor(handle_, xyz(queries), queries.size(), xyz(nn), dist.data()),
          "VoxelHashMap::GetClosestNeighbor");
    std::vector<std::tuple<Eigen::Vector3d, double>> out;
    out.reserve(queries.size());
    for (size_t i = 0; i < queries.size(); ++i) out.emplace_back(nn[i], dist[i]);
    return out;
}
std::tuple<Eigen::Vector3d, double> VoxelHashMap::GetClosestNeighbor(const Eigen::Vector3d &query) const {
    return GetClosestNeighbors({query})[0];
}

// ---- Registration -----------------------------------------------------------------------------------
Registration::Registration(int max_num_iteration, double convergence_criterion, int max_num_threads)
    : Registration(max_num_iteration, convergence_criterion, max_num_threads, DefaultDevice()) {}

Registration::Registration(int max_num_iteration, double convergence_criterion, int max_num_threads, int device_id)
    : max_num_iterations_(max_num_iteration),
      convergence_criterion_(convergence_criterion),
      max_num_threads_(max_num_threads),
      device_id_(device_id) {
    check(kicp_registration_create(max_num_iteration, convergence_criterion, max_num_threads, device_id, &handle_),
          "Registration");
}
Registration::~Registration() {
    if (handle_) kicp_registration_destroy(handle_);
}
Registration::Registration(const Registration &o)
    : Registration(o.max_num_iterations_, o.convergence_criterion_, o.max_num_threads_, o.device_id_) {
    last_iterations_ = o.last_iterations_;
    last_converged_ = o.last_converged_;
    last_points_examined_ = o.last_points_examined_;
}
Registration &Registration::operator=(const Registration &o) {
    if (this == &o) return *this;
    Registration fresh(o);  // (the public parameter fields may have been edited since construction: a new handle from what they say now)
    *this = std::move(fresh);
    return *this;
}
Registration::Registration(Registration &&o) noexcept
    : max_num_iterations_(o.max_num_iterations_),
      convergence_criterion_(o.convergence_criterion_),
      max_num_threads_(o.max_num_threads_),
      last_iterations_(o.last_iterations_),
      last_converged_(o.last_converged_),
      last_points_examined_(o.last_points_examined_),
      handle_(o.handle_),
      device_id_(o.device_id_) {
    o.handle_ = nullptr;
}
Registration &Registration::operator=(Registration &&o) noexcept {
    if (this == &o) return *this;
    if (handle_) kicp_registration_destroy(handle_);
    max_num_iterations_ = o.max_num_iterations_;
    convergence_criterion_ = o.convergence_criterion_;
    max_num_threads_ = o.max_num_threads_;
    last_iterations_ = o.last_iterations_;
    last_converged_ = o.last_converged_;
    last_points_examined_ = o.last_points_examined_;
    handle_ = o.handle_;
    device_id_ = o.device_id_;
    o.handle_ = nullptr;
    return *this;
}

Sophus::SE3d Registration::AlignPointsToMap(const std::vector<Eigen::Vector3d> &frame, const VoxelHashMap &voxel_map,
                                            const Sophus::SE3d &initial_guess,
                                            const double max_correspondence_distance, const double kernel_scale) {
    return AlignPointsToMap(PointSpan(frame), voxel_map, initial_guess, max_correspondence_distance, kernel_scale);
}
Sophus::SE3d Registration::AlignPointsToMap(PointSpan frame, const VoxelHashMap &voxel_map, const Sophus::SE3d &initial_guess,
                                            const double max_correspondence_distance, const double kernel_scale) {
    double guess[16], out[16];
    detail::se3_to_rowmajor(initial_guess, guess);
    kicp_icp_stats st;
    check(kicp_align_points_to_map(handle_, frame.xyz, frame.n, voxel_map.handle_, guess,
                                   max_correspondence_distance, kernel_scale, out, &st),
          "Registration::AlignPointsToMap");
    last_iterations_ = st.iterations;
    last_converged_ = st.converged != 0;
    last_points_examined_ = st.points_examined;
    return detail::se3_from_rowmajor(out);
}

// ---- Preprocessor -----------------------------------------------------------------------------------
Preprocessor::Preprocessor(const double max_range, const double min_range, const bool deskew,
                           const int max_num_threads)
    : max_range_(max_range), min_range_(min_range), deskew_(deskew), max_num_threads_(max_num_threads) {}

std::vector<Eigen::Vector3d> Preprocessor::Preprocess(const std::vector<Eigen::Vector3d> &frame,
                                                      const std::vector<double> &timestamps,
                                                      const Sophus::SE3d &relative_motion) const {
    return Preprocess(PointSpan(frame), timestamps.empty() ? nullptr : timestamps.data(), timestamps.size(), relative_motion);
}
std::vector<Eigen::Vector3d> Preprocessor::Preprocess(PointSpan frame, const double *timestamps, std::size_t n_timestamps,
                                                      const Sophus::SE3d &relative_motion) const {
    double T[16];
    detail::se3_to_rowmajor(relative_motion, T);
    std::vector<Eigen::Vector3d> out(frame.n);
    size_t n = 0;
    check(kicp_preprocess(frame.xyz, frame.n, n_timestamps ? timestamps : nullptr, n_timestamps, T, max_range_, min_range_,
                          deskew_ ? 1 : 0,
                          device_id_ >= 0 ? device_id_ : DefaultDevice(), xyz(out), &n),
          "Preprocessor::Preprocess");
    out.resize(n);
    out.shrink_to_fit();
    return out;
}

// ---- AdaptiveThreshold (core/Threshold.cpp:38-49) ------------------------------------------------------
void AdaptiveThreshold::UpdateModelDeviation(const Sophus::SE3d &current_deviation) {
    const double model_error = [&]() {
        // Eigen::AngleAxisd(R).angle(): 2 * atan2(|q.vec|, |q.w|) of the rotation's quaternion
        const auto R = current_deviation.rotationMatrix();
        const double c = 0.5 * (R(0, 0) + R(1, 1) + R(2, 2) - 1.0);
        const double sx = R(2, 1) - R(1, 2), sy = R(0, 2) - R(2, 0), sz = R(1, 0) - R(0, 1);
        const double theta = std::atan2(0.5 * std::sqrt(sx * sx + sy * sy + sz * sz), c);
        const double delta_rot = 2.0 * max_range_ * std::sin(theta / 2.0);
        const double delta_trans = current_deviation.translation().norm();
        return delta_trans + delta_rot;
    }();
    if (model_error > min_motion_threshold_) {
        model_sse_ += model_error * model_error;
        num_samples_++;
    }
}

namespace pipeline {

namespace {
kicp_config to_c(const KISSConfig &c) {
    kicp_config k;
    kicp_config_default(&k);
    k.voxel_size = c.voxel_size;
    k.max_range = c.max_range;
    k.min_range = c.min_range;
    k.max_points_per_voxel = c.max_points_per_voxel;
    k.min_motion_th = c.min_motion_th;
    k.initial_threshold = c.initial_threshold;
    k.max_num_iterations = c.max_num_iterations;
    k.convergence_criterion = c.convergence_criterion;
    k.max_num_threads = c.max_num_threads;
    k.deskew = c.deskew ? 1 : 0;
    return k;
}
kicp_pipeline *make_pipeline(const KISSConfig &c, int device_id) {
    const kicp_config k = to_c(c);
    kicp_pipeline *h = nullptr;
    check(kicp_pipeline_create(&k, device_id, &h), "KissICP");
    return h;
}
kicp_map *map_of(kicp_pipeline *h) {
    kicp_map *m = nullptr;
    check(kicp_pipeline_map(h, &m), "KissICP::VoxelMap");
    return m;
}
}  // namespace

KissICP::KissICP(const KISSConfig &config) : KissICP(config, DefaultDevice()) {}

KissICP::KissICP(const KISSConfig &config, int device_id)
    : config_(config),
      device_id_(device_id),
      handle_(make_pipeline(config, device_id)),
      local_map_(VoxelHashMap::Borrow(map_of(handle_), config.voxel_size, config.max_range,
                                      static_cast<unsigned>(config.max_points_per_voxel))) {
    detail::se3_to_rowmajor(last_pose_, dev_pose_);
    detail::se3_to_rowmajor(last_delta_, dev_delta_);
}

KissICP::~KissICP() {
    if (handle_) kicp_pipeline_destroy(handle_);
}

void KissICP::PushPoseEdits() {
    // pose()/delta() hand out mutable references (KissICP.hpp:80-84): push edits to the device
    double T[16];
    detail::se3_to_rowmajor(last_pose_, T);
    if (std::memcmp(T, dev_pose_, sizeof T) != 0) check(kicp_pipeline_set_pose(handle_, T), "KissICP::pose");
    detail::se3_to_rowmajor(last_delta_, T);
    if (std::memcmp(T, dev_delta_, sizeof T) != 0) check(kicp_pipeline_set_delta(handle_, T), "KissICP::delta");
}

void KissICP::CollectState() {
    check(kicp_pipeline_pose(handle_, dev_pose_), "KissICP::pose");
    check(kicp_pipeline_delta(handle_, dev_delta_), "KissICP::delta");
    last_pose_ = detail::se3_from_rowmajor(dev_pose_);
    last_delta_ = detail::se3_from_rowmajor(dev_delta_);
    // re-derive what the device will compare against (quaternion <-> matrix round trip is not bit-stable)
    detail::se3_to_rowmajor(last_pose_, dev_pose_);
    detail::se3_to_rowmajor(last_delta_, dev_delta_);

    kicp_frame_stats fs;
    check(kicp_pipeline_last_stats(handle_, &fs), "KissICP::RegisterFrame");
    last_iterations_ = fs.icp.iterations;
    last_sigma_ = fs.sigma;
    last_n_pre_ = fs.n_preprocessed;
    last_n_src_ = fs.n_source;
}

KissICP::Vector3dVectorTuple KissICP::CollectFrame() {
    CollectState();
    Vector3dVector pre(last_n_pre_), source(last_n_src_);
    size_t n = 0;
    check(kicp_pipeline_output(handle_, KICP_OUT_PREPROCESSED, xyz(pre), pre.size(), &n), "KissICP::RegisterFrame");
    check(kicp_pipeline_output(handle_, KICP_OUT_SOURCE, xyz(source), source.size(), &n), "KissICP::RegisterFrame");
    return {std::move(pre), std::move(source)};  // KissICP.cpp:67
}

KissICP::Vector3dVectorTuple KissICP::RegisterFrame(const std::vector<Eigen::Vector3d> &frame,
                                                    const std::vector<double> &timestamps) {
    return RegisterFrame(PointSpan(frame), timestamps.empty() ? nullptr : timestamps.data(), timestamps.size());
}

KissICP::Vector3dVectorTuple KissICP::RegisterFrame(PointSpan frame, const double *timestamps, std::size_t n_timestamps) {
    PushPoseEdits();
    // The two vectors the reference's signature returns are new memory every call, and 3 MB of fresh pages cost what the whole
    // registration costs.  So: queue the scan, allocate the large one WHILE the device works, then collect -- the
    // preprocessed frame reaches the vector while the registration is still running (kicp.h: kicp_pipeline_collect_outputs).
    check(kicp_pipeline_sync(handle_), "KissICP::RegisterFrame");  // (frames a caller queued through other entries)
    check(kicp_pipeline_register_frame_async(handle_, frame.xyz, frame.n, n_timestamps ? timestamps : nullptr, n_timestamps),
          "KissICP::RegisterFrame");
    Vector3dVector pre(frame.n);
    const double *src = nullptr;
    size_t n_pre = 0, n_src = 0;
    check(kicp_pipeline_collect_outputs(handle_, xyz(pre), pre.size(), &n_pre, &src, &n_src), "KissICP::RegisterFrame");
    pre.resize(n_pre);  // (shrinks in place)
    // (the vector first, the getters after: the view is only promised until the next call that stages an output)
    const auto *s3 = reinterpret_cast<const Eigen::Vector3d *>(src);
    Vector3dVectorTuple out{std::move(pre), Vector3dVector(s3, s3 + n_src)};  // KissICP.cpp:67
    CollectState();
    return out;
}

KissICP::Vector3dVectorTuple KissICP::RegisterFrameDevice(const double *d_xyz, std::size_t n, const double *d_timestamps,
                                                          std::size_t n_timestamps) {
    PushPoseEdits();
    check(kicp_pipeline_register_frame_device(handle_, d_xyz, n, n_timestamps ? d_timestamps : nullptr, n_timestamps),
          "KissICP::RegisterFrameDevice");
    check(kicp_pipeline_sync(handle_), "KissICP::RegisterFrameDevice");
    return CollectFrame();
}

KissICP::Vector3dVectorTuple KissICP::Voxelize(const std::vector<Eigen::Vector3d> &frame) const {
    return Voxelize(PointSpan(frame));
}

KissICP::Vector3dVectorTuple KissICP::Voxelize(PointSpan frame) const {
    const auto voxel_size = config_.voxel_size;  // KissICP.cpp:70-75, on THIS pipeline's device
    auto downsample = [this](PointSpan in, double v) {
        Vector3dVector out(in.n);
        size_t n = 0;
        check(kicp_voxel_downsample(in.xyz, in.n, v, device_id_, xyz(out), &n), "KissICP::Voxelize");
        out.resize(n);
        return out;
    };
    auto frame_downsample = downsample(frame, voxel_size * 0.5);
    auto source = downsample(PointSpan(frame_downsample), voxel_size * 1.5);
    return {std::move(source), std::move(frame_downsample)};
}

}  // namespace pipeline
}  // namespace kiss_icp
