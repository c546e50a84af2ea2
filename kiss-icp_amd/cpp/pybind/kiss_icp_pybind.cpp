// kiss_icp_pybind.cpp -- a pybind11 module with the names and signatures of the reference's
// python/kiss_icp/pybind/kiss_icp_pybind.cpp:48-144 (PRBonn/kiss-icp v1.2.3), built on the C++
// mirror in ../include (which sits on the C-ABI of libkicp.so).  The reference's thin Python
// wrappers (python/kiss_icp/{registration,mapping,voxelization,preprocess,threshold}.py) and its
// Python KissICP (python/kiss_icp/kiss_icp.py) run on this module unmodified.
//
//   _Vector3dVector      opaque std::vector<Eigen::Vector3d>, built from an (N,3) float64 array,
//                        buffer protocol + __len__/__bool__ (stl_vector_eigen.h:44-117)
//   _VoxelHashMap, _Preprocessor, _Registration, _AdaptiveThreshold, _voxel_down_sample,
//   _correct_kitti_scan  as in the reference (note the kwarg spelling max_correspondance_distance)
//   _KissICP             extra: the fused device pipeline (pipeline::KissICP)
//   _kitti_seq_error, _absolute_trajectory_error   offline metrics (host arithmetic), kept so the
//                        reference's python/kiss_icp/metrics.py works on this module
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cmath>
#include <stdexcept>
#include <vector>

#include "kiss_icp/metrics/Metrics.hpp"
#include "kiss_icp/pipeline/KissICP.hpp"

namespace py = pybind11;
using namespace py::literals;

using Vec3Vector = std::vector<Eigen::Vector3d>;
PYBIND11_MAKE_OPAQUE(Vec3Vector);

namespace {
using ArrayD = py::array_t<double, py::array::c_style | py::array::forcecast>;

Vec3Vector from_array(const ArrayD &a) {
    if (a.ndim() != 2 || a.shape(1) != 3) throw py::cast_error("expected an (N, 3) float64 array");
    Vec3Vector v(static_cast<size_t>(a.shape(0)));
    if (!v.empty()) std::memcpy(static_cast<void *>(v.data()), a.data(), v.size() * sizeof(Eigen::Vector3d));
    return v;
}

Sophus::SE3d se3_from_array(const ArrayD &T) {
    if (T.ndim() != 2 || T.shape(0) != 4 || T.shape(1) != 4) throw py::cast_error("expected a 4x4 float64 matrix");
    return kiss_icp::detail::se3_from_rowmajor(T.data());  // throws std::invalid_argument if not rigid
}

py::array_t<double> se3_to_array(const Sophus::SE3d &T) {
    py::array_t<double> out({4, 4});
    kiss_icp::detail::se3_to_rowmajor(T, out.mutable_data());
    return out;
}

// (N, 4, 4) array or a sequence of 4x4 matrices -> std::vector<Eigen::Matrix4d> (no rigidity check:
// the reference's metrics take plain Matrix4d)
std::vector<Eigen::Matrix4d> poses_from_array(const ArrayD &a) {
    if (a.ndim() != 3 || a.shape(1) != 4 || a.shape(2) != 4) throw py::cast_error("expected an (N, 4, 4) float64 array");
    std::vector<Eigen::Matrix4d> out(static_cast<size_t>(a.shape(0)));
    for (size_t i = 0; i < out.size(); ++i)
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) out[i](r, c) = a.data()[i * 16 + r * 4 + c];
    return out;
}

Eigen::Vector3d vec3_from_array(const ArrayD &a) {
    if (a.size() != 3) throw py::cast_error("expected 3 values");
    return Eigen::Vector3d(a.data()[0], a.data()[1], a.data()[2]);
}
}  // namespace

PYBIND11_MODULE(kiss_icp_pybind, m) {
    using namespace kiss_icp;
    m.doc() = "KISS-ICP registration path on MI355X (HIP) behind the reference's pybind surface";

    py::class_<Vec3Vector>(m, "_Vector3dVector", py::buffer_protocol(), "std::vector<Eigen::Vector3d>")
        .def(py::init<>())
        .def(py::init(&from_array), "array"_a)
        .def("__len__", [](const Vec3Vector &v) { return v.size(); })
        .def("__bool__", [](const Vec3Vector &v) { return !v.empty(); })
        .def("__copy__", [](const Vec3Vector &v) { return Vec3Vector(v); })
        .def("__deepcopy__", [](const Vec3Vector &v, py::dict) { return Vec3Vector(v); })
        .def_buffer([](Vec3Vector &v) {
            return py::buffer_info(v.data(), sizeof(double), py::format_descriptor<double>::format(), 2,
                                   {static_cast<py::ssize_t>(v.size()), static_cast<py::ssize_t>(3)},
                                   {static_cast<py::ssize_t>(sizeof(double) * 3), static_cast<py::ssize_t>(sizeof(double))});
        });
    py::implicitly_convertible<py::array, Vec3Vector>();

    m.def("_set_default_device", &SetDefaultDevice, "device_id"_a);

    // Map representation
    py::class_<VoxelHashMap>(m, "_VoxelHashMap", "Don't use this")
        .def(py::init<double, double, unsigned int>(), "voxel_size"_a, "max_distance"_a, "max_points_per_voxel"_a)
        .def("_clear", &VoxelHashMap::Clear)
        .def("_empty", &VoxelHashMap::Empty)
        .def(
            "_update",
            [](VoxelHashMap &self, const Vec3Vector &points, const ArrayD &pose_or_origin) {
                if (pose_or_origin.ndim() == 2) self.Update(points, se3_from_array(pose_or_origin));
                else self.Update(points, vec3_from_array(pose_or_origin));
            },
            "points"_a, "pose"_a)
        .def("_add_points", &VoxelHashMap::AddPoints, "points"_a)
        .def(
            "_remove_far_away_points",
            [](VoxelHashMap &self, const ArrayD &origin) { self.RemovePointsFarFromLocation(vec3_from_array(origin)); },
            "origin"_a)
        .def("_point_cloud", &VoxelHashMap::Pointcloud)
        .def("_num_voxels", &VoxelHashMap::NumVoxels);

    py::class_<Preprocessor>(m, "_Preprocessor", "Don't use this")
        .def(py::init<double, double, bool, int>(), "max_range"_a, "min_range"_a, "deskew"_a, "max_num_threads"_a)
        .def(
            "_preprocess",
            [](Preprocessor &self, const Vec3Vector &points, const std::vector<double> &timestamps,
               const ArrayD &relative_motion) { return self.Preprocess(points, timestamps, se3_from_array(relative_motion)); },
            "points"_a, "timestamps"_a, "relative_motion"_a);

    // Point Cloud registration
    py::class_<Registration>(m, "_Registration", "Don't use this")
        .def(py::init<int, double, int>(), "max_num_iterations"_a, "convergence_criterion"_a, "max_num_threads"_a)
        .def(
            "_align_points_to_map",
            [](Registration &self, const Vec3Vector &points, const VoxelHashMap &voxel_map, const ArrayD &T_guess,
               double max_correspondence_distance, double kernel) {
                return se3_to_array(self.AlignPointsToMap(points, voxel_map, se3_from_array(T_guess),
                                                          max_correspondence_distance, kernel));
            },
            "points"_a, "voxel_map"_a, "initial_guess"_a, "max_correspondance_distance"_a, "kernel"_a)
        .def_readonly("_last_iterations", &Registration::last_iterations_);

    // AdaptiveThreshold bindings
    py::class_<AdaptiveThreshold>(m, "_AdaptiveThreshold", "Don't use this")
        .def(py::init<double, double, double>(), "initial_threshold"_a, "min_motion_th"_a, "max_range"_a)
        .def("_compute_threshold", &AdaptiveThreshold::ComputeThreshold)
        .def(
            "_update_model_deviation",
            [](AdaptiveThreshold &self, const ArrayD &T) { self.UpdateModelDeviation(se3_from_array(T)); },
            "model_deviation"_a);

    // preprocessing modules
    m.def("_voxel_down_sample", &VoxelDownsample, "frame"_a, "voxel_size"_a);
    // KITTI-only scan correction (kiss_icp_pybind.cpp:127-138): rotate every point by 0.205 deg
    // about the axis pt x (0,0,1).  Host arithmetic, it runs once per scan in the dataloader.
    m.def(
        "_correct_kitti_scan",
        [](const Vec3Vector &frame) {
            constexpr double kVerticalAngleOffset = (0.205 * M_PI) / 180.0;
            Vec3Vector out = frame;
            const double c = std::cos(kVerticalAngleOffset), s = std::sin(kVerticalAngleOffset);
            for (auto &pt : out) {
                // axis = normalize(pt x e_z) = (y, -x, 0) / |(x, y)|;  Rodrigues' formula
                const double x = pt.x(), y = pt.y(), z = pt.z();
                const double nxy = std::sqrt(x * x + y * y);
                if (!(nxy > 0.0)) continue;  // on the z axis: Eigen's normalized() of a zero vector leaves it
                const double ax = y / nxy, ay = -x / nxy;
                const double kxv[3] = {ay * z, -ax * z, ax * y - ay * x};  // axis x pt
                const double kdv = ax * x + ay * y;                        // axis . pt
                pt = Eigen::Vector3d(x * c + kxv[0] * s + ax * kdv * (1 - c), y * c + kxv[1] * s + ay * kdv * (1 - c),
                                     z * c + kxv[2] * s);
            }
            return out;
        },
        "frame"_a);

    // Metrics (kiss_icp_pybind.cpp:141-143)
    m.def(
        "_kitti_seq_error",
        [](const ArrayD &gt, const ArrayD &res) { return metrics::SeqError(poses_from_array(gt), poses_from_array(res)); },
        "gt_poses"_a, "results_poses"_a);
    m.def(
        "_absolute_trajectory_error",
        [](const ArrayD &gt, const ArrayD &res) {
            return metrics::AbsoluteTrajectoryError(poses_from_array(gt), poses_from_array(res));
        },
        "gt_poses"_a, "results_poses"_a);

    // the fused device pipeline (not in the reference's module: its C++ KissICP is used by ROS only)
    py::class_<pipeline::KISSConfig>(m, "_KISSConfig")
        .def(py::init<>())
        .def_readwrite("voxel_size", &pipeline::KISSConfig::voxel_size)
        .def_readwrite("max_range", &pipeline::KISSConfig::max_range)
        .def_readwrite("min_range", &pipeline::KISSConfig::min_range)
        .def_readwrite("max_points_per_voxel", &pipeline::KISSConfig::max_points_per_voxel)
        .def_readwrite("min_motion_th", &pipeline::KISSConfig::min_motion_th)
        .def_readwrite("initial_threshold", &pipeline::KISSConfig::initial_threshold)
        .def_readwrite("max_num_iterations", &pipeline::KISSConfig::max_num_iterations)
        .def_readwrite("convergence_criterion", &pipeline::KISSConfig::convergence_criterion)
        .def_readwrite("max_num_threads", &pipeline::KISSConfig::max_num_threads)
        .def_readwrite("deskew", &pipeline::KISSConfig::deskew);
    py::class_<pipeline::KissICP>(m, "_KissICP")
        .def(py::init<const pipeline::KISSConfig &>(), "config"_a)
        .def(
            "_register_frame",
            [](pipeline::KissICP &self, const Vec3Vector &frame, const std::vector<double> &timestamps) {
                return self.RegisterFrame(frame, timestamps);
            },
            "frame"_a, "timestamps"_a)
        .def("_voxelize", &pipeline::KissICP::Voxelize, "frame"_a)
        .def("_local_map", &pipeline::KissICP::LocalMap)
        .def("_pose", [](const pipeline::KissICP &self) { return se3_to_array(self.pose()); })
        .def("_delta", [](const pipeline::KissICP &self) { return se3_to_array(self.delta()); })
        .def("_set_pose", [](pipeline::KissICP &self, const ArrayD &T) { self.pose() = se3_from_array(T); })
        .def("_set_delta", [](pipeline::KissICP &self, const ArrayD &T) { self.delta() = se3_from_array(T); })
        .def("_last_iterations", &pipeline::KissICP::LastIterations);
}
