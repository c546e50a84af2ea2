// VoxelUtils.hpp -- mirrors cpp/kiss_icp/core/VoxelUtils.hpp:24-51 of PRBonn/kiss-icp v1.2.3.
// PointToVoxel and the Voxel hash are plain host arithmetic; VoxelDownsample runs on the GPU
// (kicp_voxel_downsample, include/kicp.h).
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <cmath>
#include <cstdint>
#include <functional>
#include <vector>

#include "Linalg.hpp"

namespace kiss_icp {

using Voxel = Eigen::Vector3i;
inline Voxel PointToVoxel(const Eigen::Vector3d &point, const double voxel_size) {
    return Voxel(static_cast<int>(std::floor(point.x() / voxel_size)),
                 static_cast<int>(std::floor(point.y() / voxel_size)),
                 static_cast<int>(std::floor(point.z() / voxel_size)));
}

/// Voxelize a point cloud keeping the original coordinates: the first point of every voxel, emitted in the REFERENCE's
/// order -- the bucket order of the tsl::robin_map it collects them in (VoxelUtils.cpp:7-21; option "downsample_order"
/// = 0 gives ascending input order instead)
/// A view of N packed points (N x 3 float64, row-major): what a numpy array, a DLPack tensor or a
/// std::vector<Eigen::Vector3d> all are.  Every entry that takes a point vector also takes a span, so callers
/// holding their points in another container reach the device without building a vector first (not in the
/// reference, whose pybind layer copies an array into a vector: python/kiss_icp/pybind/stl_vector_eigen.h:67-80).
struct PointSpan {
    const double *xyz = nullptr;
    std::size_t n = 0;
    PointSpan() = default;
    PointSpan(const double *p, std::size_t count) : xyz(count ? p : nullptr), n(count) {}
    PointSpan(const std::vector<Eigen::Vector3d> &v)  // NOLINT: implicit on purpose
        : xyz(v.empty() ? nullptr : reinterpret_cast<const double *>(v.data())), n(v.size()) {}
};

std::vector<Eigen::Vector3d> VoxelDownsample(PointSpan frame, const double voxel_size);
std::vector<Eigen::Vector3d> VoxelDownsample(const std::vector<Eigen::Vector3d> &frame,
                                             const double voxel_size);

/// device used by the free functions and by objects created without an explicit device (default 0)
void SetDefaultDevice(int device_id);
int DefaultDevice();

}  // namespace kiss_icp

template <>
struct std::hash<kiss_icp::Voxel> {
    std::size_t operator()(const kiss_icp::Voxel &voxel) const {
        const uint32_t *vec = reinterpret_cast<const uint32_t *>(voxel.data());
        return (vec[0] * 73856093 ^ vec[1] * 19349669 ^ vec[2] * 83492791);
    }
};
