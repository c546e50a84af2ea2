// VoxelHashMap.hpp -- mirrors cpp/kiss_icp/core/VoxelHashMap.hpp:38-57 of PRBonn/kiss-icp v1.2.3.
// Same constructor and methods; the map itself lives in HBM behind a kicp_map handle
// (include/kicp.h) instead of a tsl::robin_map member.
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <tuple>
#include <vector>

#include "Linalg.hpp"
#include "VoxelUtils.hpp"

struct kicp_map;

namespace kiss_icp {
struct VoxelHashMap {
    explicit VoxelHashMap(double voxel_size, double max_distance, unsigned int max_points_per_voxel);
    VoxelHashMap(double voxel_size, double max_distance, unsigned int max_points_per_voxel, int device_id);
    ~VoxelHashMap();
    // value semantics as in the reference (its VoxelHashMap is implicitly copyable, VoxelHashMap.hpp:38-57): a copy is a
    // second, independent device map with the same content (kicp_map_clone) -- also a copy of the view that
    // pipeline::KissICP::VoxelMap() returns, which is how a caller snapshots a running pipeline's local map
    VoxelHashMap(const VoxelHashMap &other);
    VoxelHashMap &operator=(const VoxelHashMap &other);
    VoxelHashMap(VoxelHashMap &&other) noexcept;

    void Clear();
    bool Empty() const;
    void Update(const std::vector<Eigen::Vector3d> &points, const Eigen::Vector3d &origin);
    void Update(const std::vector<Eigen::Vector3d> &points, const Sophus::SE3d &pose);
    void AddPoints(const std::vector<Eigen::Vector3d> &points);
    void Update(PointSpan points, const Eigen::Vector3d &origin);
    void Update(PointSpan points, const Sophus::SE3d &pose);
    void AddPoints(PointSpan points);
    void RemovePointsFarFromLocation(const Eigen::Vector3d &origin);
    std::vector<Eigen::Vector3d> Pointcloud() const;
    std::tuple<Eigen::Vector3d, double> GetClosestNeighbor(const Eigen::Vector3d &query) const;
    /// batched form of GetClosestNeighbor (one kernel launch for all queries)
    std::vector<std::tuple<Eigen::Vector3d, double>> GetClosestNeighbors(
        const std::vector<Eigen::Vector3d> &queries) const;
    std::size_t NumVoxels() const;

    double voxel_size_;
    double max_distance_;
    unsigned int max_points_per_voxel_;

    // the device map; `owned_` is false for the view returned by pipeline::KissICP::VoxelMap()
    kicp_map *handle_ = nullptr;
    bool owned_ = true;
    static VoxelHashMap Borrow(kicp_map *handle, double voxel_size, double max_distance,
                               unsigned int max_points_per_voxel);

private:
    VoxelHashMap() = default;
};
}  // namespace kiss_icp
