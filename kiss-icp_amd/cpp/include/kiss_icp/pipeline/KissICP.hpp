// KissICP.hpp -- mirrors cpp/kiss_icp/pipeline/KissICP.hpp:36-96 of PRBonn/kiss-icp v1.2.3: the same
// KISSConfig fields and defaults, the same KissICP members (RegisterFrame, Voxelize, LocalMap,
// VoxelMap, pose, delta).  RegisterFrame runs entirely on the GPU through the fused device pipeline
// (kicp_pipeline_*, include/kicp.h); only the raw scan goes in and the pose comes out.
// API declarations reproduced from PRBonn/kiss-icp (MIT License, Copyright (c) 2022 Ignacio Vizzo, Tiziano Guadagnino,
// Benedikt Mersch, Cyrill Stachniss) so that existing callers compile unchanged; the implementation behind them is this
// repository's own.
#pragma once

#include <tuple>
#include <vector>

#include "kiss_icp/core/Linalg.hpp"
#include "kiss_icp/core/Preprocessing.hpp"
#include "kiss_icp/core/Registration.hpp"
#include "kiss_icp/core/Threshold.hpp"
#include "kiss_icp/core/VoxelHashMap.hpp"

struct kicp_pipeline;

namespace kiss_icp::pipeline {

struct KISSConfig {
    // map params
    double voxel_size = 1.0;
    double max_range = 100.0;
    double min_range = 0.0;
    int max_points_per_voxel = 20;

    // th parms
    double min_motion_th = 0.1;
    double initial_threshold = 2.0;

    // registration params
    int max_num_iterations = 500;
    double convergence_criterion = 0.0001;
    int max_num_threads = 0;

    // Motion compensation
    bool deskew = true;
};

class KissICP {
public:
    using Vector3dVector = std::vector<Eigen::Vector3d>;
    using Vector3dVectorTuple = std::tuple<Vector3dVector, Vector3dVector>;

public:
    explicit KissICP(const KISSConfig &config);
    KissICP(const KISSConfig &config, int device_id);
    ~KissICP();
    KissICP(const KissICP &) = delete;
    KissICP &operator=(const KissICP &) = delete;

public:
    Vector3dVectorTuple RegisterFrame(const std::vector<Eigen::Vector3d> &frame,
                                      const std::vector<double> &timestamps);
    Vector3dVectorTuple Voxelize(const std::vector<Eigen::Vector3d> &frame) const;
    // the same on a view of packed points (a numpy array, a DLPack tensor) ...
    Vector3dVectorTuple RegisterFrame(PointSpan frame, const double *timestamps, std::size_t n_timestamps);
    Vector3dVectorTuple Voxelize(PointSpan frame) const;
    // ... and on a scan that already lies in THIS pipeline's HBM (N x 3 float64; timestamps there too or
    // nullptr).  The producer's work on the buffers must have completed (hipStreamSynchronize /
    // torch.cuda.synchronize); the call returns after the registration, so the buffers are free again.
    Vector3dVectorTuple RegisterFrameDevice(const double *d_xyz, std::size_t n, const double *d_timestamps,
                                            std::size_t n_timestamps);
    int Device() const { return device_id_; }

    std::vector<Eigen::Vector3d> LocalMap() const { return local_map_.Pointcloud(); };

    const VoxelHashMap &VoxelMap() const { return local_map_; };
    VoxelHashMap &VoxelMap() { return local_map_; };

    // mutable references like the reference's: a pose or delta edited by the caller is pushed to
    // the device before the next RegisterFrame
    const Sophus::SE3d &pose() const { return last_pose_; }
    Sophus::SE3d &pose() { return last_pose_; }

    const Sophus::SE3d &delta() const { return last_delta_; }
    Sophus::SE3d &delta() { return last_delta_; }

    /// ICP iterations of the last frame and the adaptive threshold it used
    int LastIterations() const { return last_iterations_; }
    double LastSigma() const { return last_sigma_; }

private:
    void PushPoseEdits();
    void CollectState();
    Vector3dVectorTuple CollectFrame();
    Sophus::SE3d last_pose_;
    Sophus::SE3d last_delta_;
    double dev_pose_[16];   // what the device holds (row-major), to detect edits through pose()/delta()
    double dev_delta_[16];

    KISSConfig config_;
    int device_id_ = 0;
    kicp_pipeline *handle_ = nullptr;
    VoxelHashMap local_map_;  // view of the pipeline's device map
    int last_iterations_ = 0;
    double last_sigma_ = 0.0;
    std::size_t last_n_pre_ = 0, last_n_src_ = 0;
};

}  // namespace kiss_icp::pipeline
