/*
 * kicp.h -- C-ABI of the MI355X-native KISS-ICP registration hot path (libkicp.so).
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference interface it replaces (file:line under
 * PRBonn/kiss-icp v1.2.3).  The reference's C++ classes (kiss_icp::Registration,
 * kiss_icp::VoxelHashMap, kiss_icp::pipeline::KissICP) and its pybind module are re-hosted on
 * top of these calls (kiss-icp_amd/cpp/, kiss-icp_amd/python/); INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions
 *   - points: row-major N x 3 float64, i.e. std::vector<Eigen::Vector3d>::data().
 *   - SE(3): row-major 4 x 4 float64 (numpy's natural layout; Eigen::Matrix4d transposed).
 *   - every function returns a kicp_status; nothing throws or aborts across the boundary.
 *   - the caller owns every buffer it passes; the library keeps no pointer past the call.
 *   - handles are single-threaded objects (like the reference's classes); distinct handles may
 *     be used from distinct threads / on distinct GPUs concurrently.
 *   - there is NO CPU fallback: with no gfx950 device every create call fails with
 *     KICP_ERR_NO_DEVICE.
 *   - voxel coordinates must satisfy |floor(p / voxel_size)| < 2^20 on every axis (the
 *     device hash packs a voxel into one 64-bit word); KICP_ERR_RANGE otherwise.
 */
#ifndef KICP_H
#define KICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KICP_VERSION_MAJOR 0
#define KICP_VERSION_MINOR 3

typedef enum kicp_status {
    KICP_OK = 0,
    KICP_ERR_INVALID_ARG = 1, /* null pointer, bad shape, non-orthonormal pose (SOPHUS_ENSURE) */
    KICP_ERR_HIP = 2,         /* a HIP runtime call failed; see kicp_last_error() */
    KICP_ERR_OOM = 3,         /* device or host allocation failed */
    KICP_ERR_CAPACITY = 4,    /* a device table could not grow any further */
    KICP_ERR_RANGE = 5,       /* voxel coordinate outside +-2^20 */
    KICP_ERR_TIMEOUT = 6,     /* a bounded wait gave up: inside the registration kernel (its workgroups were not all resident; the
                                 launch commits nothing and is replayed), or on the host -- EVERY wait of this library for the
                                 device has a deadline ("wait_timeout_ms"): the call returns, the work stays queued, a later
                                 sync collects it; a handle destroyed while its work does not end is leaked, not waited for */
    KICP_ERR_NO_DEVICE = 7,   /* no usable gfx950 GPU */
    KICP_ERR_TIMESTAMPS = 8   /* 0 < n_timestamps < n_points (std::vector::at would throw,
                                 core/Preprocessing.cpp:76-77) */
} kicp_status;

const char *kicp_status_string(int status);
/* message of the last failing call on this thread ("" if none) */
const char *kicp_last_error(void);
int kicp_version(int *major, int *minor);
int kicp_device_count(int *count);
/* name and gcnArchName of a device, for logs */
int kicp_device_name(int device_id, char *buf, size_t buf_len);

/* ------------------------------------------------------------------------------------------
 * VoxelHashMap  -- replaces kiss_icp::VoxelHashMap (core/VoxelHashMap.hpp:38-57,
 * core/VoxelHashMap.cpp:35-132) / pybind _VoxelHashMap (pybind/kiss_icp_pybind.cpp:54-74).
 * The map lives in HBM: an open-addressed hash of packed voxel keys over a pool of
 * fixed-stride voxel blocks.
 * ---------------------------------------------------------------------------------------- */
typedef struct kicp_map kicp_map;

/* VoxelHashMap(voxel_size, max_distance, max_points_per_voxel)  VoxelHashMap.hpp:39-42 */
int kicp_map_create(double voxel_size, double max_distance, unsigned max_points_per_voxel,
                    int device_id, kicp_map **out);
int kicp_map_destroy(kicp_map *map);
/* the implicit copy constructor of the reference's VoxelHashMap (a struct of scalars and a tsl::robin_map: copyable,
 * VoxelHashMap.hpp:38-57): a second map with the same voxels, points and parameters on the same device, from then on
 * independent of `src` (also when `src` is the map a pipeline owns).  Waits for the work queued on `src`. */
int kicp_map_clone(const kicp_map *src, kicp_map **out);
int kicp_map_clear(kicp_map *map);                          /* Clear()   VoxelHashMap.hpp:44 */
int kicp_map_empty(const kicp_map *map, int *empty);        /* Empty()   VoxelHashMap.hpp:45 */
int kicp_map_size(const kicp_map *map, size_t *n_voxels, size_t *n_points);
/* AddPoints(points)  VoxelHashMap.cpp:97-119 */
int kicp_map_add_points(kicp_map *map, const double *xyz, size_t n);
/* RemovePointsFarFromLocation(origin)  VoxelHashMap.cpp:121-132 */
int kicp_map_remove_far(kicp_map *map, const double origin[3]);
/* Update(points, origin)  VoxelHashMap.cpp:83-87 */
int kicp_map_update_origin(kicp_map *map, const double *xyz, size_t n, const double origin[3]);
/* Update(points, pose)  VoxelHashMap.cpp:89-95 (pose: row-major 4x4) */
int kicp_map_update_pose(kicp_map *map, const double *xyz, size_t n, const double pose[16]);
/* Pointcloud()  VoxelHashMap.cpp:72-81.  Writes at most cap points, *n = total points.
 * Order: pool order (the reference's is hash-bucket order; both unspecified). */
int kicp_map_pointcloud(const kicp_map *map, double *out_xyz, size_t cap, size_t *n);
/* GetClosestNeighbor(query)  VoxelHashMap.cpp:46-70, batched over nq queries.
 * nn_xyz[i] = (0,0,0) and dist[i] = DBL_MAX when the 27-neighbourhood is empty. */
int kicp_map_closest_neighbor(const kicp_map *map, const double *query_xyz, size_t nq,
                              double *nn_xyz, double *dist);

/* ------------------------------------------------------------------------------------------
 * Registration -- replaces kiss_icp::Registration (core/Registration.hpp:33-45,
 * core/Registration.cpp:55-167) / pybind _Registration (kiss_icp_pybind.cpp:90-106).
 * ---------------------------------------------------------------------------------------- */
typedef struct kicp_registration kicp_registration;

typedef struct kicp_icp_stats {
    int32_t iterations;       /* ICP iterations executed (0 when the map was empty) */
    int32_t converged;        /* dx.norm() < convergence_criterion ended the loop */
    uint64_t n_source;        /* N_src */
    uint64_t n_corr_last;     /* correspondences in the last iteration */
    uint64_t points_examined; /* E: map points examined, summed over iterations */
    uint64_t n_corr_total;    /* correspondences summed over iterations */
    double kernel_ms;         /* device time of the ICP kernel (hipEvent), 0 if not timed */
} kicp_icp_stats;

/* Registration(max_num_iteration, convergence_criterion, max_num_threads)
 * Registration.cpp:126-136.  max_num_threads is accepted for signature compatibility and
 * ignored (the device schedules its own parallelism). */
int kicp_registration_create(int max_num_iterations, double convergence_criterion,
                             int max_num_threads, int device_id, kicp_registration **out);
int kicp_registration_destroy(kicp_registration *reg);
/* AlignPointsToMap(frame, voxel_map, initial_guess, max_correspondence_distance,
 * kernel_scale)  Registration.cpp:138-167.  map must live on the same device.  stats may be
 * NULL. */
int kicp_align_points_to_map(kicp_registration *reg, const double *frame_xyz, size_t n,
                             const kicp_map *map, const double initial_guess[16],
                             double max_correspondence_distance, double kernel_scale,
                             double T_out[16], kicp_icp_stats *stats);

/* BuildLinearSystem (Registration.cpp:80-121) as accumulated by the LAST iteration of the most recent
 * kicp_align_points_to_map on this handle: JTJ row-major 6x6, JTr 6 (the reference solves
 * JTJ dx = -JTr), and the number of correspondences that went into them.  With
 * max_num_iterations = 1 this is the system of the initial guess. */
int kicp_registration_last_system(const kicp_registration *reg, double JTJ[36], double JTr[6],
                                  uint64_t *n_corr);

/* ------------------------------------------------------------------------------------------
 * Free functions of the stages either side of the path ("next" rows of SURVEY.md section 8f)
 * ---------------------------------------------------------------------------------------- */
/* VoxelDownsample(frame, voxel_size)  core/VoxelUtils.cpp:7-21 / pybind _voxel_down_sample.
 * Keeps the first point per voxel; the survivors leave in the REFERENCE's order -- the bucket order of the
 * tsl::robin_map 1.4.0 it collects them in (VoxelUtils.cpp:17-19) -- unless option "downsample_order" is 0
 * (ascending original index).  out_xyz must hold n points. */
int kicp_voxel_downsample(const double *xyz, size_t n, double voxel_size, int device_id,
                          double *out_xyz, size_t *n_out);
/* Preprocessor(max_range, min_range, deskew, max_num_threads).Preprocess(frame, timestamps,
 * relative_motion)  core/Preprocessing.cpp:40-95 / pybind _Preprocessor._preprocess.
 * out_xyz must hold n points. */
int kicp_preprocess(const double *xyz, size_t n, const double *timestamps, size_t n_timestamps,
                    const double relative_motion[16], double max_range, double min_range,
                    int deskew, int device_id, double *out_xyz, size_t *n_out);

/* ------------------------------------------------------------------------------------------
 * pipeline::KissICP -- replaces kiss_icp::pipeline::KissICP (pipeline/KissICP.hpp:36-96,
 * pipeline/KissICP.cpp:35-75).  All per-frame state (poses, adaptive threshold, local map,
 * intermediate clouds) stays in HBM; only the raw scan goes in and the pose comes out.
 * ---------------------------------------------------------------------------------------- */
typedef struct kicp_config { /* KISSConfig  pipeline/KissICP.hpp:36-54, same defaults */
    double voxel_size;           /* 1.0   */
    double max_range;            /* 100.0 */
    double min_range;            /* 0.0   */
    int max_points_per_voxel;    /* 20    */
    double min_motion_th;        /* 0.1   */
    double initial_threshold;    /* 2.0   */
    int max_num_iterations;      /* 500   */
    double convergence_criterion;/* 1e-4  */
    int max_num_threads;         /* 0 (ignored) */
    int deskew;                  /* 1     */
} kicp_config;
int kicp_config_default(kicp_config *cfg);

typedef struct kicp_pipeline kicp_pipeline;

typedef struct kicp_frame_stats {
    uint64_t n_raw, n_preprocessed, n_frame_downsample, n_source;
    uint64_t map_voxels;
    double sigma;             /* adaptive threshold used for this frame */
    kicp_icp_stats icp;
} kicp_frame_stats;

enum { /* kicp_pipeline_output: which cloud of the last frame */
    KICP_OUT_PREPROCESSED = 0, /* RegisterFrame's first return value  (KissICP.cpp:67) */
    KICP_OUT_SOURCE = 1,       /* RegisterFrame's second return value (KissICP.cpp:67) */
    KICP_OUT_FRAME_DOWNSAMPLE = 2
};

int kicp_pipeline_create(const kicp_config *cfg, int device_id, kicp_pipeline **out);
int kicp_pipeline_destroy(kicp_pipeline *p);
/* RegisterFrame(frame, timestamps)  KissICP.cpp:35-68 on host buffers (pageable memory is fine:
 * std::vector<Eigen::Vector3d>::data(), a numpy array).  Blocks until the frame's pose is available;
 * the two clouds RegisterFrame returns stay in HBM until kicp_pipeline_output() asks for them.
 * timestamps may be NULL / n_timestamps 0.  Equivalent to kicp_pipeline_register_frame_async() +
 * kicp_pipeline_sync(). */
int kicp_pipeline_register_frame(kicp_pipeline *p, const double *xyz, size_t n,
                                 const double *timestamps, size_t n_timestamps);
/* RegisterFrame exactly as the reference declares it: blocks AND returns both clouds (KissICP.cpp:67: the
 * preprocessed frame and the source).  pre_out / src_out hold pre_cap / src_cap points (n points each is always
 * enough); *n_pre / *n_src = points the clouds have.  The preprocessed frame is downloaded while the
 * registration is still running, so this costs about what kicp_pipeline_register_frame costs. */
int kicp_pipeline_register_frame_outputs(kicp_pipeline *p, const double *xyz, size_t n,
                                         const double *timestamps, size_t n_timestamps,
                                         double *pre_out, size_t pre_cap, size_t *n_pre,
                                         double *src_out, size_t src_cap, size_t *n_src);
/* ... for hosts that build their own containers from the result (std::vector's range constructor, a JNI array
 * copy): the two clouds stay in pinned host memory OWNED BY THE PIPELINE and *pre_view / *src_view point into it,
 * valid until the next kicp_pipeline_register_frame_outputs / _views, kicp_pipeline_output or kicp_pipeline_destroy on this
 * pipeline -- the only entries that write or reallocate that memory.  The getters (kicp_pipeline_pose, _delta,
 * _last_stats, _host_stats, _synced_poses, _map) leave it alone: a wrapper may read the pose before it copies the
 * clouds out, as kiss_icp::pipeline::KissICP::RegisterFrame in kiss-icp_amd/cpp does.  One copy fewer than the entry above. */
int kicp_pipeline_register_frame_views(kicp_pipeline *p, const double *xyz, size_t n,
                                       const double *timestamps, size_t n_timestamps,
                                       const double **pre_view, size_t *n_pre,
                                       const double **src_view, size_t *n_src);
/* The second half of RegisterFrame-with-its-return-value for a host whose result containers must be ALLOCATED per call
 * (std::vector<Eigen::Vector3d> by value: KissICP.cpp:35-68, consumed at ros/src/OdometryServer.cpp:162-165): queue the scan with
 * kicp_pipeline_register_frame_async -- the only frame in flight --, allocate while the device works (3 MB of fresh pages
 * are ~0.3 ms of page faults: as long as the whole registration), then call this: it waits for the frame, has the
 * preprocessed cloud in pre_out (at most pre_cap points; downloaded and spread while the registration was running) and the
 * source cloud behind *src_view, in pinned memory of the pipeline (valid as for kicp_pipeline_register_frame_views).
 * KICP_ERR_INVALID_ARG unless exactly one frame is in flight. */
int kicp_pipeline_collect_outputs(kicp_pipeline *p, double *pre_out, size_t pre_cap, size_t *n_pre,
                                  const double **src_view, size_t *n_src);
/* The same without waiting: the scan is copied into pinned staging memory (a few helper threads,
 * option "staging_threads"), so the caller's buffers are free again when the call returns; its upload and
 * the stages in front of the registration then run on a second stream UNDER the previous frame's
 * registration, and the frame is queued behind it (frame k+1 consumes frame k's pose and map on the
 * device).  float64 scans whose values are all exactly representable in float32 -- anything read from a
 * float32 sensor file (python/kiss_icp/datasets/kitti.py:66) -- are narrowed for the upload (option
 * "staging_f32"): half the bytes over PCIe, bit-identical points on the device.  Results:
 * kicp_pipeline_sync() + kicp_pipeline_synced_poses() / kicp_pipeline_pose().  This is the entry the
 * throughput figure of bench.py is measured on. */
int kicp_pipeline_register_frame_async(kicp_pipeline *p, const double *xyz, size_t n,
                                       const double *timestamps, size_t n_timestamps);
/* ... for callers that hold the sensor's native float32 points (ROS PointCloud2:
 * ros/src/Utils.hpp:201-205, KITTI .bin files): no widening on the host at all */
int kicp_pipeline_register_frame_async_f32(kicp_pipeline *p, const float *xyz, size_t n,
                                           const double *timestamps, size_t n_timestamps);
/* Same, with the scan (and timestamps) already resident in this device's HBM.  Enqueues the
 * frame on the pipeline's stream and returns without waiting; frames may be queued
 * back-to-back (frame k+1 consumes frame k's pose on the device).  The buffers must stay
 * valid until kicp_pipeline_sync(). */
int kicp_pipeline_register_frame_device(kicp_pipeline *p, const double *d_xyz, size_t n,
                                        const double *d_timestamps, size_t n_timestamps);
int kicp_pipeline_sync(kicp_pipeline *p);
/* poses (row-major 4x4 each) of the frames completed by the most recent kicp_pipeline_sync /
 * kicp_pipeline_register_frame, oldest first; *n_frames = how many there are */
int kicp_pipeline_synced_poses(kicp_pipeline *p, double *T_out, size_t cap_frames,
                               size_t *n_frames);
/* pose() / delta()  KissICP.hpp:80-84 (getters and setters for the mutable refs) */
int kicp_pipeline_pose(kicp_pipeline *p, double T[16]);
int kicp_pipeline_delta(kicp_pipeline *p, double T[16]);
int kicp_pipeline_set_pose(kicp_pipeline *p, const double T[16]);
int kicp_pipeline_set_delta(kicp_pipeline *p, const double T[16]);
/* VoxelMap()  KissICP.hpp:77-78: borrowed handle, owned by the pipeline */
int kicp_pipeline_map(kicp_pipeline *p, kicp_map **map);
/* LocalMap()  KissICP.hpp:75 == kicp_map_pointcloud(kicp_pipeline_map()) */
/* clouds of the last registered frame */
int kicp_pipeline_output_size(kicp_pipeline *p, int which, size_t *n);
int kicp_pipeline_output(kicp_pipeline *p, int which, double *out_xyz, size_t cap, size_t *n);
/* Voxelize(frame)  KissICP.cpp:70-75: (source, frame_downsample) of an arbitrary cloud */
int kicp_pipeline_voxelize(kicp_pipeline *p, const double *xyz, size_t n, double *source_xyz,
                           size_t *n_source, double *frame_downsample_xyz,
                           size_t *n_frame_downsample);
int kicp_pipeline_last_stats(kicp_pipeline *p, kicp_frame_stats *stats);
/* device time (ms, hipEvents on the pipeline's stream) of the ICP kernel over all frames since
 * the last reset, its launch count and iteration count -- for roofline accounting */
int kicp_pipeline_icp_timing(kicp_pipeline *p, double *total_ms, uint64_t *launches,
                             uint64_t *iterations, uint64_t *algorithmic_bytes, int reset);
/* shader-clock cycles workgroup 0 spent in the four phases of the LAST ICP launch
 * ([0] association + accumulation, [1] workgroup reduction + publish, [2] cross-workgroup gather,
 * [3] 6x6 solve + pose update) and the number of workgroups that took part */
int kicp_pipeline_icp_profile(kicp_pipeline *p, uint64_t cycles[4], int *workgroups);
/* duration of the LAST ICP launch as seen by its workgroup 0: shader-clock cycles (s_memtime) and
 * 100 MHz wall ticks (s_memrealtime); cycles / ticks * 100 = effective shader clock in MHz */
int kicp_pipeline_icp_clock(kicp_pipeline *p, uint64_t *cycles, uint64_t *ticks);
/* the LAST ICP launch as seen by its workgroup 0, in 100 MHz wall ticks: time to the end of its first
 * iteration (which stages the neighbourhood windows from HBM), total time, iterations executed */
int kicp_pipeline_icp_first_iteration(kicp_pipeline *p, uint64_t *ticks_first, uint64_t *ticks_total,
                                      int *iterations);
/* per-iteration profile of the LAST ICP launch (at most the first 24 iterations), 6 uint32 per
 * iteration in 10 ns ticks: workgroup 0's {associate, publish, gather, solve}, the slowest
 * 32-lane group's associate time over all workgroups, gather polling passes of workgroup 0 */
int kicp_pipeline_icp_iteration_profile(kicp_pipeline *p, uint32_t *out, int cap_iters, int *n_iters);
/* per-GROUP records of the LAST ICP launch ("icp_profile" = 1 only): for each of the first *n_iters
 * iterations and each of the *n_groups 32-lane groups, 4 uint32: {wait-in | transform+window test << 16,
 * window fill | scan << 16} in 10 ns ticks, {staged points | examined points << 16}, path (0 staged
 * 27-voxel window, 1 staged widened window, 2 window staged in this iteration, 3 direct HBM search).
 * out must hold n_iters * n_groups * 4 words (cap_words) */
int kicp_pipeline_icp_group_profile(kicp_pipeline *p, uint32_t *out, size_t cap_words, int *n_iters,
                                    int *n_groups);
/* Where the HOST side of the queued frames spent its time (since the last reset): the asynchronous entries are
 * meant to return without ever waiting for the device; every wait they did take is counted here, so that a test
 * (tests/test_gpu_paths.py) and bench.py can hold them to it.  KICP_HOST_TRACE=1 in the environment additionally
 * prints one line per queued frame on stderr. */
typedef struct kicp_host_stats {
    uint64_t frames;            /* frames queued */
    uint64_t backpressure_waits;/* waits because "queue_depth" frames were already queued on the device (the caller
                                   is ahead of the device: benign, the device has work) */
    uint64_t capacity_waits;    /* waits for a queued frame to finish so that the map's growth could be bounded */
    uint64_t staging_waits;     /* waits for a pinned staging slot whose upload had not finished */
    uint64_t ring_syncs;        /* full drains because the frame-record ring (256 frames in flight) was full */
    uint64_t counter_refreshes; /* blocking read-backs of the map counters (a stream synchronisation each) */
    uint64_t map_grows;         /* reallocations of the map's slot array or block pool */
    uint64_t map_rehashes;      /* rebuilds of the slot array (growth or tombstone clean-up) */
    uint64_t buffer_grows;      /* reallocations of the frame buffers (a scan larger than any before) */
    double stage_ms;            /* host time copying scans into pinned memory */
    double enqueue_ms;          /* host time queueing copies and kernels */
    double backpressure_ms;     /* host time in back-pressure waits */
    double wait_ms;             /* host time blocked in every OTHER wait counted above (these starve the device) */
    double device_gap_ms;       /* device time between the end of a registration and the start of the next, summed over
                                   the frames collected by kicp_pipeline_sync since the last reset (map update + front
                                   of the next registration; a starved device shows up here) */
    double max_device_gap_ms;   /* ... the largest single one */
    double max_call_ms;         /* the longest single call of an asynchronous entry */
    /* placement of the host side (not counters: never reset).  -1 = unknown / not allocated yet */
    int32_t device_numa_node;   /* NUMA node of the pipeline's GPU (sysfs, by PCI address) */
    int32_t staging_numa_node;  /* node the pinned staging slots lie on */
    int32_t staging_helpers;    /* helper threads of this pipeline ("staging_threads", capped by the usable CPUs) */
    int32_t helpers_bound;      /* ... of which run restricted to the GPU's node */
} kicp_host_stats;
int kicp_pipeline_host_stats(kicp_pipeline *p, kicp_host_stats *stats, int reset);
/* the HIP stream (hipStream_t) the pipeline launches on, as an opaque pointer */
int kicp_pipeline_stream(kicp_pipeline *p, void **stream);

/* ------------------------------------------------------------------------------------------
 * Multi-stream batch mode (new functionality; the reference registers one stream in one process).
 * S independent LiDAR streams, one pipeline + one local map per GPU, one WORKER THREAD PER STREAM inside
 * this library (each bound to its device with hipSetDevice), no collective on the data path: frame k of a
 * stream needs pose k-1 and the map holding frame k-1 (pipeline/KissICP.cpp:47,61), so streams are the unit
 * of parallelism.  The one exchange is the "pose-graph sync" of kicp_batch_sync(): an all-gather of the poses
 * each stream completed since the previous sync, after which every rank holds all S trajectories.
 *
 * Ranks: a batch handle owns the n_local consecutive global ranks first_rank .. first_rank + n_local - 1 out of
 * n_total.  All GPUs of a node in one process: n_local == n_total, first_rank 0.  One process per GPU
 * (torchrun / MPI): n_local 1, first_rank = the process's rank, and the 128-byte id made by
 * kicp_batch_unique_id() on one rank and distributed by the launcher's own means.
 *
 * The exchange: comm == NULL uses RCCL called directly (ncclCommInitRank / ncclAllGather from librccl,
 * resolved with dlopen) on a stream of its own; one fixed-size block of 32 + 128 * frames_per_gather bytes per
 * rank and gather.  A block carries its rank's frame count and pipeline status, so several gathers follow when
 * ANY rank completed more frames than a block holds (ranks need not queue the same number of frames), and a
 * rank whose pipeline failed still takes part -- every process then returns that rank's status from
 * kicp_batch_sync instead of waiting in the collective for ever.  A host may supply its own communicator (MPI,
 * a test stub) instead.
 * ---------------------------------------------------------------------------------------- */
#define KICP_BATCH_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */
typedef struct kicp_batch kicp_batch;
typedef struct kicp_batch_comm {
    void *ctx;
    /* once per local rank, on that rank's worker thread, after its pipeline exists; may be NULL */
    int (*init)(void *ctx, int rank, int n_ranks, int device_id);
    /* gather `bytes` bytes from d_send of every rank into d_recv (n_ranks * bytes, rank order), ordered on the
     * hipStream_t `stream`; both buffers are device memory of this rank's GPU; called concurrently by the
     * local ranks' threads */
    int (*all_gather)(void *ctx, int rank, const void *d_send, void *d_recv, size_t bytes, void *stream);
    int (*finalize)(void *ctx, int rank); /* may be NULL */
} kicp_batch_comm;
int kicp_batch_unique_id(unsigned char id[KICP_BATCH_ID_BYTES]);
/* devices[n_local]: the GPU of each local stream (the same device may appear more than once);
 * unique_id: NULL when n_local == n_total; frames_per_gather: 0 = 64 */
int kicp_batch_create(const kicp_config *cfg, const int *devices, int n_local, int first_rank, int n_total,
                      const unsigned char *unique_id, const kicp_batch_comm *comm, size_t frames_per_gather,
                      kicp_batch **out);
int kicp_batch_destroy(kicp_batch *b);
/* one host scan per local stream (xyz[i] == NULL: stream i has no frame this round), handed to that
 * stream's kicp_pipeline_register_frame_async by its worker thread; returns when every scan has been staged
 * (the buffers are free again), not when it has been registered */
int kicp_batch_register_frames(kicp_batch *b, const double *const *xyz, const size_t *n,
                               const double *const *timestamps, const size_t *n_timestamps);
int kicp_batch_register_frames_f32(kicp_batch *b, const float *const *xyz, const size_t *n,
                                   const double *const *timestamps, const size_t *n_timestamps);
/* wait for every queued frame of every local stream, then all-gather the new poses */
int kicp_batch_sync(kicp_batch *b);
/* poses (row-major 4x4) that GLOBAL rank `rank` completed between the last two syncs, oldest first */
int kicp_batch_poses(kicp_batch *b, int rank, double *T_out, size_t cap_frames, size_t *n_frames);
/* borrowed handle of a local stream's pipeline (map, outputs, stats); use it between syncs only */
int kicp_batch_pipeline(kicp_batch *b, int local_stream, kicp_pipeline **pipe);
/* wall time the last kicp_batch_sync spent in the pose exchange (upload, all-gather, download; all ranks) */
int kicp_batch_gather_seconds(kicp_batch *b, double *seconds);

/* ------------------------------------------------------------------------------------------
 * Device-memory helpers for hosts without a HIP binding of their own (cgo / JNI / ctypes): enough
 * to stage scans in HBM for kicp_pipeline_register_frame_device.  Plain hipMalloc / hipMemcpy.
 * ---------------------------------------------------------------------------------------- */
int kicp_device_alloc(int device_id, size_t bytes, void **d_ptr);
int kicp_device_free(int device_id, void *d_ptr);
int kicp_device_upload(int device_id, void *d_dst, const void *h_src, size_t bytes);
int kicp_device_download(int device_id, void *h_dst, const void *d_src, size_t bytes);
/* waits, with the library's deadline (wait_timeout_ms), for what is queued at the moment of the call on the library's own streams
 * of the device AND on its NULL stream; work on streams the caller created itself is the caller's to wait for */
int kicp_device_synchronize(int device_id);

/* Device self-test: n systems A x = b (A row-major 6x6, symmetric) solved by the ICP kernel's own solver (the
 * scalar statement of Eigen 3.4.0's pivoted LDLT that Registration.cpp:156 calls), so that its arithmetic can be
 * compared with the oracle's on the same inputs -- pivot order, zero-pivot rule and all -- without a registration
 * around it (tests/test_gpu_paths.py). */
int kicp_selftest_solve(int device_id, const double *A, const double *b, size_t n, double *x);
/* Device self-test: the spatial order k_icp takes its runs from (kicp_sort.hip) -- keys[i] = {30-bit Morton code of the
 * 2-voxel cell of a point << 24 | index of the point}, ascending, for the n points of xyz, computed as the pipeline
 * computes it: the count lives on the device, the host knows only a bound (n_bound >= n, at most 2^24) and a hint (n_hint,
 * 0 = none) that decides how the sorted runs of 2048 keys are merged -- never the result. */
int kicp_selftest_tile_sort(int device_id, const double *xyz, size_t n, double voxel_size, size_t n_hint, size_t n_bound, uint64_t *keys);
/* Host self-test (no device needed): the float64 -> float32 narrowing the host-input entries apply to a scan before
 * its upload (option "staging_f32").  dst receives the narrowed values; *exact = 1 iff every value survives the
 * round trip, which is the only case in which a scan travels as float32 (one NaN, one value with more than 24
 * significant bits, one value outside float32's range: the scan goes up as float64). */
int kicp_selftest_narrow(const double *src, size_t count, float *dst, int *exact);

/* ------------------------------------------------------------------------------------------
 * tuning knobs (process-wide; read when a handle is created).  Unknown names are an error.
 *   "icp_blocks"      workgroups taking part in the persistent ICP kernel (0 = derive from N_src
 *                     on the device: ceil(N_src / (16 * icp_points_per_group)), at most 256)
 *   "icp_points_per_group"  target source points per 32-lane group and iteration (default 1)
 *   "icp_weight_base", "icp_weight_quad", "icp_weight_dense_min", "icp_weight_dense_div", "icp_weight_long_base"
 *                     the workgroups of the ICP kernel take contiguous runs of the spatially sorted source cloud of equal
 *                     WEIGHT; a point weighs  base + c + c^2 / quad + max(0, E - dense_min) / dense_div,  c = population
 *                     of the map voxel it falls in under the initial guess, E = population of the 27 voxels around it.
 *                     Defaults 128, -1 (= 10 when the cloud has at most 64 points per workgroup, else no quadratic term),
 *                     200, 1 (dense_div 0 switches the last term off).  Clouds of more than 64 points per workgroup (the
 *                     1M-point / 0.1 m configuration) use  "icp_weight_long_base" (128) + c + "icp_weight_long_emul" (1) * E  instead.  The weights decide nothing but which workgroup
 *                     serves which points -- hence the order of the sums, deterministically (integer arithmetic on data).
 *                     What they are tuned for: no run's voxel neighbourhood may outgrow a workgroup's LDS (~5.3 k points:
 *                     the densest runs near the sensor), and no run may need many more 16-point rounds than the others
 *   "icp_use_lds"     1 = stage each query's candidate voxels in LDS and reuse them across ICP
 *                     iterations (default 1)
 *   "icp_timing"      1 = bracket every ICP launch with hipEvents (default 1)
 *   "icp_lds_kib"     LDS per ICP workgroup in KiB, 96..160 (0 = all 160: one workgroup per CU)
 *   "icp_reserve_cus" CUs left out of the ICP grid for the front stages of the next frame, which run
 *                     concurrently on a second stream (default 32 = one per shader engine: workgroups are
 *                     handed to the shader engines in turn, so a kernel whose next workgroup falls on a
 *                     full engine waits even when other engines have room; the grid is the device's
 *                     co-resident maximum minus this)
 *   "staging_threads" helper threads (besides the caller) that copy a host scan into pinned memory (default 3)
 *   "staging_f32"     1 = narrow float64 host scans to float32 for the upload when lossless (default 1)
 *   "staging_zero_copy"  1 (default) = no upload call: the two kernels that read the raw scan fetch it over PCIe from
 *                     the pinned, device-mapped staging slot themselves; 0 = hipMemcpyAsync into HBM first
 *   "stage_in"        1 (default): a frame that DESKEWS has its scan and timestamps copied from the staging slot into HBM by a kernel
 *                     queued in front of the wait for the previous pose -- with deskewing the front stages sit on the frame's
 *                     serial chain, and reading the scan over PCIe there cost 30 us per frame; 0 = they read the slot themselves.
 *                     Same points, same poses.
 *   "staging_numa"    placement of the host side on the NUMA node the GPU hangs off.  1 (default): the node the pinned staging slots
 *                     landed on is looked up (move_pages) and reported next to the GPU's (kicp_host_stats) -- on every box this
 *                     library has run on ROCm's pinned allocation was on the GPU's node already; 2 = slots that are NOT are
 *                     replaced by node-bound pages registered with the runtime, and the helper threads (and a batch's worker
 *                     threads) are restricted to that node's CPUs (opt-in: the replacement has never had a box to run on, and
 *                     bound threads measured the same on one GPU, 2955 against 2958 scans/s, while a thread that may run anywhere
 *                     gets out of a busy neighbour's way -- boxes are shared); 0 = nothing is looked up.  Without NUMA
 *                     information (one node, a container that hides it) nothing changes.
 *   "staging_numa_pretend"  test hook (default -1 = off): n >= 0 declares NUMA node n to be the one every GPU hangs off, so that on a
 *                     one-GPU box -- where the runtime's pinned allocation already lies on the GPU's real node -- level 2 of
 *                     "staging_numa" has slots to replace: mbind + hipHostRegister + bound helpers run, and kicp_host_stats reports
 *                     the node the replaced slots landed on (tests/test_gpu_paths.py::test_host_placement_moves_the_slots).
 *   "queue_depth"     frames an asynchronous entry keeps queued on the device before it waits for the oldest (default 4,
 *                     >= 2; 0 = no limit: the host may run ahead until the 256-frame record ring is full)
 *   "collective_timeout_ms"  kicp_batch_*: how long a step that waits for PEERS may take -- the communicator's rendezvous
 *                     (ncclCommInitRank returns when ALL ranks have joined) and a pose all-gather through a host communicator --
 *                     before kicp_batch_create / kicp_batch_sync return KICP_ERR_TIMEOUT (default 1800000 = 30 min: a box's first
 *                     RCCL initialisation has been seen to take 5; 0 = never).  The batch is then broken -- the threads inside the
 *                     exchange cannot be cancelled --: every later call says so, and kicp_batch_destroy shuts down what responds,
 *                     LEAKS the handle and returns KICP_ERR_TIMEOUT.  (An all-gather through RCCL waits on the stream and has the
 *                     device deadline, "wait_timeout_ms", like every other wait.)  Read when the batch is created.
 *   "relaxed_backpressure"  1 (default): that wait sleeps between its polls (40 us at a time) -- the caller is frames ahead of the
 *                     device, nobody waits for a result, and a polling loop costs a core per stream; 0 = poll closely, as the waits
 *                     for a RESULT do (kicp_pipeline_sync, the blocking entries)
 *   "downsample_order"  order in which VoxelDownsample emits its survivors: 1 (default) = the reference's, i.e. the bucket
 *                     order of the tsl::robin_map 1.4.0 it collects them in (VoxelUtils.cpp:7-21); 0 = ascending point index
 *                     (rounds 1-2).  The order decides which points AddPoints' first-come rule and the second downsample
 *                     keep: the two trajectories differ by centimetres
 *   "icp_bulk_fill"   1 (default): in a registration's first iteration the workgroup establishes all its queries' windows
 *                     together (distinct cells, one wave of map lookups, one of point fetches); 0: query by query, as in later
 *                     iterations.  Results are bitwise the same either way.
 *   "icp_schur_solve"  1 (default): the 6 x 6 normal equations of a Gauss-Newton step, whose top-left block is (sum w) I, are solved
 *                     through their 3 x 3 Schur complement when that is well conditioned (pivots above 1e-9 of the diagonal);
 *                     0: always by the pivoted 6 x 6 LDLT of Eigen that the reference calls (Registration.cpp:156).  The two
 *                     differ by rounding (poses ~1e-13 apart); rank-deficient systems always take the LDLT and its zero-pivot rule.
 *                     The default is a DELIBERATE departure from the reference's arithmetic (it can move the |dx| < 1e-4 stopping
 *                     test by one iteration on a borderline frame); set 0 where the reference's own solve is wanted -- the test suite
 *                     holds that path to the CPU oracle (tests/test_gpu_parity.py::test_registration_with_the_references_own_solve).
 *   "icp_wide"        form of the registration's association phase: 0 = a 32-lane group per source point (a few dozen points
 *                     per workgroup, neighbourhoods of hundreds of map points: full-size voxels); 1 = a thread per source
 *                     point (hundreds of points per workgroup: small voxels, large clouds); -1 (default) = by the size of
 *                     the cloud (more than 64 points per workgroup -> 1).  The pose is bitwise the same either way.
 *   "icp_wide_prune"  thread-per-point form: 0 = visit every occupied voxel of the 27; 1 = skip voxels whose box lies farther
 *                     than the best candidate / the correspondence threshold; 2 (default) = also bounded by the previous
 *                     iteration's neighbour.  Exact: skipped voxels lose every comparison of VoxelHashMap.cpp:58-63 anyway.
 *   "icp_group_stable"  group form (workgroups of at most 64 points, which keep a scan list per point): 1 (default) = a source point
 *                     that has stayed in its voxel and whose last neighbour is still provably closer than every other candidate of
 *                     its list (the list scan's second smallest distance, minus how far the point has moved since) keeps that
 *                     neighbour without a search; 0 = every point is searched in every iteration.  Exact either way: pose,
 *                     iterations, correspondences and examined points are bitwise the same (IcpQueryMeta::Lr).
 *   "icp_wide_stable"  thread-per-point form: 1 (default) = a source point that has stayed in its voxel and whose last neighbour is
 *                     still provably closer than any other map point (a bound kept from its last search, minus how far the
 *                     point has moved) keeps that neighbour without a search, and the searches that remain run on a few lanes;
 *                     0 = every point is searched in every iteration.  Exact either way (kicp_icp_wide.hpp, WideQuery::Lr).
 *   "icp_wide_prefill"  thread-per-point form: eighths (0 .. 8, default 0) of the workgroup's LDS point store that the first
 *                     iteration's window phase fills; the rest is filled by the searches with the voxels they really read
 *   "icp_wide_promote_from"  ... from this iteration on (default 1: the first iteration's reads are the widest, not the lasting ones)
 *   "icp_wide_per_round"  ... items a thread files per round of the voxel queue (default 4; the rest is held against the answers)
 *   "icp_wide_load_eighths"  ... eighths of the workgroup's voxel table that may fill (2 .. 7, default 5)
 *   "frame_events"    pipelines created afterwards: 1 = the events that order a frame's buffers, tell the front stages that the
 *                     pose is ready and time the registration are attached to the dispatches of the registration and of the last
 *                     map kernel (hipExtLaunchKernel) instead of being recorded between the kernels of the serial chain (0, the
 *                     default).  Same results; kept as the record of an experiment: a dispatch that carries a completion signal
 *                     costs more than the recorded event it saves (0.387 against 0.360 ms per frame).
 *   "icp_wide_group_max"  thread-per-point form: when at most this many points of a workgroup need a search in an iteration (the
 *                     later iterations: most keep their neighbour, see icp_wide_stable), each is searched by a 32-lane group
 *                     reading all 27 voxels, without the voxel queues (0 .. 512, default 128; 0 = always the queues).  Same answers.
 *   "icp_wide_flat"   thread-per-point form: how the queued voxels are read -- bit 0: the map's queue, bit 1: the LDS store's queue
 *                     by a thread per POINT of all queued voxels at once (minima settled by LDS atomics) instead of a 32-lane
 *                     group per voxel (0 .. 3, default 3: 15 % off the registration of the 1M-point
 *                     configuration, profiles/r04_an_flat_sweep.txt).  Same answers, ties included.
 *   "icp_device_streams"  batch mode on ONE GPU (default 1): pipelines created afterwards share their device with this many
 *                     streams in all -- each registers with 1 / n of the co-resident workgroups, and up to n registrations
 *                     of different pipelines run side by side instead of one behind the other.  A single stream gets slower
 *                     (fewer workgroups), the device as a whole faster; kicp_batch_create sets it by itself when a device
 *                     appears more than once in its list.  A pipeline's trajectory depends on its share (the summation
 *                     tree follows the number of workgroups), not on what runs beside it.
 *   "icp_inject_timeout"  test hook: the first N registrations of a pipeline created afterwards behave as if
 *                     their workgroups never became co-resident (exercises the replay path)
 *   "icp_inject_timeout_skip"  ... after leaving its first M registrations alone
 *   "map_rehash_every"  test hook: pipelines rebuild their map's slot array (in stream order, no host wait) every N frames
 *   "wait_timeout_ms"  deadline of every host-side wait for the device, in milliseconds (default 120000; 0 = none).  The library
 *                     never blocks in hipStreamSynchronize / hipEventSynchronize: it polls, and a wait that does not end in time
 *                     returns KICP_ERR_TIMEOUT and names what it waited for.  The reference cannot hang
 *                     (core/Registration.cpp:138-167 terminates, always); neither may its drop-in
 *   "inject_stall_ms"  test hook: the next piece of work any handle queues is preceded by a kernel that occupies its stream this
 *                     long (0 .. 60000; consumed by the first taker) -- the dependency that is not signalled in time
 *                     (tests/test_gpu_deadlines.py)
 *   "map_apply_threads"  workgroup size of the AddPoints apply kernel: 256, 512 (default) or 1024
 *   "icp_weights_kernel"  1 (default) = the weights that cut a source cloud of short runs (at most 64 points per workgroup) into runs of
 *                     equal weight are computed by a kernel of their own in front of the registration launch, and every workgroup
 *                     finds its boundaries from all of them by itself; 0 = by the launch's prologue, with an exchange of the
 *                     workgroups' sums.  The same weights, the same runs, the same pose bit for bit
 *   "sort_by_rank"    1 (default) = a source cloud of a few thousand points (judged by the previous frame's count: up to ~6.5 k) gets
 *                     its spatial order for the registration's workgroups in ONE launch, every key placed by its rank among all;
 *                     0 = always sorted runs + merge passes.  The same order either way (the keys are unique)
 *   "map_fused_update"  1 (default) = local_map_.Update (KissICP.cpp:61; VoxelHashMap.cpp:83-132) is two kernels: the verdicts of
 *                     RemovePointsFarFromLocation are taken beside AddPoints' first kernel and carried out by its second (a
 *                     sentenced voxel receives nothing and goes, a created one is judged by its first point); 0 = a third
 *                     kernel behind them.  The same map either way, voxel for voxel, point for point.  Also kicp_map_update_*
 *   "icp_profile"     1 = launch the ICP kernel variant that records the in-kernel phase timers read
 *                     by kicp_pipeline_icp_profile / _icp_iteration_profile (default 0)
 * ---------------------------------------------------------------------------------------- */
int kicp_set_option(const char *name, long value);

#ifdef __cplusplus
}
#endif
#endif /* KICP_H */
