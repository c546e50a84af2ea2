/*
 * kiss_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A dependency-free plain-C restatement of the KISS-ICP registration hot path and the
 * stages either side of it, written from the reference's sources at /root/reference
 * (PRBonn/kiss-icp v1.2.3).  Every function cites the reference file:line it follows.
 *
 * PINNING.  The reference ships no golden vectors / known-answer tests for this path
 * (python/tests/test_kiss_icp.py:1-4 is an import smoke test) and its own build needs Eigen
 * 3.4.0 / Sophus 1.24.6 / tsl::robin_map 1.4.0 / oneTBB 2022.1.0 FetchContent'ed from the network.
 *  - PINNED against the reference's own code: every line of cpp/kiss_icp/{core,pipeline} (.cpp files)
 *    on this path.  oracle/ref_build/ compiles those files UNMODIFIED from /root/reference
 *    against stand-in third-party headers into oracle/_ref/libkiss_ref.so, and
 *    tests/test_ref_pins_oracle.py holds this restatement to it (bit-exact survivors, map
 *    content, neighbours, thresholds; poses to 1e-11); tests/golden/ *.npz come from it.
 *  - PARITY UNPINNED for the third-party arithmetic only (Eigen's pivoted LDLT and its
 *    zero-pivot rule, Sophus' SE3 exp / log / product branches): those libraries are absent, the
 *    stand-in headers forward these operations to the restatement in this file, which is
 *    written from the published algorithms of the pinned versions and checked against
 *    scipy/numpy in tests/test_oracle.py.  Likewise unpinned: tsl::robin_map's iteration order
 *    (bucket order upstream, insertion order here and in the stand-in), which orders
 *    VoxelDownsample's output and so decides which points AddPoints' order-dependent rule keeps
 *    in a contested voxel.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (kiss-icp_amd/) never links, imports or calls it.
 *
 * Conventions: points are row-major N x 3 f64 (== std::vector<Eigen::Vector3d>::data()).
 * SE(3) elements cross this interface as row-major 4x4 f64 matrices; internally they are
 * {unit quaternion (x,y,z,w), translation}, Sophus' own storage.
 */
#ifndef KISS_ORACLE_H
#define KISS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SE(3) --------------------------------------------------------------------------- */
typedef struct {
    double q[4]; /* x, y, z, w  (Eigen::Quaterniond::coeffs() order) */
    double t[3];
} ko_se3;

void ko_se3_identity(ko_se3 *T);
/* Sophus::SE3d(Matrix4d): Eigen quaternion-from-matrix (Shepperd).  Returns 0, or -1 if R is
 * not orthogonal / det <= 0 (SOPHUS_ENSURE would abort there). */
int ko_se3_from_matrix(const double M[16], ko_se3 *T);
void ko_se3_matrix(const ko_se3 *T, double M[16]);
void ko_se3_mul(const ko_se3 *A, const ko_se3 *B, ko_se3 *out); /* out = A * B */
void ko_se3_inverse(const ko_se3 *A, ko_se3 *out);
void ko_se3_exp(const double a[6], ko_se3 *out); /* a = (upsilon[3], omega[3]) */
void ko_se3_log(const ko_se3 *A, double a[6]);
void ko_se3_act(const ko_se3 *T, const double p[3], double out[3]); /* out = T * p */

/* Eigen::LDLT<Matrix6d>(A).solve(b).  A row-major (symmetric; lower triangle is read). */
void ko_ldlt6_solve(const double A[36], const double b[6], double x[6]);

/* ---- voxel utilities (core/VoxelUtils.hpp:32-51, VoxelUtils.cpp:7-21) ------------------ */
void ko_point_to_voxel(const double p[3], double voxel_size, int32_t v[3]);
/* keep the first point that falls in each voxel.  Output order is DEFINED here as ascending
 * original index (the reference's is tsl::robin_map bucket order -- unspecified).
 * out must hold n points; returns the number kept. */
size_t ko_voxel_downsample(const double *xyz, size_t n, double voxel_size, double *out);
/* output order of ko_voxel_downsample (and of the pipeline's two downsamples): 1 = the reference's, i.e. bucket order
 * of tsl::robin_map 1.4.0 (default); 0 = ascending original index.  Process-wide. */
void ko_set_downsample_order(int order);
int ko_get_downsample_order(void);

/* ---- VoxelHashMap (core/VoxelHashMap.{hpp,cpp}) ---------------------------------------- */
typedef struct ko_map ko_map;
ko_map *ko_map_create(double voxel_size, double max_distance, unsigned max_points_per_voxel);
void ko_map_destroy(ko_map *m);
void ko_map_clear(ko_map *m);
int ko_map_empty(const ko_map *m);
size_t ko_map_num_voxels(const ko_map *m);
size_t ko_map_num_points(const ko_map *m);
void ko_map_add_points(ko_map *m, const double *xyz, size_t n);
void ko_map_remove_far(ko_map *m, const double origin[3]);
void ko_map_update_origin(ko_map *m, const double *xyz, size_t n, const double origin[3]);
void ko_map_update_pose(ko_map *m, const double *xyz, size_t n, const double T[16]);
/* out must hold ko_map_num_points() points; order is storage order (unspecified upstream) */
size_t ko_map_pointcloud(const ko_map *m, double *out);
/* returns distance; nn = (0,0,0) and DBL_MAX when the 27-neighbourhood is empty */
double ko_map_closest_neighbor(const ko_map *m, const double q[3], double nn[3]);
/* same, also counts the map points examined */
double ko_map_closest_neighbor_counted(const ko_map *m, const double q[3], double nn[3],
                                       uint64_t *examined);

/* ---- Registration (core/Registration.cpp:55-167) ---------------------------------------- */
typedef struct {
    int32_t iterations;       /* ICP iterations executed (>=1 unless the map was empty) */
    int32_t converged;        /* 1 if dx.norm() < convergence_criterion ended the loop */
    uint64_t n_source;        /* N_src */
    uint64_t n_corr_last;     /* correspondences in the last iteration */
    uint64_t points_examined; /* E summed over all iterations */
    uint64_t n_corr_total;    /* correspondences summed over all iterations */
} ko_icp_stats;

/* max_threads <= 0 -> all cores (Registration.cpp:126-136).  Returns 0. */
int ko_align_points_to_map(const double *frame_xyz, size_t n, const ko_map *m,
                           const double T_guess[16], double max_correspondence_distance,
                           double kernel_scale, int max_num_iterations,
                           double convergence_criterion, int max_threads, double T_out[16],
                           ko_icp_stats *stats);
/* one BuildLinearSystem pass on already-transformed source points (for unit tests):
 * JTJ row-major 36, JTr 6 */
void ko_build_linear_system(const double *source_xyz, size_t n, const ko_map *m,
                            double max_correspondence_distance, double kernel_scale,
                            double JTJ[36], double JTr[6], uint64_t *n_corr);

/* ---- Preprocessor (core/Preprocessing.cpp:55-95) ---------------------------------------- */
/* timestamps may be NULL / n_ts == 0.  out must hold n points; returns the number kept,
 * or (size_t)-1 when 0 < n_ts < n (std::vector::at would throw). */
size_t ko_preprocess(const double *xyz, size_t n, const double *timestamps, size_t n_ts,
                     const double relative_motion[16], double max_range, double min_range,
                     int deskew, int max_threads, double *out);

/* ---- AdaptiveThreshold (core/Threshold.{hpp,cpp}) ---------------------------------------- */
typedef struct {
    double min_motion_threshold, max_range, model_sse;
    int num_samples;
} ko_threshold;
void ko_threshold_init(ko_threshold *t, double initial_threshold, double min_motion_threshold,
                       double max_range);
double ko_threshold_compute(const ko_threshold *t);
void ko_threshold_update(ko_threshold *t, const double model_deviation[16]);

/* ---- pipeline::KissICP (pipeline/KissICP.{hpp,cpp}) --------------------------------------- */
typedef struct {
    double voxel_size, max_range, min_range;
    int max_points_per_voxel;
    double min_motion_th, initial_threshold;
    int max_num_iterations;
    double convergence_criterion;
    int max_num_threads;
    int deskew;
} ko_config;
void ko_config_default(ko_config *c);

typedef struct ko_pipeline ko_pipeline;
ko_pipeline *ko_pipeline_create(const ko_config *c);
void ko_pipeline_destroy(ko_pipeline *p);
/* RegisterFrame (KissICP.cpp:35-68).  Returns 0; the frame's outputs stay inside the handle. */
int ko_pipeline_register_frame(ko_pipeline *p, const double *xyz, size_t n,
                               const double *timestamps, size_t n_ts);
void ko_pipeline_pose(const ko_pipeline *p, double T[16]);
void ko_pipeline_delta(const ko_pipeline *p, double T[16]);
const ko_map *ko_pipeline_map(const ko_pipeline *p);
/* last frame's outputs: which = 0 preprocessed_frame, 1 source, 2 frame_downsample */
size_t ko_pipeline_output_size(const ko_pipeline *p, int which);
void ko_pipeline_output(const ko_pipeline *p, int which, double *out);
void ko_pipeline_last_stats(const ko_pipeline *p, ko_icp_stats *s, double *sigma);

int ko_num_procs(void);

#ifdef __cplusplus
}
#endif
#endif
