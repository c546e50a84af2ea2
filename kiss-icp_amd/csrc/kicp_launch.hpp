// kicp_launch.hpp -- kernel parameter blocks and launch wrappers (kicp_kernels.hip <-> kicp_api.hip)
#pragma once

#include "kicp_internal.hpp"

namespace kicp {

// Preprocessor::Preprocess (core/Preprocessing.cpp:55-95) + fused first-downsample claim
struct PreParams {
    const void *xyz;    // raw scan: xyz triples, float64 or (xyz_f32) float32
    int xyz_f32;
    const double *ts;   // timestamps or nullptr
    int n;              // raw point count (host value: the scan just arrived)
    int deskew;         // deskew_ && !timestamps.empty()
    int use_state_motion;
    SE3 motion;         // relative_motion when !use_state_motion
    const PipeState *state;  // last_delta (pipeline mode)
    PrepState *prep;    // timestamp min/max words (re-armed by k_pre_scatter)
    double max_range, min_range;
    double *tmp;        // n x 3 deskewed cloud
    int *blk_counts;    // one per 1024-thread workgroup
    double *out;        // cropped cloud
    int *n_out;         // device count of `out`
    // fused stage A of VoxelDownsample(out, ds_voxel); ds_tab == nullptr disables it
    DsSlot *ds_tab;
    int ds_order;  // 1: the table is the reference's grid (its hash, its bucket count: DsParams::order)
    uint32_t ds_mask;
    double ds_voxel;
    int *ds_slot_of;
    int *err;
    BoundsRec *dbg;  // (bounds-asserting build)
};

// VoxelDownsample (core/VoxelUtils.cpp:7-21)
struct DsParams {
    // order == 1: the survivors leave in the order the reference emits them -- the iteration (bucket) order of the
    // tsl::robin_map VoxelDownsample collects them in (VoxelUtils.cpp:9-19).  The scratch table then IS that grid:
    // the reference's hash (VoxelUtils.hpp:46-50), its bucket count (reserve(frame.size()) -> the power of two >=
    // 2 n, from the device-side count), linear probing -- which occupies exactly the buckets robin-hood probing
    // occupies; k_ds_arrange then settles, cluster by cluster, WHICH survivor sits in which bucket.
    // order == 0: ascending original index (the table is only a set; own hash, host-side mask).
    int order;
    int tab_cap;   // buckets allocated (>= any bucket count the mode needs); sizes the bucket-wise launches
    int *rb_elem;  // [tab_cap] order == 1: index (into `in`) of the point the reference holds in each bucket
    int *rb_home;  // [tab_cap] scratch of k_ds_arrange
    const double *in;
    const int *n_ptr;  // device count of `in`, or nullptr -> n_imm
    int n_imm;
    int n_max;         // host upper bound of the count (sizes the grid)
    double voxel;
    DsSlot *tab;
    uint32_t mask;
    int *slot_of;
    int *blk_counts;
    double *out;
    int *n_out;
    // fused stage A of the next VoxelDownsample(out, next_voxel); next_tab == nullptr disables it
    DsSlot *next_tab;
    uint32_t next_mask;
    double next_voxel;
    int *next_slot_of;
    int *err;
    BoundsRec *dbg;  // (bounds-asserting build)
};

// spatial order of the source cloud (kicp_sort.hip): keys = {Morton code of the 2-voxel cell, index}, sorted runs merged by rank
size_t tile_sort_temp_bytes(size_t n_max);
int tile_sort_prepare(int device_id);  // LDS opt-in of the block sort, once per device
int launch_tile_sort(const double *xyz, const int *n_ptr, int n_imm, size_t n_max, double voxel_size, unsigned long long *keys_in,
                     unsigned long long *keys_out, size_t n_hint, hipStream_t s);
int icp_prepare(int device_id);
int icp_blocks_per_cu(int lds_bytes);
void launch_selftest_solve(const double *A, const double *b, int n, double *x, hipStream_t s);  // co-resident k_icp workgroups per CU (occupancy query, current device)
size_t icp_granule_words(int G);
// start / stop: events attached to the dispatch itself (hipExtLaunchKernel: its own completion signal and timestamps --
// no packet of their own in the queue, unlike hipEventRecord); either may be null
void launch_icp(IcpParams P, int G, bool profile, bool wide, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
void launch_closest_neighbor(const MapView &m, const double *q, int nq, double *nn, double *dist,
                             hipStream_t s);
void launch_ts_minmax(const double *ts, int n_ts, PrepState *prep, hipStream_t s);
void launch_stage_in(const void *src, void *dst, size_t bytes, hipStream_t s);  // device-mapped host memory -> HBM
void launch_pre_flags(const PreParams &P, hipStream_t s);
void launch_pre_scatter(const PreParams &P, hipStream_t s);
void launch_ds_claim(const DsParams &P, hipStream_t s);
void launch_ds_flags(const DsParams &P, hipStream_t s);
void launch_ds_scatter(const DsParams &P, hipStream_t s);
void launch_ds_arrange(const DsParams &P, hipStream_t s);      // order == 1: instead of launch_ds_flags
void launch_ds_scatter_rb(const DsParams &P, hipStream_t s);   // order == 1: instead of launch_ds_scatter
void launch_map_link(const MapView &m, const InsertScratch &sc, const double *in, const int *n_ptr, int n_imm,
                     int n_max, const PipeState *state, int use_pose, hipStream_t s);
void launch_map_apply(const MapView &m, const InsertScratch &sc, int n_max, hipStream_t s);
void launch_map_prune(const MapView &m, long bump_ub, const PipeState *state, int use_state_origin,
                      const double origin[3], unsigned *host_rec, int rec_words, hipStream_t s, hipEvent_t done = nullptr);  // done: attached to the dispatch (see launch_icp)
void launch_map_rehash(const MapView &m, long bump_ub, hipStream_t s);
void launch_map_count_points(const MapView &m, long bump_ub, hipStream_t s);

}  // namespace kicp
