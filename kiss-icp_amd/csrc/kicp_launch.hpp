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
    unsigned long long *sort_keys;  // non-null: the scatter also leaves each emitted point's sort key (tile_key; kicp_sort.hip)
    double sort_inv_cell;
    // fused stage A of the next VoxelDownsample(out, next_voxel); next_tab == nullptr disables it
    DsSlot *next_tab;
    uint32_t next_mask;
    double next_voxel;
    int *next_slot_of;
    int *err;
    BoundsRec *dbg;  // (bounds-asserting build)
};

// spatial order of the source cloud (kicp_sort.hip): keys = {Morton code of the 2-voxel cell, index}, sorted runs merged by rank
// (small clouds: placed by rank in one launch)
inline double tile_sort_inv_cell(double voxel_size) { return 1.0 / (2.0 * voxel_size); }
#ifdef __HIPCC__
__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ unsigned long long tile_key_of(double x, double y, double z, int i, double inv_cell) {
    // 2-voxel cells, offset so that +-512 cells around the sensor map to 0..1023 (farther points clamp: only the
    // quality of the order is at stake)
    const double cx = floor(x * inv_cell) + 512.0, cy = floor(y * inv_cell) + 512.0, cz = floor(z * inv_cell) + 512.0;
    const unsigned ux = (unsigned)fmin(fmax(cx, 0.0), 1023.0), uy = (unsigned)fmin(fmax(cy, 0.0), 1023.0),
                   uz = (unsigned)fmin(fmax(cz, 0.0), 1023.0);
    const unsigned long long m = (unsigned long long)(spread10(ux) | (spread10(uy) << 1) | (spread10(uz) << 2));
    return (m << 24) | (unsigned long long)(unsigned)i;  // morton30(cell of point i) << 24 | i: unique
}
__device__ __forceinline__ unsigned long long tile_key(const double *xyz, int i, double inv_cell) {
    return tile_key_of(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], i, inv_cell);
}
#endif
// will a cloud of about n_hint points (bound n_max) be ordered by rank?  Then the stage that writes the cloud may leave its
// keys in launch_tile_sort's `keys_in` (DsParams::sort_keys) and say so (keys_ready)
bool tile_sort_by_rank(size_t n_max, size_t n_hint);
size_t tile_sort_temp_bytes(size_t n_max);
int tile_sort_prepare(int device_id);  // LDS opt-in of the block sort, once per device
int launch_tile_sort(const double *xyz, const int *n_ptr, int n_imm, size_t n_max, double voxel_size, unsigned long long *keys_in,
                     unsigned long long *keys_out, size_t n_hint, hipStream_t s, bool keys_ready = false);
int icp_prepare(int device_id);
int icp_blocks_per_cu(int lds_bytes);
void launch_selftest_solve(const double *A, const double *b, int n, double *x, hipStream_t s);  // co-resident k_icp workgroups per CU (occupancy query, current device)
size_t icp_granule_words(int G);
// start / stop: events attached to the dispatch itself (hipExtLaunchKernel: its own completion signal and timestamps --
// no packet of their own in the queue, unlike hipEventRecord); either may be null
void launch_icp(IcpParams P, int G, bool profile, bool wide, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// the run weights of a cloud of short runs, in front of launch_icp on the same stream (P as launch_icp will get it, wts32 set;
// icp_grid: its grid; n_hint: about how many source points there are)
void launch_icp_weights(const IcpParams &P, int icp_grid, size_t n_hint, hipStream_t s);
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
// pr (both launches, or neither): the FUSED update -- RemovePointsFarFromLocation's verdicts are taken beside k_map_link and
// carried out by k_map_apply, which also hands the frame record to the host (kicp_map.hip); no launch_map_prune then
struct MapPrune {
    long bump_ub;          // upper bound of the blocks carved so far (grid of the verdict pass)
    const PipeState *state;
    int use_state_origin;  // origin = state->new_pose.t, else `origin`
    double origin[3];
    unsigned *host_rec;    // the frame's slot of the host-pinned ring, or null
    int rec_words;
};
void launch_map_link(const MapView &m, const InsertScratch &sc, const double *in, const int *n_ptr, int n_imm,
                     int n_max, const PipeState *state, int use_pose, hipStream_t s, const MapPrune *pr = nullptr);
void launch_map_apply(const MapView &m, const InsertScratch &sc, int n_max, hipStream_t s, const MapPrune *pr = nullptr,
                      hipEvent_t done = nullptr);  // done: attached to the dispatch (see launch_icp)
void launch_map_prune(const MapView &m, long bump_ub, const PipeState *state, int use_state_origin,
                      const double origin[3], unsigned *host_rec, int rec_words, hipStream_t s, hipEvent_t done = nullptr);  // done: attached to the dispatch (see launch_icp)
void launch_map_rehash(const MapView &m, long bump_ub, hipStream_t s);
void launch_map_count_points(const MapView &m, long bump_ub, hipStream_t s);

}  // namespace kicp
