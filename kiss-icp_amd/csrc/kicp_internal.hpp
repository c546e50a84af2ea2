// kicp_internal.hpp -- device data layout and host-side objects of libkicp (not installed).
//
// HBM layout
//   Voxel map (kiss_icp::VoxelHashMap, core/VoxelHashMap.hpp:38-57)
//     slots[C]   : 16-byte open-addressed slots {u64 packed voxel key, i32 block id, i32 point
//                  count}; C a power of two; linear probing; EMPTY / TOMBSTONE sentinels.  One
//                  probe = one aligned 16-byte load and tells the prober how many points to read.
//     blocks[B]  : fixed-stride voxel blocks, stride = roundup(32 + 24*max_points, 128) bytes
//                  (512 B for the default 20 points = four 128-byte lines):
//                  {u64 key, i32 count, i32 slot, 16 B pad | xy[max_points] as 16-byte pairs |
//                  z[max_points]}.  Lane i of a 32-lane group reads point i of a voxel with one
//                  16-byte and one 8-byte load, both coalesced across the group -- instead of the
//                  reference's std::vector pointer chase.  count == 0 marks a free block.
//     heads[C]   : per-slot list head used only while inserting a frame.
//     free_ids[] : queue (ring) of recycled block ids;  ctr[] : device counters.
//   Frame clouds : row-major xyz f64 (the reference's std::vector<Eigen::Vector3d> layout).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdint>
#include <string>

#include "kicp_math.hpp"
#include "../../include/kicp.h"

namespace kicp {

// ---- device-visible structures -------------------------------------------------------------
struct alignas(16) Slot {
    unsigned long long key;
    int block;  // -1 until a block is attached
    int count;  // points stored in the voxel (mirror of the block header's count)
};

struct alignas(16) DsSlot {  // scratch table of VoxelDownsample
    unsigned long long key;
    int minidx;
    int pad;
};

constexpr int kBlockHeader = 32;  // bytes before the first point of a voxel block
struct BlockHdr {
    unsigned long long key;
    int count;
    int slot;
    int doom;  // fused map update (k_map_link / k_map_apply with a doomed list): 0 = stays, 1 = sentenced by this frame's
               // verdict pass and not yet removed, 2 = removed (a block on the free ring); reset when the block is handed out
    int pad[3];
};
static_assert(sizeof(BlockHdr) == kBlockHeader, "the block header fills the bytes in front of the first point");

// Every counter sits in a 128-byte line of its own: the hot ones are bumped by one lane per workgroup of k_map_apply /
// k_map_prune (returning atomics, ~12 ns each where the line lives), and three of them in ONE line made 625 workgroups
// x 3 atomics queue up behind each other (k_map_apply: 37 us, most of it that queue -- profiles/r03_j_timeline.txt).
constexpr int kCtrStride = 32;  // ints per counter
enum MapCtr {
    C_BUMP = 0 * kCtrStride,   // blocks ever carved from the pool (high-water mark)
    C_FHEAD = 1 * kCtrStride,  // free-block queue: pop cursor (may overshoot C_FTAIL during an insert; k_map_link clamps)
    C_LIVE = 2 * kCtrStride,   // live voxels
    C_TOMB = 3 * kCtrStride,   // tombstoned slots
    C_USED = 4 * kCtrStride,   // slots ever claimed since the last rehash (live + tombstones)
    C_ERR = 5 * kCtrStride,    // sticky error bits (ErrBits)
    C_NPTS = 6 * kCtrStride,   // scratch: point count of the last pointcloud query
    C_TOUCHED0 = 7 * kCtrStride,  // voxel records opened by the running insert (two words used alternately:
    C_TOUCHED1 = 8 * kCtrStride,  //   an insert counts in one and re-arms the other for the next insert)
    C_FTAIL = 9 * kCtrStride,     // free-block queue: end of the entries an insert may pop
    C_FPEND = 10 * kCtrStride,    // free-block queue: push cursor of RemovePointsFarFromLocation (merged into
                                  //   C_FTAIL by the next k_map_link)
    C_DONE = 11 * kCtrStride,     // workgroups of a frame's last kernel that have finished (the last one writes the frame record)
    C_DOOMED0 = 12 * kCtrStride,  // fused map update: voxels sentenced by the verdict pass beside k_map_link = entries of
    C_DOOMED1 = 13 * kCtrStride,  //   MapView::doomed (two words used alternately, like C_TOUCHED0 / 1)
    C_COUNT = 14 * kCtrStride
};

constexpr int kDoneSub = 16;  // first-level sign-off words of a frame's last kernel (kicp_map.hip: frame_record_handoff)

enum ErrBits { E_RANGE = 1, E_TABLE_FULL = 2, E_POOL_FULL = 4, E_TIMEOUT = 8, E_BOUNDS = 16 };

// ---- the bounds-asserting build (make DEBUG_BOUNDS=1 -> -DKICP_DEBUG_BOUNDS) --------------------------------------
// Every index a kernel takes FROM DEVICE MEMORY and then dereferences -- sorted keys, point indices, block ids, slot
// numbers, store positions, list offsets -- goes through KICP_IDX(dbg, err, idx, limit, tag).  The debug build checks it:
// the first violation of a launch is recorded in the device's BoundsRec {source line, tag, index, limit, workgroup,
// thread}, E_BOUNDS is raised in the error word the kernel reports through, and the index is replaced by 0 so that the
// launch ends instead of faulting; the host turns the bit into KICP_ERR_HIP with the record spelled out.  The release
// build compiles the macro to the bare index: no instruction, no register.
struct BoundsRec {
    int taken;  // 0 until the first violation claims the record (atomicCAS)
    int line;
    long long idx, limit;
    int block, thread;
    int tag, pad;
};
#ifdef KICP_DEBUG_BOUNDS
__device__ __forceinline__ long long kicp_bounds_fail(BoundsRec *dbg, int *err, long long idx, long long limit, int line, int tag) {
    if (err) atomicOr(err, E_BOUNDS);
    if (dbg && atomicCAS(&dbg->taken, 0, 1) == 0) {
        dbg->line = line;
        dbg->idx = idx;
        dbg->limit = limit;
        dbg->block = (int)blockIdx.x;
        dbg->thread = (int)threadIdx.x;
        dbg->tag = tag;
    }
    return 0;
}
template <class T>
__device__ __forceinline__ T kicp_idx(BoundsRec *dbg, int *err, T idx, long long limit, int line, int tag) {
    if ((unsigned long long)(long long)idx < (unsigned long long)limit) return idx;
    return (T)kicp_bounds_fail(dbg, err, (long long)idx, limit, line, tag);
}
#define KICP_IDX(dbg, err, idx, limit, tag) ::kicp::kicp_idx((dbg), (err), (idx), (long long)(limit), __LINE__, (tag))
constexpr bool kDebugBounds = true;
#else
#define KICP_IDX(dbg, err, idx, limit, tag) (idx)
constexpr bool kDebugBounds = false;
#endif

struct MapView {
    Slot *slots;
    uint32_t mask;
    char *blocks;
    int stride;
    int max_points;
    int blocks_cap;
    int *ctr;
    int *free_ids;  // ring of recycled block ids, indexed modulo free_cap by the C_F* cursors
    int free_cap;
    int *done_sub;  // kDoneSub sign-off words, a 128-byte line each (frame_record_handoff: the first level of a two-level count)
    int *doomed;  // fused map update: block ids of the voxels this frame removes (C_DOOMED0 / 1 entries; room for every block)
    int *heads;  // per-slot id of the voxel record opened by the running insert (-1 when idle)
    int z_off;   // byte offset of the z array inside a block = 32 + 16 * max_points
    double voxel_size;
    double max_distance;
    double map_resolution;  // sqrt(voxel_size^2 / max_points)  VoxelHashMap.cpp:98
    BoundsRec *dbg;         // the device's record of the bounds-asserting build (never touched by the release build)
};

// (every block id comes from device memory -- a slot, a table entry, the free ring: the bounds-asserting build checks them all here)
__device__ __forceinline__ BlockHdr *block_hdr(const MapView &m, int b) {
    b = KICP_IDX(m.dbg, m.ctr + C_ERR, b, m.blocks_cap, 100);
    return reinterpret_cast<BlockHdr *>(m.blocks + (size_t)b * m.stride);
}
__device__ __forceinline__ double2 *block_xy(const MapView &m, int b) {
    b = KICP_IDX(m.dbg, m.ctr + C_ERR, b, m.blocks_cap, 101);
    return reinterpret_cast<double2 *>(m.blocks + (size_t)b * m.stride + kBlockHeader);
}
__device__ __forceinline__ double *block_z(const MapView &m, int b) {
    b = KICP_IDX(m.dbg, m.ctr + C_ERR, b, m.blocks_cap, 102);
    return reinterpret_cast<double *>(m.blocks + (size_t)b * m.stride + m.z_off);
}

// Scratch of one AddPoints call (sized by the number of incoming points).  k_map_link opens one
// "voxel record" per touched voxel and files every incoming point under its voxel's record;
// k_map_apply then serves one record per 32-lane group.
constexpr int kRecList = 32;  // point indices kept per record (more -> serial fallback via the chain)
struct InsertScratch {
    double *world;  // incoming points in the map frame
    int *next;      // per point: next point of the same voxel (chain, newest first; fallback only)
    int *rec_slot;  // per record: hash slot of its voxel (-1: record lost its race, empty)
    int *rec_block; // per record: the voxel's block when the record was opened (-1: a voxel this insert creates)
    int *rec_count; // per record: incoming points filed (0 when idle)
    int *rec_head;  // per record: chain head (-1 when idle)
    int *rec_list;  // per record: the first kRecList point indices, unordered
    int parity;     // which C_TOUCHED word this insert counts in
};

// Per-frame words written by the stages in FRONT of the registration (Preprocess + the two
// VoxelDownsamples).  The pipeline keeps two copies and alternates between them, so that those
// stages of frame k+1 can run on a second stream while frame k's map update is still in flight.
struct PrepState {
    unsigned long long tmin_bits, tmax_bits;  // timestamp min/max as order-preserving u64 (idle: ~0, 0)
    int n_pre, n_fd, n_src;                   // points after range crop / 0.5 v / 1.5 v downsample
    int pad;
};

constexpr int kIcpProfIters = 24;

// Per-pipeline state that never leaves the device between frames
// (pipeline/KissICP.hpp:87-95 + core/Threshold.hpp:44-50).
struct PipeState {
    SE3 last_pose;
    SE3 last_delta;
    SE3 new_pose;  // result of the frame being processed
    SE3 guess;     // initial guess used by the frame being processed
    double model_sse;
    double sigma;  // threshold used by the frame being processed
    int num_samples;
    unsigned epoch_base;  // tag base of the in-kernel exchange, advanced by every ICP launch
    // per-frame result block (copied D2H at sync)
    int n_raw, n_pre, n_fd, n_src;
    int icp_iterations, icp_converged;
    unsigned long long icp_examined, icp_ncorr_last, icp_ncorr_total;
    int err;
    int icp_blocks_used;  // workgroups that took part in the last ICP launch
    // the 16 unique scalars of J^T W J / J^T W r (BuildLinearSystem, Registration.cpp:80-121), the
    // correspondence count and the examined count of the LAST iteration of the last launch (k_icp's sum order)
    double icp_last_sums[18];
    // shader-clock cycles spent by workgroup 0 in the phases of the last ICP launch:
    // [0] association+accumulate, [1] workgroup reduce+publish, [2] gather, [3] solve+update
    unsigned long long prof[4];
    // whole-launch duration of the last ICP kernel seen by workgroup 0: shader-clock cycles (s_memtime)
    // and 100 MHz wall ticks (s_memrealtime) -> effective shader clock
    unsigned long long prof_clock[2];
    unsigned long long prof_it0_ticks;  // 100 MHz ticks from the launch's start to the end of its first iteration
    // the first kIcpProfIters iterations of the last launch, 10 ns ticks: workgroup 0's
    // {associate, publish, gather, solve}, the slowest group's associate time over ALL workgroups,
    // and the number of polling passes workgroup 0's thread 0 needed in the gather
    unsigned prof_iter[kIcpProfIters][6];
};

constexpr int kIcpSums = 19;  // 16 normal-equation scalars + correspondence count + examined count
                              // + one profiling slot (association ticks, max-reduced)
constexpr int kIcpTickSlot = 18;
// 16-byte slots from one workgroup's block of granule pairs to the next: 24 -- blocks of 384 bytes, three 128-byte lines that no
// other workgroup writes.  Packed (19 slots, 304 bytes) most lines were written by TWO workgroups, as a rule on different XCDs:
// the exchange alone, 224 workgroups, scripts/probes/xchg_bench.hip: 3.23 us per round packed, 2.97 at 24 slots (32: 3.24 --
// four lines, the same channels again); profiles/r06_ac_*.
constexpr int kIcpGranStride = 24;
static_assert(kIcpGranStride >= kIcpSums && (kIcpGranStride * 16) % 128 == 0, "whole lines per workgroup");
constexpr int kIcpThreads = 512;
constexpr int kIcpGroup = 32;  // lanes cooperating on one source point (27 probe lanes)
constexpr int kIcpGroupsPerBlock = kIcpThreads / kIcpGroup;  // 16
constexpr int kIcpSolveThreads = 256;  // waves 0..3 (one per SIMD) solve the 6x6 system, the rest wait
constexpr int kIcpBookThread = kIcpThreads - 64;  // first lane of the last wave: pose / statistics bookkeeping
constexpr int kIcpExchangeGroups = 16;  // leaders of the two-level exchange (workgroups 0..15).  Measured on one box, bench
                                        // scene (profiles/r04_ak_exchange_leaders_ab.txt): 4 leaders 2602, 8: 2785, 16: 2856,
                                        // 32: 2715, 64: 2459 scans/s -- the first hop gets shorter, the second longer
constexpr int kIcpGroupCopies = 8;      // copies of every leader's group sums (kicp_icp.hip: a workgroup polls copy b mod 8)
constexpr bool kIcpPublishDpp = true;   // a workgroup's sums over its 16 groups by DPP row operations, stored by the lane that holds them (kicp_icp.hip)
constexpr bool kIcpPollAll = true;      // the second hop without its watcher lanes (kicp_icp.hip; false: round 3's form, for A/Bs)
constexpr bool kIcpSpreadSearches = true;  // group form, phase B: the first searches of a workgroup on different waves (kicp_icp.hip)
constexpr int kIcpMaxMembers = 16;      // workgroups per leader at most (256 / 16)
constexpr int kIcpSumRows = kIcpMaxMembers > kIcpExchangeGroups ? kIcpMaxMembers : kIcpExchangeGroups;
constexpr int kIcpMaxBlocks = 256;
constexpr int kIcpLdsBytesMax = 160 * 1024;  // one workgroup per CU owns the whole LDS
constexpr int kIcpChunk = 128;     // local points a workgroup carries through the phases of an iteration at a time
constexpr int kIcpTermChunk = 64;  // points whose products are in LDS together (a multiple of the group count)
constexpr int kIcpTerms = 18;      // 16 normal-equation scalars + correspondence count + examined count
constexpr int kIcpMaxMeta = 448;   // local points of a workgroup that may use the workgroup's voxel tile (112 bytes each)
constexpr int kIcpTileSlots = 4096;  // slots of the workgroup's voxel table (occupied voxels only; power of two)
constexpr int kIcpListRunMax = 64;   // workgroups that serve at most this many points keep a scan list per point
constexpr int kIcpWeightedMin = 2048;  // source clouds of at least this many points are cut into runs of equal weight
constexpr int kIcpListPool = 12288;  // 16-bit entries of the scan-list pool (24 KiB)
constexpr size_t kIcpGroupProfileWords = (size_t)kIcpProfIters * kIcpMaxBlocks * kIcpGroupsPerBlock * 4;

// LDS record of one source point ("query") of a workgroup of the persistent ICP kernel
struct IcpQueryMeta {
    double s[3];  // running transformed source point
    int v[3];     // voxel the known window is centred on
    signed char lo[3], hi[3];  // extent of the known window per axis, in voxels relative to v (-2..-1, 1..2)
    signed char valid;       // 1: every occupied voxel of the window is in the tile; 0: not looked yet; -1: cannot use the tile
    signed char list_state;  // 1: the scan list is current for voxel lv; 0: none yet; -1: cannot have one
    int lv[3];               // voxel of the query the scan list was built for
    int list_base;           // first entry of the list in the pool
    unsigned short list_n, list_cap;
    // STABILITY (option icp_group_stable; the thread-per-query form's WideQuery::Lr, kicp_icp_wide.hpp).  The last full search,
    // made from position ss in voxel lv, left L2: the SECOND smallest squared distance over all candidates of the 27 cells
    // (shaved by 2^-20), i.e. every candidate but the neighbour found was at least sqrt(L2) from ss.  The query is now at s:
    // those candidates are at least sqrt(L2) - |s - ss| away (triangle inequality), and while the neighbour's new distance is
    // strictly below that -- tested on the squares, no root: with a = |nn - s|^2, b = |s - ss|^2, R = L2 - a - b the
    // condition sqrt(a) + sqrt(b) < sqrt(L2) is R > 0 and 4 a b < R^2 -- and the query is still in voxel lv (same 27 cells, same
    // candidates), the neighbour is still what the reference's strict '<' loops would find: no search.
    double L2;
    double ss[3];
    int lr_state;  // 1: L2 / ss, and the neighbour / count kept in the workgroup's point slot, belong to voxel lv
    int pad[3];
};
static_assert(sizeof(IcpQueryMeta) == 112 && sizeof(IcpQueryMeta) % 16 == 0, "IcpQueryMeta layout");

struct IcpParams {
    const double *frame;  // N x 3 source points in the sensor frame
    const unsigned long long *order;  // sorted tile keys (low 24 bits: index into frame) or nullptr (identity)
    unsigned long long *wts;  // one tagged granule per sorted position: the point's run weight, written by k_icp's prologue
                              // (runs of equal weight), or nullptr (runs of equal length)
    int weight_base, weight_quad;  // a point weighs base + c (+ c^2 / quad; quad < 0: when runs are short), c = population of its voxel,
    int weight_long_base;                    //   (clouds of more than 64 points per workgroup: long_base + c + long_emul * E)
    int weight_long_emul;
    int weight_dense_min, weight_dense_div;  //   + max(0, E - dense_min) / dense_div, E = population of its 27 voxels (dense_div 0: off)
    double *work;         // N x 3 transformed source, private to the launch
    const int *n_ptr;     // device count (pipeline) or nullptr
    int n_imm;            // count when n_ptr == nullptr
    MapView map;
    PipeState *state;       // guess / sigma / result live here
    int pipeline_mode;      // 1: compute guess+sigma from state and do the frame bookkeeping
    double max_dist, kernel_scale;  // used when pipeline_mode == 0
    double min_motion_th;           // AdaptiveThreshold::min_motion_threshold_ (pipeline mode)
    int max_iters;
    double conv;
    unsigned long long *granules;  // [2][gridDim.x][kIcpGranStride * 2] tagged 8-byte words (kIcpSums pairs used of each block)
    unsigned spin_limit;
    int points_per_group;  // target points per 32-lane group and iteration (sets how many
                           // of the launched workgroups take part: ceil(n / (8 * this)))
    int force_blocks;      // > 0: exactly this many workgroups take part
    int use_lds;           // stage candidate voxels in LDS (0 disables)
    int bulk_fill;         // first iteration: establish all windows of a chunk workgroup-wide (tile_fill_bulk) instead of query by query
    int lds_bytes;         // dynamic LDS of the launch (kIcpLdsBytesShared or kIcpLdsBytesMax)
    int schur_solve;       // solve well-conditioned normal equations through their 3 x 3 Schur complement (kicp_math.hpp: schur3_solve)
    int use_wide;          // host side only: launch the thread-per-query form (k_icp<.., true>)
    unsigned *wts32;       // non-null: k_icp_weights has left the run weights here as plain 32-bit words (short runs only: at most
                           // kIcpListRunMax * kIcpMaxBlocks of them), in front of this launch
    int group_stable;      // group form: queries whose neighbour cannot have changed skip the search (IcpQueryMeta::Lr)
    int wide_stable;       // thread-per-query form: queries whose neighbour cannot have changed skip the search (WideQuery::Lr), the rest
                           // are searched on the first lanes (0: every query is searched in place, every iteration)
    int wide_promote_from; // thread-per-query form: first iteration whose map reads leave their voxels in the LDS store
    int wide_load_eighths; // thread-per-query form: the tile's table is at most this many eighths full
    int wide_per_round;    // thread-per-query form: items a thread files per round and queue
    int wide_group_max;    // thread-per-query form: at most this many full searches of a workgroup go to the 32-lane groups, one query each (0: never)
    int wide_flat;         // thread-per-query form: bit 0 / bit 1: the map / the LDS queue is served by a thread per point (wide_serve_flat)
    int wide_prefill;      // thread-per-query form: eighths of the LDS store the window phase fills with points (0: the table only; the
                           // store is then filled by the searches themselves, from the second iteration on)
    int wide_prune;        // thread-per-query form: 0 visit every occupied voxel, 1 skip voxels by their box distance, 2 also bounded
                           // by the previous iteration's neighbour (kicp_icp_wide.hpp; the result is the same)
    int inject_timeout;    // test hook: behave like a launch whose workgroups never became co-resident
    const PrepState *prep;  // pipeline mode: this frame's counts (copied into the frame record), or nullptr
    unsigned *prof_groups;  // profiling variant only: [kIcpProfIters][256 * 16][4] per-group records, or nullptr
};

// ---- host-side objects ------------------------------------------------------------------------
void set_error(const char *fmt, ...);
const char *get_error();

#define KICP_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ::kicp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                    \
            return (_e == hipErrorOutOfMemory) ? KICP_ERR_OOM : KICP_ERR_HIP;               \
        }                                                                                   \
    } while (0)

#define KICP_TRY(expr)                  \
    do {                                \
        int _s = (expr);                \
        if (_s != KICP_OK) return _s;   \
    } while (0)

struct Options {
    long icp_blocks = 0;
    long icp_points_per_group = 1;
    long icp_use_lds = 1;        // stage candidate voxels in LDS and reuse them across iterations
    long icp_bulk_fill = 1;      // first iteration: all windows of a workgroup established in two bulk waves of loads
    long icp_schur_solve = 1;    // well-conditioned normal equations are solved through their 3 x 3 Schur complement (0: always the 6 x 6 pivoted LDLT)
    long icp_wide = -1;          // association form: 1 a thread per source point (kicp_icp_wide.hpp), 0 a 32-lane group, -1 by the cloud's size
    long icp_weights_kernel = 1; // the run weights of short runs are computed by a kernel in front of k_icp (0: inside the launch, by its prologue)
    long sort_by_rank = 1;       // source clouds of up to ~6.5 k points (by the previous frame's count) are ordered by rank in one launch (kicp_sort.hip); 0: always runs + merge passes
    long map_fused_update = 1;   // RemovePointsFarFromLocation inside the two kernels of AddPoints (kicp_map.hip); 0: a third kernel behind them
    long icp_group_stable = 1;   // group form: skip the search of queries whose neighbour provably stays (0: search every query, every iteration)
    long icp_wide_stable = 1;    // thread-per-query form: skip the search of queries whose neighbour provably stays (0: search every query, every iteration)
    long icp_wide_promote_from = 1;
    long icp_wide_load_eighths = 5;
    long icp_wide_per_round = 4; // thread-per-query form: items a thread files per round and queue (1 .. 27)
    long frame_events = 0;       // pipelines created afterwards: 1 = the per-frame events ride on the dispatches of k_icp and k_map_prune (hipExtLaunchKernel)
                                 // instead of being recorded between them -- measured: 26 us per frame SLOWER (profiles/r04_av_frame_events_sweep.txt)
    long icp_wide_group_max = 128; // thread-per-query form: up to this many full searches per workgroup are run by the 32-lane groups (wide_group_scan)
    long icp_wide_flat = 3;      // thread-per-query form: serve the voxel queues by a thread per point (bit 0: the map's, bit 1: the LDS store's)
    long icp_wide_prefill = 0;   // thread-per-query form: eighths of the LDS store filled by the window phase (0 .. 8)
    long icp_wide_prune = 2;     // thread-per-query form: 0 visit every occupied voxel, 1 skip by box distance, 2 + bound from the last neighbour
    long icp_profile = 0;        // 1: launch the ICP kernel variant that records phase timers
    long icp_timing = 1;
    long map_apply_threads = 512;  // workgroup size of k_map_apply (256 / 512 / 1024)
    long icp_lds_kib = 0;        // dynamic LDS per ICP workgroup in KiB (0: all 160)
    long icp_device_streams = 1; // pipelines created from now on share their GPU with this many streams in all: each registers with 1 / n
                                 // of the persistent grid, and up to n registrations run side by side (kicp_api.hip, the gate)
    long icp_reserve_cus = 32;   // CUs left out of the ICP grid (one per shader engine) for the front stages of the next frame
    long staging_threads = 3;    // helper threads (besides the caller) for host-side staging copies
    long staging_f32 = 1;        // narrow float64 scans to float32 for the upload when that is lossless
    long staging_zero_copy = 1;  // the front kernels read the scan straight from the pinned staging slot (no upload call)
    long stage_in = 1;           // ... unless the frame deskews: then a copy kernel brings the scan into HBM under the previous registration
    long staging_numa_pretend = -1;  // test hook: >= 0 = the node every GPU is declared to hang off (so that "staging_numa" = 2 has something to move on a one-GPU box)
    long staging_numa = 1;       // 1: where the staging slots lie is checked and reported (kicp_numa.hpp); 2: slots on another node than the GPU's are re-made on it, helper threads and batch workers run on its CPUs; 0: nothing
    long relaxed_backpressure = 1;  // a caller that is queue_depth frames ahead of the device sleeps between polls instead of yielding in a loop
    long collective_timeout_ms = 1800000;  // kicp_batch_*: a step that waits for PEERS (communicator rendezvous, pose all-gather through a host communicator) is given up after this long; 0 = never
    long icp_weight_base = 128;  // run boundaries: a source point weighs this + the population of its voxel
    long icp_weight_long_base = 128;  // the base for clouds of more than 64 points per workgroup (weight = this + c + emul * E)
    long icp_weight_long_emul = 1;    // ... and the multiplier of E there
    long icp_weight_dense_min = 200;  // ... + (population of its 27 voxels - this) / icp_weight_dense_div when positive
    long icp_weight_dense_div = 1;    //     (0: off)
    long icp_weight_quad = -1;   // the weight also carries population^2 / this; 0: never; -1: when runs are short (kicp_sort.hip)
    long icp_inject_timeout = 0; // test hook: the first N registrations of a new pipeline give up at once
    long icp_inject_timeout_skip = 0;  // ... after this many registrations that are left alone
    long map_rehash_every = 0;   // test hook: a pipeline rebuilds its map's slot array in stream order every N frames
    long downsample_order = 1;   // VoxelDownsample output order: 1 the reference's (tsl::robin_map bucket order), 0 ascending index
    long queue_depth = 4;        // frames a pipeline keeps queued on the device before an asynchronous entry waits (>= 2; 0: no limit)
    long wait_timeout_ms = 120000;  // deadline of every host-side wait for the device (kicp_wait.hpp); 0: none
    long inject_stall_ms = 0;    // test hook: the next piece of work a handle queues is preceded by a kernel that spins this long
};
Options &options();

// ---- every host-side wait for the device has a deadline (kicp_api.hip) ---------------------------------------------
// hipStreamSynchronize / hipEventSynchronize block for as long as the device takes -- for ever, if a kernel never ends.
// The library never calls them: a stream or an event is POLLED (hipStreamQuery / hipEventQuery) against a wall-clock
// limit (option "wait_timeout_ms"), and a wait that does not end returns KICP_ERR_TIMEOUT naming what was waited for.
// The calls that synchronise implicitly (hipFree, hipHostFree, hipStreamDestroy, the blocking hipMemcpy) are only made
// behind such a wait, over every stream the library has created on the device; when that wait gives up the resource is
// leaked, not freed -- a leak can be lived with, a process that never returns cannot.
int wait_stream(hipStream_t s, const char *what);
int wait_stream_peers(hipStream_t s, const char *what);  // the same under collective_timeout_ms: work that waits for other ranks
int wait_event(hipEvent_t e, const char *what, bool relaxed = false);
int wait_device(int device_id, const char *what, bool null_stream = false);  // what is queued NOW on every stream libkicp has on the device (a snapshot, like hipFree's own wait)
hipStream_t util_stream(int device_id);            // the library's own stream for buffer initialisation (nullptr: creation failed)
void stream_register(int device_id, hipStream_t s);
void stream_forget(hipStream_t s);
int stream_destroy(hipStream_t s);  // waits (bounded), forgets, destroys; KICP_ERR_TIMEOUT: the stream is leaked
void inject_stall(hipStream_t s);   // test hook (option "inject_stall_ms")

// growable device buffer; a new buffer starts as zeros (the tagged-granule exchanges rely on it: no launch uses tag 0)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need, bool keep = false, hipStream_t s = nullptr);  // s: the stream the old contents are final on (keep); nullptr: the device's utility stream
    void release();             // behind a bounded wait for the device (hipFree synchronises); a device that does not answer keeps the memory
    void drop(bool device_idle);  // the caller has made that wait for several buffers at once: free (idle) or leak (not)
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

int check_device(int device_id);
BoundsRec *bounds_rec(int device_id);  // the device's BoundsRec (allocated on first use, zeroed), or nullptr

}  // namespace kicp

struct kicp_pipeline;
namespace kicp {
// kicp_pipeline_create with the pipeline's share of its device's persistent grid given explicitly (0: option "icp_device_streams")
int pipeline_create_shared(const kicp_config *cfg, int device_id, int share, kicp_pipeline **out);
int device_numa_node(int device);  // the NUMA node a device hangs off (sysfs); -1: unknown
}  // namespace kicp

// The opaque handles of the C-ABI
struct kicp_map {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    double voxel_size = 1.0, max_distance = 100.0;
    unsigned max_points = 20;
    int stride = 512;
    kicp::DevBuf slots, heads, blocks, free_ids, doomed, ctr;
    uint32_t slot_cap = 0;
    int blocks_cap = 0;
    // host-side upper bounds of the device counters (exact after refresh_counters)
    long used_ub = 0, bump_ub = 0, live_ub = 0;
    int h_ctr[kicp::C_COUNT] = {0};
    // host-side event counters (kicp_pipeline_host_stats)
    uint64_t n_refresh = 0, n_grow = 0, n_rehash = 0;
    double wait_ms = 0.0;  // host time blocked in refreshes / growth
    // scratch of add_points
    kicp::DevBuf pts_in, world, next, rec_slot, rec_block, rec_count, rec_head, rec_list;
    unsigned insert_seq = 0;
    int scratch_reserve(size_t n_max, kicp::InsertScratch &sc);  // sizes the buffers, picks the parity
    kicp::MapView view() const;
    bool capacity_ok(size_t incoming_points) const;  // by the host-side upper bounds alone
    int ensure_capacity(size_t incoming_points);
    int refresh_counters();  // D2H of ctr (synchronises the stream)
    int check_errors();
};

struct kicp_registration {
    int device = 0;
    hipStream_t stream = nullptr;
    int max_iters = 500;
    double conv = 1e-4;
    kicp::DevBuf frame, work, granules, state, sort_in, sort_out, sort_tmp, run_wts, run_wts32;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_sums[18] = {0};  // of the most recent kicp_align_points_to_map
};
