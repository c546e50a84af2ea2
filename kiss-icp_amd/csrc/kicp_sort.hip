// kicp_sort.hip -- spatial order of the source cloud for the persistent ICP kernel.
//
// k_icp gives every workgroup a CONTIGUOUS run of the source cloud and keeps the map voxels that run
// needs in LDS (its "tile").  For the tile to be small the run must be compact in space, so the cloud is
// ordered by {Morton code of the point's 2-voxel cell in the sensor frame, original index}: the index
// makes the keys unique, hence the order -- and with it every floating-point sum downstream --
// deterministic.  Rigid motion preserves neighbourhoods, so the sensor frame is as good as the map frame
// (and is available before the pose is).
//
// The sort sorts what EXISTS: the cloud has a few thousand points (N_src, known on the device only), the scan it came
// from 130 k.  A library radix sort sized by the host-side bound spent ~70 us in nine launches ordering 126 k padding
// keys (round 2).  Here: k_tile_sort_blocks computes the keys and sorts runs of 2048 of them in LDS (bitonic network,
// in place, one workgroup per run; workgroups beyond N_src leave at once), and k_tile_merge_runs merges the runs by
// rank: position = rank in the own run + number of smaller keys in every other run of its group (binary searches; the keys
// are unique) -- all runs in one step when they are few, eight at a time first when they are many.  The result is THE
// ascending order of the keys, whatever produced it.
#include <cstring>
#include <mutex>

#include <hip/hip_runtime.h>

#include "kicp_search.hpp"

namespace kicp {

constexpr int kSortRun = 2048;     // keys one workgroup sorts in LDS (a single CU moves 128 B of LDS per clock: a bitonic
                                   // network over 8192 keys is ~40 us of LDS traffic alone, over 2048 keys ~6)
constexpr int kSortThreads = 1024;

// run r = keys of the points [r * kSortRun, min(n, (r + 1) * kSortRun)), sorted, written to `runs` at the same positions
// (a cloud of one run goes straight to `final`)
__global__ __launch_bounds__(kSortThreads) void k_tile_sort_blocks(const double *xyz, const int *n_ptr, int n_imm, double inv_cell,
                                                                   unsigned long long *runs, unsigned long long *final) {
    __shared__ unsigned long long skeys[kSortRun];
    const int n = n_ptr ? *n_ptr : n_imm;
    unsigned long long *out = n <= kSortRun ? final : runs;
    const int first = (int)blockIdx.x * kSortRun;
    if (first >= n) return;
    const int cnt = min(kSortRun, n - first);
    int m = 64;  // padded to a power of two with keys that sort to the end
    while (m < cnt) m <<= 1;
    for (int i = threadIdx.x; i < m; i += kSortThreads) skeys[i] = i < cnt ? tile_key(xyz, first + i, inv_cell) : ~0ull;
    __syncthreads();
    // Bitonic network, pair index i -> elements {lo, lo + j}.  A wave's 64 consecutive pair indices touch one aligned
    // segment of 128 elements whenever j <= 64, and the same segment in the next such stage: those stages need ordering
    // inside the wave only, not a workgroup barrier.
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (m >> 1); i += kSortThreads) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo + j;
                const unsigned long long a = skeys[lo], b = skeys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    skeys[lo] = b;
                    skeys[hi] = a;
                }
            }
            if (j > 64 || (j == 1 && (k << 1) > 128)) __syncthreads();  // the next stage leaves the segment (or was the last)
            else group_lds_sync();
        }
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kSortThreads) out[first + i] = skeys[i];
}

// A merge PASS by rank: the keys come as sorted runs of run_len (the last one shorter); consecutive runs are merged in
// groups of `fan` (0: all of them in one group): a key's position = the group's start + its rank in its own run + the
// number of smaller keys in every other run of the group (a binary search each; the keys are unique).  A group of one
// run is copied.  With fan = 0 and run_len = kSortRun this is "all runs in one step" -- right for the few runs of a
// full-size-voxel scan (3 runs: two searches per key); its cost grows as n x runs x log, so a cloud of many runs
// (the 1M-point configuration: 46; the entry admits 8192) is first merged eight runs at a time: launch_tile_sort.
__global__ __launch_bounds__(256) void k_tile_merge_runs(const unsigned long long *runs, unsigned long long *out, const int *n_ptr, int n_imm,
                                                         int run_len, int fan) {
    const int n = n_ptr ? *n_ptr : n_imm;
    if (n <= kSortRun) return;  // a single run: k_tile_sort_blocks has already written it to the final buffer
    const int n_runs = (n + run_len - 1) / run_len;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = runs[i];
        const int own = i / run_len;
        const int g0 = fan > 0 ? (own / fan) * fan : 0, g1 = fan > 0 ? min(g0 + fan, n_runs) : n_runs;
        int pos = g0 * run_len + (i - own * run_len);
        // (the searches one after the other.  Eight runs in lock step -- one load per run in flight instead of a chain of
        // 7 x 11 -- measured SLOWER: 156 + 55 us against 120 + 18 for the two passes of the 1M-point configuration, 8 - 12
        // against 6 - 7 us on the bench scene, profiles/r04_ax_*: the merge runs beside k_map_prune, which keeps the memory
        // system saturated, and more requests in flight buy nothing there)
        for (int r = g0; r < g1; ++r) {
            if (r == own) continue;
            const int r0 = r * run_len;
            int lo = 0, hi = min(run_len, n - r0);
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (runs[r0 + mid] < key) lo = mid + 1;
                else hi = mid;
            }
            pos += lo;
        }
        out[pos] = key;
    }
}

// ------------------------------------------------------------------------------------------
// A cloud of a few thousand points -- the source cloud of a full-size-voxel scan: 3 - 5 k -- sorted BY RANK in one launch:
// a key's place is the number of smaller keys (the keys are unique), so every workgroup settles kRankKeys keys against
// all n without talking to anybody: no network, no merge, no second launch.  n^2 comparisons -- 22 M for 4.7 k keys --
// spread over the whole device are 2 - 3 us; the bitonic network above is 66 dependent stages in ONE workgroup's LDS per
// run (21 us for 2048 keys, however few runs there are) plus the merge pass (5 us).  With deskewing the sort is on the
// serial chain of every frame (the front stages wait for the previous pose, KissICP.cpp:38-47).
// Every workgroup reads all keys (tiles of kRankTile in LDS) -- written beside the cloud by the stage that produced it
// (DsParams::sort_keys: the second VoxelDownsample's scatter), else computed here, by every workgroup for itself: 10 us
// instead of 5 on the MulRan-like scene (profiles/r06_i_timeline_mulran.txt) --; 16 threads
// share one key's comparisons, lane s taking the candidates s, s + 16, ... (a wave reads 16 consecutive LDS words per step,
// each broadcast to its four keys).  Correct for ANY n (tiles and key blocks are loops); chosen by the hint only.
// ------------------------------------------------------------------------------------------
constexpr int kRankThreads = 256;
constexpr int kRankSplit = 16;                         // threads per key
constexpr int kRankKeys = kRankThreads / kRankSplit;   // keys per workgroup and trip
constexpr int kRankTile = 4096;                        // candidate keys in LDS at a time (32 KB)
constexpr size_t kRankMaxHint = 8192;                  // beyond: the bitonic runs + merge passes
__global__ __launch_bounds__(kRankThreads) void k_tile_rank_sort(const double *xyz, const int *n_ptr, int n_imm, double inv_cell,
                                                                 const unsigned long long *keys, unsigned long long *out) {
    __shared__ unsigned long long skeys[kRankTile];
    const int n = n_ptr ? *n_ptr : n_imm;
    const int sub = (int)threadIdx.x % kRankSplit, slot = (int)threadIdx.x / kRankSplit;
    for (int base = (int)blockIdx.x * kRankKeys; base < n; base += (int)gridDim.x * kRankKeys) {  // workgroup-uniform
        const int i = base + slot;
        const unsigned long long key = i < n ? (keys ? keys[i] : tile_key(xyz, i, inv_cell)) : ~0ull;
        int rank = 0;
        for (int t0 = 0; t0 < n; t0 += kRankTile) {
            const int cnt = min(kRankTile, n - t0);
            __syncthreads();  // (the tile before has been read)
            if (keys) {  // (left by the stage that wrote the cloud: DsParams::sort_keys)
                for (int j = threadIdx.x; j < cnt; j += kRankThreads) skeys[j] = keys[t0 + j];
            } else {
                for (int j = threadIdx.x; j < cnt; j += kRankThreads) skeys[j] = tile_key(xyz, t0 + j, inv_cell);
            }
            __syncthreads();
            int j = sub;
            for (; j + 3 * kRankSplit < cnt; j += 4 * kRankSplit) {  // four reads in flight
                const unsigned long long a = skeys[j], b = skeys[j + kRankSplit], c = skeys[j + 2 * kRankSplit], d = skeys[j + 3 * kRankSplit];
                rank += (a < key) + (b < key) + (c < key) + (d < key);
            }
            for (; j < cnt; j += kRankSplit) rank += skeys[j] < key;
        }
#pragma unroll
        for (int off = 1; off < kRankSplit; off <<= 1) rank += __shfl_xor(rank, off, 64);
        if (sub == 0 && i < n) out[rank] = key;
    }
}

size_t tile_sort_temp_bytes(size_t) { return 256; }  // (the sort needs no scratch beyond its two key buffers)

int tile_sort_prepare(int) { return 0; }  // (the block sort's 16 KiB of LDS need no opt-in)

// sorted keys of the cloud -> keys_out; keys_in is the other buffer of the merge passes (both hold n_max keys).
// n_hint: about how many points there will be (the previous frame's count; 0: unknown -> n_max): it only decides HOW the
// runs are merged -- passes of fan-in 8 first when there are many, and a last pass that merges whatever is left all
// against all, so a wrong hint costs time, never the order.
bool tile_sort_by_rank(size_t n_max, size_t n_hint) {
    if (n_hint == 0 || n_hint > n_max) n_hint = n_max;
    return options().sort_by_rank != 0 && n_hint + n_hint / 4 + 64 <= kRankMaxHint;
}
int launch_tile_sort(const double *xyz, const int *n_ptr, int n_imm, size_t n_max, double voxel_size, unsigned long long *keys_in,
                     unsigned long long *keys_out, size_t n_hint, hipStream_t s, bool keys_ready) {
    if (n_max == 0) return 0;
    if (n_max > ((size_t)1 << 24)) return (int)hipErrorInvalidValue;  // 24 index bits
    const int runs = (int)((n_max + kSortRun - 1) / kSortRun);
    if (n_hint == 0 || n_hint > n_max) n_hint = n_max;
    if (tile_sort_by_rank(n_max, n_hint)) {  // a small cloud: one launch, by rank
        const size_t cover = n_hint + n_hint / 4 + 64 < n_max ? n_hint + n_hint / 4 + 64 : n_max;
        hipLaunchKernelGGL(k_tile_rank_sort, dim3((unsigned)((cover + kRankKeys - 1) / kRankKeys)), dim3(kRankThreads), 0, s, xyz, n_ptr, n_imm,
                           tile_sort_inv_cell(voxel_size), keys_ready ? keys_in : nullptr, keys_out);
        return (int)hipGetLastError();
    }
    const size_t runs_hint = (n_hint + n_hint / 4 + kSortRun - 1) / kSortRun + 1;
    int passes = 1;
    for (size_t cover = 12; runs_hint > cover && passes < 6; cover *= 8) ++passes;  // (the last pass takes ~a dozen runs gladly)
    unsigned long long *a = (passes & 1) ? keys_in : keys_out, *b = (passes & 1) ? keys_out : keys_in;  // ... so that the last pass ends in keys_out
    hipLaunchKernelGGL(k_tile_sort_blocks, dim3(runs), dim3(kSortThreads), 0, s, xyz, n_ptr, n_imm, tile_sort_inv_cell(voxel_size), a, keys_out);
    if (runs > 1) {
        const int grid = (int)((n_max + 255) / 256 < 2048 ? (n_max + 255) / 256 : 2048);
        long run_len = kSortRun;
        for (int k = 1; k <= passes; ++k) {
            // (run_len beyond n: one run, the pass copies)
            const int rl = (int)(run_len < ((long)1 << 24) ? run_len : ((long)1 << 24));
            hipLaunchKernelGGL(k_tile_merge_runs, dim3(grid), dim3(256), 0, s, a, b, n_ptr, n_imm, rl, k == passes ? 0 : 8);
            unsigned long long *t = a;
            a = b;
            b = t;
            run_len *= 8;
        }
    }
    return (int)hipGetLastError();
}

}  // namespace kicp
