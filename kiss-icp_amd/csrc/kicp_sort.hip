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
// keys (round 2).  Here: k_tile_sort_blocks computes the keys and sorts runs of up to 16384 of them in LDS (bitonic
// network, in place; workgroups beyond N_src leave at once) -- one workgroup and ~15 us for any full-size-voxel scan --
// and k_tile_merge passes (as many as the host-side bound needs; a pass with nothing to merge just hands the run on)
// merge pairs of runs by rank: position = own rank + number of smaller keys in the partner run (binary search; the
// keys are unique).  The result is THE ascending order of the keys, whatever produced it.
#include <cstring>
#include <mutex>

#include <hip/hip_runtime.h>

#include "kicp_search.hpp"

namespace kicp {

__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ unsigned long long tile_key(const double *xyz, int i, double inv_cell) {
    // 2-voxel cells, offset so that +-512 cells around the sensor map to 0..1023 (farther points clamp: only the
    // quality of the order is at stake)
    const double cx = floor(xyz[3 * i] * inv_cell) + 512.0, cy = floor(xyz[3 * i + 1] * inv_cell) + 512.0,
                 cz = floor(xyz[3 * i + 2] * inv_cell) + 512.0;
    const unsigned ux = (unsigned)fmin(fmax(cx, 0.0), 1023.0), uy = (unsigned)fmin(fmax(cy, 0.0), 1023.0),
                   uz = (unsigned)fmin(fmax(cz, 0.0), 1023.0);
    const unsigned long long m = (unsigned long long)(spread10(ux) | (spread10(uy) << 1) | (spread10(uz) << 2));
    return (m << 24) | (unsigned long long)(unsigned)i;  // morton30(cell of point i) << 24 | i: unique
}

constexpr int kSortRun = 16384;     // keys one workgroup sorts in LDS (128 KiB)
constexpr int kSortThreads = 1024;

// run r = keys of the points [r * kSortRun, min(n, (r + 1) * kSortRun)), sorted, written to out at the same positions
// (a cloud of one run -- any full-size-voxel scan -- goes straight to `final`: the merge passes then have nothing to do)
__global__ __launch_bounds__(kSortThreads) void k_tile_sort_blocks(const double *xyz, const int *n_ptr, int n_imm, double inv_cell,
                                                                   unsigned long long *runs, unsigned long long *final) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
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
    // segment of 128 elements whenever j <= 64, and the same segment in the next such stage: those stages (70 of the
    // 91 for 8192 keys) need ordering inside the wave only, not a workgroup barrier.
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

// one merge level: runs of `run` keys -> runs of 2 * run keys; a run without a partner is handed on as it is
__global__ __launch_bounds__(256) void k_tile_merge(const unsigned long long *in, unsigned long long *out, const int *n_ptr, int n_imm, int run) {
    const int n = n_ptr ? *n_ptr : n_imm;
    if (n <= kSortRun) return;  // a single run: k_tile_sort_blocks has already put it where the last pass would
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = in[i];
        const int r = i / run, base = (r & ~1) * run;
        const int p0 = (r ^ 1) * run;  // the partner run
        int smaller = 0;
        if (p0 < n) {
            const int plen = min(run, n - p0);
            // number of partner keys below this one (keys are unique, so no tie rule is needed)
            int lo = 0, hi = plen;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (in[p0 + mid] < key) lo = mid + 1;
                else hi = mid;
            }
            smaller = lo;
        }
        out[base + (i - r * run) + smaller] = key;
    }
}

size_t tile_sort_temp_bytes(size_t) { return 256; }  // (the sort needs no scratch beyond its two key buffers)

int tile_sort_prepare(int device_id) {
    // the block sort's 128 KiB of dynamic LDS need the opt-in attribute, once per device
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device_id < 0 || device_id >= 64) return (int)hipErrorInvalidDevice;
    if (done[device_id]) return 0;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_sort_blocks), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kSortRun * (int)sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    done[device_id] = true;
    return 0;
}

// sorted keys of the cloud -> keys_out; keys_in is the other buffer of the ping-pong (both hold n_max keys)
int launch_tile_sort(const double *xyz, const int *n_ptr, int n_imm, size_t n_max, double voxel_size, unsigned long long *keys_in,
                     unsigned long long *keys_out, void *, size_t, hipStream_t s) {
    if (n_max == 0) return 0;
    if (n_max > ((size_t)1 << 24)) return (int)hipErrorInvalidValue;  // 24 index bits
    const int runs = (int)((n_max + kSortRun - 1) / kSortRun);
    int passes = 0;
    while ((1 << passes) < runs) ++passes;
    // the passes alternate between the two buffers; the block sort starts in the one that makes the last pass end in keys_out
    unsigned long long *a = (passes & 1) ? keys_in : keys_out, *b = (passes & 1) ? keys_out : keys_in;
    hipLaunchKernelGGL(k_tile_sort_blocks, dim3(runs), dim3(kSortThreads), kSortRun * sizeof(unsigned long long), s, xyz, n_ptr, n_imm,
                       1.0 / (2.0 * voxel_size), a, keys_out);
    const int grid = (int)((n_max + 255) / 256 < 1024 ? (n_max + 255) / 256 : 1024);
    for (int l = 0; l < passes; ++l) {
        hipLaunchKernelGGL(k_tile_merge, dim3(grid), dim3(256), 0, s, a, b, n_ptr, n_imm, kSortRun << l);
        unsigned long long *t = a;
        a = b;
        b = t;
    }
    return (int)hipGetLastError();
}

}  // namespace kicp
