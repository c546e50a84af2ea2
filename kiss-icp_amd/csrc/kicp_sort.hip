// kicp_sort.hip -- spatial order of the source cloud for the persistent ICP kernel.
//
// k_icp gives every workgroup a CONTIGUOUS run of the source cloud and keeps the map voxels that run
// needs in LDS (its "tile").  For the tile to be small the run must be compact in space, so the cloud is
// ordered by {Morton code of the point's 2-voxel cell in the sensor frame, original index}: the index
// makes the keys unique, hence the order -- and with it every floating-point sum downstream --
// deterministic.  Rigid motion preserves neighbourhoods, so the sensor frame is as good as the map frame
// (and is available before the pose is).  The sort itself is rocPRIM's device radix sort (a plain library
// sort; everything on the registration path proper is hand-written).
#include <cstring>

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "kicp_launch.hpp"

namespace kicp {

__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// keys[i] = morton30(cell of point i) << 24 | i for i < n, all ones behind (they sort to the end)
__global__ __launch_bounds__(256) void k_tile_keys(const double *xyz, const int *n_ptr, int n_imm, int n_max, double inv_cell,
                                                   unsigned long long *keys) {
    const int n = n_ptr ? *n_ptr : n_imm;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_max; i += gridDim.x * blockDim.x) {
        unsigned long long k = ~0ull;
        if (i < n) {
            // 2-voxel cells, offset so that +-512 cells around the sensor map to 0..1023 (farther points clamp:
            // only the quality of the order is at stake)
            const double cx = floor(xyz[3 * i] * inv_cell) + 512.0, cy = floor(xyz[3 * i + 1] * inv_cell) + 512.0,
                         cz = floor(xyz[3 * i + 2] * inv_cell) + 512.0;
            const unsigned ux = (unsigned)fmin(fmax(cx, 0.0), 1023.0), uy = (unsigned)fmin(fmax(cy, 0.0), 1023.0),
                           uz = (unsigned)fmin(fmax(cz, 0.0), 1023.0);
            const unsigned long long m = (unsigned long long)(spread10(ux) | (spread10(uy) << 1) | (spread10(uz) << 2));
            k = (m << 24) | (unsigned long long)(unsigned)i;
        }
        keys[i] = k;
    }
}

// ---- run weights -------------------------------------------------------------------------------------------
// A workgroup's tile must hold the map voxels its run of source points can reach, and the map is far from
// uniform: next to the sensor voxels are full (max_points_per_voxel), far away they hold a point or two.  Runs
// of equal LENGTH would need tiles of very different sizes -- and an iteration is as slow as its slowest
// workgroup.  So runs are cut to equal WEIGHT: a point weighs kWeightBase plus the number of map points in the
// voxel it falls in under the initial guess (one lookup per point), and workgroup b takes the points whose
// exclusive weight prefix lies in [b W / G, (b + 1) W / G).

// weight of the point at sorted position q under the initial guess
__device__ __forceinline__ int tile_weight(const unsigned long long *order, const double *frame, const MapView &m, const SE3 &guess, int q, int weight_base) {
    const int p = order ? (int)(order[q] & 0xFFFFFFull) : q;
    const double pin[3] = {frame[3 * p], frame[3 * p + 1], frame[3 * p + 2]};
    double s[3];
    se3_act(guess, pin, s);
    const int vx = voxel_coord(s[0], m.voxel_size), vy = voxel_coord(s[1], m.voxel_size), vz = voxel_coord(s[2], m.voxel_size);
    int cnt = 0;
    if (voxel_in_range(vx, vy, vz)) {
        const unsigned long long key = pack_voxel(vx, vy, vz);
        uint32_t sidx = hash_key(key, m.mask);
        for (uint32_t probes = 0; probes <= m.mask; ++probes) {
            const unsigned long long k = m.slots[sidx].key;
            if (k == key) {
                cnt = m.slots[sidx].count;
                break;
            }
            if (k == kKeyEmpty) break;
            sidx = (sidx + 1) & m.mask;
        }
    }
    return weight_base + cnt;
}

// weights and their inclusive prefix in ONE launch of one 1024-thread workgroup (the source cloud has a few
// thousand points, at most ~10^5: a multi-kernel device scan would cost more in launches -- on the serial chain
// of the frame, right in front of the registration -- than the work itself)
__global__ __launch_bounds__(1024) void k_tile_weights_scan(const unsigned long long *order, const double *frame, const int *n_ptr, int n_imm,
                                                            MapView m, const PipeState *state, int pipeline_mode, int weight_base, int *prefix) {
    __shared__ int wave_sum[16];
    const int n = n_ptr ? *n_ptr : n_imm;
    const SE3 guess = pipeline_mode ? se3_mul(state->last_pose, state->last_delta) : state->guess;
    const int t = threadIdx.x;
    // pass 1 (coalesced): the weights themselves
    for (int q = t; q < n; q += 1024) prefix[q] = tile_weight(order, frame, m, guess, q, weight_base);
    __threadfence_block();
    __syncthreads();
    // pass 2: thread t owns the contiguous slice [t E, (t + 1) E)
    const int E = (n + 1023) / 1024;
    const int a = min(n, t * E), b = min(n, a + E);
    int sum = 0;
    for (int q = a; q < b; ++q) sum += __hip_atomic_load(&prefix[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // exclusive scan of the 1024 slice sums
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if ((t & 63) >= o) incl += up;
    }
    if ((t & 63) == 63) wave_sum[t >> 6] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) base += (w < (t >> 6)) ? wave_sum[w] : 0;
    int run = base + incl - sum;
    for (int q = a; q < b; ++q) {
        run += __hip_atomic_load(&prefix[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prefix[q] = run;
    }
}

int launch_tile_weights(const unsigned long long *order, const double *frame, const int *n_ptr, int n_imm, size_t n_max, const MapView &m,
                        const PipeState *state, int pipeline_mode, int weight_base, int *prefix, hipStream_t s) {
    if (n_max == 0) return 0;
    hipLaunchKernelGGL(k_tile_weights_scan, dim3(1), dim3(1024), 0, s, order, frame, n_ptr, n_imm, m, state, pipeline_mode, weight_base, prefix);
    return (int)hipGetLastError();
}

size_t tile_sort_temp_bytes(size_t n_max) {
    size_t bytes = 0;
    unsigned long long *k = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, bytes, k, k, n_max ? n_max : 1, 0, 54, (hipStream_t) nullptr);
    return bytes + 256;
}

int launch_tile_sort(const double *xyz, const int *n_ptr, int n_imm, size_t n_max, double voxel_size, unsigned long long *keys_in,
                     unsigned long long *keys_out, void *temp, size_t temp_bytes, hipStream_t s) {
    if (n_max == 0) return 0;
    if (n_max > ((size_t)1 << 24)) return (int)hipErrorInvalidValue;  // 24 index bits
    const int grid = (int)((n_max + 255) / 256 < 1024 ? (n_max + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_tile_keys, dim3(grid), dim3(256), 0, s, xyz, n_ptr, n_imm, (int)n_max, 1.0 / (2.0 * voxel_size), keys_in);
    const hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n_max, 0, 54, s);
    return (int)e;
}

}  // namespace kicp
