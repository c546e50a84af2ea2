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
__global__ __launch_bounds__(kSortThreads) void k_tile_sort_blocks(const double *xyz, const int *n_ptr, int n_imm, double inv_cell,
                                                                   unsigned long long *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const int n = n_ptr ? *n_ptr : n_imm;
    const int first = (int)blockIdx.x * kSortRun;
    if (first >= n) return;
    const int cnt = min(kSortRun, n - first);
    int m = 64;  // padded to a power of two with keys that sort to the end
    while (m < cnt) m <<= 1;
    for (int i = threadIdx.x; i < m; i += kSortThreads) skeys[i] = i < cnt ? tile_key(xyz, first + i, inv_cell) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (m >> 1); i += kSortThreads) {
                const int lo = ((i / j) * 2 * j) + (i % j), hi = lo + j;
                const unsigned long long a = skeys[lo], b = skeys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    skeys[lo] = b;
                    skeys[hi] = a;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < cnt; i += kSortThreads) out[first + i] = skeys[i];
}

// one merge level: runs of `run` keys -> runs of 2 * run keys; a run without a partner is handed on as it is
__global__ __launch_bounds__(256) void k_tile_merge(const unsigned long long *in, unsigned long long *out, const int *n_ptr, int n_imm, int run) {
    const int n = n_ptr ? *n_ptr : n_imm;
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

// ---- run weights -------------------------------------------------------------------------------------------
// A workgroup's tile must hold the map voxels its run of source points can reach, and the map is far from
// uniform: next to the sensor voxels are full (max_points_per_voxel), far away they hold a point or two.  Runs
// of equal LENGTH would need tiles of very different sizes -- and an iteration is as slow as its slowest
// workgroup.  So runs are cut to equal WEIGHT: a point weighs kWeightBase plus the number of map points in the
// voxel it falls in under the initial guess (one lookup per point), and workgroup b takes the points whose
// exclusive weight prefix lies in [b W / G, (b + 1) W / G).

// weights and their inclusive prefix in ONE launch of one 1024-thread workgroup (the source cloud has a few
// thousand points, at most ~10^5: a multi-kernel device scan would cost more in launches -- on the serial chain
// of the frame, right in front of the registration -- than the work itself)
__global__ __launch_bounds__(1024) void k_tile_weights_scan(const unsigned long long *order, const double *frame, const int *n_ptr, int n_imm,
                                                            MapView m, const PipeState *state, int pipeline_mode, int weight_base, int weight_quad_in, int small_limit,
                                                            int *prefix) {
    __shared__ int wave_sum[16];
    constexpr int kLdsWeights = 24576;
    __shared__ unsigned short lds_w[kLdsWeights];  // (a weight is at most base + 255)
    const int n = n_ptr ? *n_ptr : n_imm;
    if (n < kIcpWeightedMin) return;  // k_icp cuts runs of equal LENGTH below this and never reads the prefix
    const SE3 guess = pipeline_mode ? se3_mul(state->last_pose, state->last_delta) : state->guess;
    const int t = threadIdx.x;
    const bool in_lds = n <= kLdsWeights && weight_base <= 1024;
    // The quadratic term (weight_quad_in < 0: automatic) is for clouds of at most small_limit points, i.e. runs of
    // a few dozen points with full-size voxels: there a workgroup's time is its tile's overflow and the number of
    // 16-point rounds, and dense runs must be SHORT (measured: 17.3 vs 19.8 us per iteration on the KITTI-like
    // scene with c^2 / 10); with hundreds of points per run the per-point work dominates and the term only
    // starves the sparse runs (1M-point configuration: 136 -> 153 us with c^2 / 16).  profiles/r02_ak, r02_al.
    const int weight_quad = weight_quad_in >= 0 ? weight_quad_in : (n <= small_limit ? 10 : 0);
    // pass 1 (coalesced): the weights themselves.  This launch sits on the serial chain of a frame, right in front
    // of the registration, and a thread's lookups are chains of dependent loads (sort key -> point -> map slot):
    // kBatch of them are kept in flight per thread, stage by stage, instead of one after the other.  (What is left
    // of the ~19 us of this launch on a KITTI-like frame is the chain itself -- count, key, point, slot: four
    // dependent round trips to memory written by other XCDs a moment ago -- and the launch.)
    constexpr int kBatch = 8;
    for (int q0 = t; q0 < n; q0 += 1024 * kBatch) {
        int pidx[kBatch];
        double pin[kBatch][3];
        unsigned long long key[kBatch];
        uint32_t sidx[kBatch];
        int cnt[kBatch];
        bool pend[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int q = q0 + 1024 * u;
            pidx[u] = q < n ? (order ? (int)(order[q] & 0xFFFFFFull) : q) : -1;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
            if (pidx[u] >= 0) {
                pin[u][0] = frame[3 * pidx[u]];
                pin[u][1] = frame[3 * pidx[u] + 1];
                pin[u][2] = frame[3 * pidx[u] + 2];
            }
        Slot first[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            cnt[u] = 0;
            pend[u] = false;
            first[u].key = kKeyEmpty;
            first[u].count = 0;
            if (pidx[u] >= 0) {
                double sp[3];
                se3_act(guess, pin[u], sp);
                const int vx = voxel_coord(sp[0], m.voxel_size), vy = voxel_coord(sp[1], m.voxel_size), vz = voxel_coord(sp[2], m.voxel_size);
                if (voxel_in_range(vx, vy, vz)) {
                    key[u] = pack_voxel(vx, vy, vz);
                    sidx[u] = hash_key(key[u], m.mask);
                    pend[u] = true;
                    first[u] = load_slot(m.slots + sidx[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (pend[u]) {
                if (first[u].key == key[u]) {
                    cnt[u] = first[u].count;
                } else if (first[u].key != kKeyEmpty) {  // the rare longer chain: walk it
                    uint32_t si = (sidx[u] + 1) & m.mask;
                    for (uint32_t probes = 1; probes <= m.mask; ++probes) {
                        const Slot sl = load_slot(m.slots + si);
                        if (sl.key == key[u]) {
                            cnt[u] = sl.count;
                            break;
                        }
                        if (sl.key == kKeyEmpty) break;
                        si = (si + 1) & m.mask;
                    }
                }
            }
            const int q = q0 + 1024 * u;
            if (q < n) {
                // base + c + c^2 / quad: the points a run's tile must hold grow faster than linearly with the
                // population c of the voxels around it (full voxels have full neighbours)
                const int w = weight_base + cnt[u] + (weight_quad > 0 ? (cnt[u] * cnt[u]) / weight_quad : 0);
                if (in_lds) lds_w[q] = (unsigned short)min(w, 0xFFFF);
                else prefix[q] = w;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // pass 2: thread t owns the contiguous slice [t E, (t + 1) E).  Up to kLdsWeights points (any full-size-voxel
    // scan) the weights never leave the workgroup: re-reading them from L2 point by point was most of this launch.
    const int E = (n + 1023) / 1024;
    const int a = min(n, t * E), b = min(n, a + E);
    int sum = 0;
    if (in_lds) {
        for (int q = a; q < b; ++q) sum += (int)lds_w[q];
    } else {
        for (int q = a; q < b; ++q) sum += __hip_atomic_load(&prefix[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
    if (in_lds) {
        for (int q = a; q < b; ++q) {
            run += (int)lds_w[q];
            prefix[q] = run;
        }
    } else {
        for (int q = a; q < b; ++q) {
            run += __hip_atomic_load(&prefix[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prefix[q] = run;
        }
    }
}

int launch_tile_weights(const unsigned long long *order, const double *frame, const int *n_ptr, int n_imm, size_t n_max, const MapView &m,
                        const PipeState *state, int pipeline_mode, int weight_base, int weight_quad, int small_limit, int *prefix,
                        hipStream_t s) {
    if (n_max == 0) return 0;
    hipLaunchKernelGGL(k_tile_weights_scan, dim3(1), dim3(1024), 0, s, order, frame, n_ptr, n_imm, m, state, pipeline_mode, weight_base, weight_quad, small_limit, prefix);
    return (int)hipGetLastError();
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
                       1.0 / (2.0 * voxel_size), a);
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
