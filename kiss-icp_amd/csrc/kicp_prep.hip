// kicp_prep.hip -- the stages in front of the registration, as order-preserving compactions:
//   k_ts_minmax, k_pre_*   Preprocessor::Preprocess   core/Preprocessing.cpp:55-95
//   k_ds_*                 VoxelDownsample            core/VoxelUtils.cpp:7-21
#include "kicp_launch.hpp"

namespace kicp {

__device__ __forceinline__ int count_of(const int *n_ptr, int n_imm) { return n_ptr ? *n_ptr : n_imm; }

static inline int grid_for(long n, int threads, int cap) {
    long g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// order-preserving compaction helpers (1024-thread workgroups, one element per thread)
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

// exclusive position of this thread's flag inside the workgroup + workgroup total
__device__ __forceinline__ int block_exclusive_scan(bool flag, int &total) {
    __shared__ int wave_tot[kScanThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long ball = __ballot(flag);
    const int before = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(ball);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        const int c = wave_tot[w];
        off += (w < wave) ? c : 0;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return off + before;
}

// sum of counts[0 .. blockIdx.x) -- every workgroup recomputes its own base (counts are few)
__device__ __forceinline__ int block_base(const int *counts, int *grand_total) {
    __shared__ int sh_base, sh_all;
    if (threadIdx.x < 64) {
        int b = 0, a = 0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) {
            const int c = counts[i];
            a += c;
            if (i < (int)blockIdx.x) b += c;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            b += __shfl_xor(b, off, 64);
            a += __shfl_xor(a, off, 64);
        }
        if (threadIdx.x == 0) {
            sh_base = b;
            sh_all = a;
        }
    }
    __syncthreads();
    if (grand_total) *grand_total = sh_all;
    return sh_base;
}

__device__ __forceinline__ unsigned long long f64_order_bits(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_from_order_bits(unsigned long long o) {
    const unsigned long long b = (o & 0x8000000000000000ull) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    return __longlong_as_double((long long)b);
}

// ---- Preprocess ---------------------------------------------------------------------------
// min / max of the timestamps (Preprocessing.cpp:62)
__global__ __launch_bounds__(256) void k_ts_minmax(const double *ts, int n_ts, PrepState *st) {
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ts; i += gridDim.x * blockDim.x) {
        const unsigned long long o = f64_order_bits(ts[i]);
        lo = o < lo ? o : lo;
        hi = o > hi ? o : hi;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&st->tmin_bits, lo);
        atomicMax(&st->tmax_bits, hi);
    }
}

// deskew (Preprocessing.cpp:59-84) into tmp[], range test (:86-92), workgroup counts
__global__ __launch_bounds__(kScanThreads) void k_pre_flags(PreParams P) {
    const int n = P.n;
    const int i = blockIdx.x * kScanThreads + threadIdx.x;
    bool keep = false;
    if (i < n) {
        double p[3];
        if (P.xyz_f32) {  // float32 sensor data (or a float64 scan narrowed losslessly by the host): widening is exact
            const float *f = static_cast<const float *>(P.xyz);
            p[0] = (double)f[3 * i];
            p[1] = (double)f[3 * i + 1];
            p[2] = (double)f[3 * i + 2];
        } else {
            const double *d = static_cast<const double *>(P.xyz);
            p[0] = d[3 * i];
            p[1] = d[3 * i + 1];
            p[2] = d[3 * i + 2];
        }
        if (P.deskew) {
            const double mn = f64_from_order_bits(P.prep->tmin_bits);
            const double mx = f64_from_order_bits(P.prep->tmax_bits);
            double omega[6];
            se3_log(P.use_state_motion ? P.state->last_delta : P.motion, omega);
            const double stamp = (P.ts[i] - mn) / (mx - mn);
            double a[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = (stamp - 1.0) * omega[k];
            const SE3 pose = se3_exp(a);
            double o[3];
            se3_act(pose, p, o);
            p[0] = o[0];
            p[1] = o[1];
            p[2] = o[2];
        }
        P.tmp[3 * i] = p[0];
        P.tmp[3 * i + 1] = p[1];
        P.tmp[3 * i + 2] = p[2];
        const double r = sqrt(sqnorm3(p[0], p[1], p[2]));
        keep = (r < P.max_range) && (r > P.min_range);
    }
    int total;
    block_exclusive_scan(keep, total);
    if (threadIdx.x == 0) P.blk_counts[blockIdx.x] = total;
}

// ---- the reference's grid (DsParams::order == 1) --------------------------------------------------------------
// std::hash<Voxel> (core/VoxelUtils.hpp:46-50): u32 wrap-around products, xor-ed; bucket = hash & (bucket_count - 1)
__device__ __forceinline__ uint32_t ref_home(unsigned long long key, uint32_t mask) {
    int x, y, z;
    unpack_voxel(key, x, y, z);
    return (((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349669u) ^ ((uint32_t)z * 83492791u)) & mask;
}
// grid.reserve(n) (VoxelUtils.cpp:10) = rehash(ceil(float(n) / max_load_factor 0.5f)) rounded up to a power of two
__device__ __forceinline__ uint32_t ref_grid_mask(int n) {
    if (n <= 0) return 0u;
    const unsigned want = (unsigned)ceilf((float)n / 0.5f);
    unsigned b = 1u;
    while (b < want) b <<= 1;
    return b - 1u;
}

// find-or-claim the slot of a voxel in the downsample scratch table
__device__ __forceinline__ int ds_claim(DsSlot *tab, uint32_t mask, unsigned long long key, int ref = 0) {
    // Read before CAS: ~15 scan points share a voxel, so most arrivals find their key already
    // there and never touch the atomic unit.  Keys are stable for the lifetime of a claim phase,
    // so a (possibly L1-stale) plain read can only cost an extra CAS, never a wrong answer.
    uint32_t s = ref ? ref_home(key, mask) : hash_key(key, mask);
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        unsigned long long cur = tab[s].key;
        if (cur == kKeyEmpty) cur = atomicCAS(&tab[s].key, kKeyEmpty, key);
        if (cur == kKeyEmpty || cur == key) return (int)s;
        s = (s + 1) & mask;
    }
    return -1;
}
// atomicMin that skips the atomic when a plain read already shows a smaller index (the stored
// index only ever decreases, so a stale read can only cause a redundant atomic)
__device__ __forceinline__ void ds_min_index(DsSlot *tab, int s, int idx) {
    if (tab[s].minidx > idx) atomicMin(&tab[s].minidx, idx);
}

// Workgroup-level aggregation of downsample claims: the (up to 1024) points of a workgroup are
// consecutive in the scan, so most of them share a voxel with a neighbour.  They first meet in a
// small LDS hash (LDS atomics: no HBM traffic, no cross-XCD contention); only one claim and one
// atomicMin per DISTINCT voxel of the workgroup then go to the table in HBM.
constexpr int kAggSlots = 2048;
struct ClaimAgg {
    unsigned long long key[kAggSlots];
    int minidx[kAggSlots];
    int slot[kAggSlots];
};

// all kScanThreads threads call this; returns the table slot of this thread's voxel (-1: none)
__device__ __forceinline__ int ds_claim_aggregated(ClaimAgg &agg, DsSlot *tab, uint32_t mask, bool valid,
                                                   unsigned long long key, int idx, int *err, int ref = 0) {
    for (int e = threadIdx.x; e < kAggSlots; e += kScanThreads) {
        agg.key[e] = kKeyEmpty;
        agg.minidx[e] = 0x7FFFFFFF;
    }
    __syncthreads();
    int ls = -1;
    if (valid) {
        uint32_t h = hash_key(key, kAggSlots - 1);
        for (int probes = 0; probes < kAggSlots; ++probes) {
            unsigned long long cur = agg.key[h];
            if (cur == kKeyEmpty) cur = atomicCAS(&agg.key[h], kKeyEmpty, key);
            if (cur == kKeyEmpty || cur == key) {
                ls = (int)h;
                break;
            }
            h = (h + 1) & (kAggSlots - 1);
        }
        atomicMin(&agg.minidx[ls], idx);  // 1024 points never fill 2048 slots: ls >= 0
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kAggSlots; e += kScanThreads) {
        const unsigned long long k = agg.key[e];
        if (k == kKeyEmpty) continue;
        const int s = ds_claim(tab, mask, k, ref);
        if (s >= 0) ds_min_index(tab, s, agg.minidx[e]);
        else atomicOr(err, E_TABLE_FULL);
        agg.slot[e] = s;
    }
    __syncthreads();
    return valid ? agg.slot[ls] : -1;
}

// scatter the range-cropped cloud (order preserving) and, fused, stage A of the first
// VoxelDownsample: claim the voxel and atomicMin the (new) point index into it
__global__ __launch_bounds__(kScanThreads) void k_pre_scatter(PreParams P) {
    __shared__ ClaimAgg agg;
    const int n = P.n;
    const int i = blockIdx.x * kScanThreads + threadIdx.x;
    bool keep = false;
    double p[3] = {0, 0, 0};
    if (i < n) {
        p[0] = P.tmp[3 * i];
        p[1] = P.tmp[3 * i + 1];
        p[2] = P.tmp[3 * i + 2];
        const double r = sqrt(sqnorm3(p[0], p[1], p[2]));
        keep = (r < P.max_range) && (r > P.min_range);
    }
    int total, grand;
    const int base = block_base(P.blk_counts, &grand);
    const int j = base + block_exclusive_scan(keep, total);
    if (keep) {
        P.out[3 * j] = p[0];
        P.out[3 * j + 1] = p[1];
        P.out[3 * j + 2] = p[2];
    }
    if (P.ds_tab) {
        bool valid = false;
        unsigned long long key = 0;
        if (keep) {
            const int vx = voxel_coord(p[0], P.ds_voxel), vy = voxel_coord(p[1], P.ds_voxel),
                      vz = voxel_coord(p[2], P.ds_voxel);
            valid = voxel_in_range(vx, vy, vz);
            if (valid) key = pack_voxel(vx, vy, vz);
            else atomicOr(P.err, E_RANGE);
        }
        // (the reference's grid has 2^ceil(log2(2 n)) buckets, n = the points this stage hands on: `grand`)
        const int s = ds_claim_aggregated(agg, P.ds_tab, P.ds_order ? ref_grid_mask(grand) : P.ds_mask, valid, key, j, P.err, P.ds_order);
        if (keep) P.ds_slot_of[j] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *P.n_out = grand;
        if (P.prep) {  // k_pre_flags (the only reader) is done: re-arm the timestamp words
            P.prep->tmin_bits = ~0ull;
            P.prep->tmax_bits = 0ull;
        }
    }
}

// ---- VoxelDownsample ----------------------------------------------------------------------
// stage A standalone (when the input is not produced by a fused scatter)
__global__ __launch_bounds__(256) void k_ds_claim(DsParams P) {
    const int n = count_of(P.n_ptr, P.n_imm);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int vx = voxel_coord(P.in[3 * i], P.voxel), vy = voxel_coord(P.in[3 * i + 1], P.voxel),
                  vz = voxel_coord(P.in[3 * i + 2], P.voxel);
        int s = -1;
        if (voxel_in_range(vx, vy, vz)) {
            s = ds_claim(P.tab, P.order ? ref_grid_mask(n) : P.mask, pack_voxel(vx, vy, vz), P.order);
            if (s >= 0) ds_min_index(P.tab, s, i);
            else atomicOr(P.err, E_TABLE_FULL);
        } else {
            atomicOr(P.err, E_RANGE);
        }
        P.slot_of[i] = s;
    }
}

// stage B: a point survives iff it is the first (lowest index) of its voxel
__global__ __launch_bounds__(kScanThreads) void k_ds_flags(DsParams P) {
    const int n = count_of(P.n_ptr, P.n_imm);
    const int i = blockIdx.x * kScanThreads + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const int s = P.slot_of[i];
        keep = (s >= 0) && (P.tab[s].minidx == i);
    }
    int total;
    block_exclusive_scan(keep, total);
    if (threadIdx.x == 0) P.blk_counts[blockIdx.x] = total;
}

// stage C: scatter in ascending original index, wipe the scratch slot, and (fused) stage A of the
// next downsample on the surviving point
__global__ __launch_bounds__(kScanThreads) void k_ds_scatter(DsParams P) {
    const int n = count_of(P.n_ptr, P.n_imm);
    const int i = blockIdx.x * kScanThreads + threadIdx.x;
    bool keep = false;
    int s = -1;
    if (i < n) {
        s = P.slot_of[i];
        if (s >= 0) s = KICP_IDX(P.dbg, P.err, s, (long long)P.mask + 1, 21);
        keep = (s >= 0) && (P.tab[s].minidx == i);
    }
    int total, grand;
    const int base = block_base(P.blk_counts, &grand);
    const int j = base + block_exclusive_scan(keep, total);
    if (keep) {
        const double x = P.in[3 * i], y = P.in[3 * i + 1], z = P.in[3 * i + 2];
        P.out[3 * j] = x;
        P.out[3 * j + 1] = y;
        P.out[3 * j + 2] = z;
        P.tab[s].key = kKeyEmpty;  // each claimed slot has exactly one winner: self-cleaning
        P.tab[s].minidx = 0x7FFFFFFF;
        if (P.sort_keys) P.sort_keys[j] = tile_key_of(x, y, z, j, P.sort_inv_cell);
        if (P.next_tab) {
            const int vx = voxel_coord(x, P.next_voxel), vy = voxel_coord(y, P.next_voxel),
                      vz = voxel_coord(z, P.next_voxel);
            int s2 = -1;
            if (voxel_in_range(vx, vy, vz)) {
                s2 = ds_claim(P.next_tab, P.next_mask, pack_voxel(vx, vy, vz));
                if (s2 >= 0) ds_min_index(P.next_tab, s2, j);
                else atomicOr(P.err, E_TABLE_FULL);
            } else {
                atomicOr(P.err, E_RANGE);
            }
            P.next_slot_of[j] = s2;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.n_out = grand;
}

// ---- the reference's output order (DsParams::order == 1) --------------------------------------------------------
// After the claims the table holds, for every distinct voxel, {key, index of its first point} somewhere in the run of
// occupied buckets that starts at or before its home: the same buckets the reference's robin-hood table occupies
// (linear probing fills the same set whatever the insertion order and whoever displaces whom).  What the reference
// fixes beyond that is who sits where INSIDE a maximal run of occupied buckets (a cluster) -- and clusters never
// interact.  tsl::robin_map 1.4.0 inserts like this (oracle/ref_build/shim/tsl/robin_map.h states the rules): the new
// element walks from its home bucket; at each bucket it takes the place of an occupant that is STRICTLY closer to its
// own home -- i.e. whose home lies further right -- and the occupant walks on by the same rule; at an empty bucket the
// walker settles.  (Equal homes: the walker moves on, so a group of one home is rotated by an insertion in front of
// it -- the result is NOT simply "sorted by home, then by arrival".)  k_ds_arrange replays these insertions cluster
// by cluster, in arrival order = ascending index of the voxel's first point: a lone element needs nothing; a cluster of
// 2..64 is replayed by ONE WAVE with the table in its lanes -- an insertion is two ballots and a shuffle, because
// only the head of each home group behind the insertion point moves (to where the next group's head was); anything
// longer (the table is at most half full: mean cluster 1.1, longest ~20 on a 130k-point scan) by one lane in memory.
// one cluster of at most 64 buckets, replayed by one wave: lane j holds what sits at the cluster's j-th bucket
__device__ __forceinline__ void ds_arrange_wave(const DsParams &P, uint32_t mask, int b, int c, int t, int h) {
    const int lane = threadIdx.x & 63;
    // arrival order: rank of this lane's element among the first-point indices (all distinct)
    int rank = 0;
    for (int i = 0; i < c; ++i) rank += (__shfl(t, i, 64) < t) ? 1 : 0;
    int Et = -1, Eh = 0;  // the table being rebuilt: element (first-point index, relative home) at relative position `lane`
    for (int k = 0; k < c; ++k) {
        const unsigned long long who = __ballot(lane < c && rank == k);
        const int src = __ffsll((long long)who) - 1;
        const int zt = __shfl(t, src, 64), zh = __shfl(h, src, 64);
        // the walker stops at the first position >= its home that is empty or whose occupant's home lies further right
        const unsigned long long stop = __ballot(Et < 0 || Eh > zh) & (~0ull << zh);
        const int p = __ffsll((long long)stop) - 1;
        const unsigned long long occ = __ballot(Et >= 0);
        if ((occ >> p) & 1ull) {
            // occupied: the occupant -- the FIRST of its home group -- walks on past its own group and takes the place
            // of the next group's first element, and so on up to the first empty position g.  Only group heads move.
            const unsigned long long gt_p = (p >= 63) ? 0ull : (~0ull << (p + 1));
            const int g = __ffsll((long long)(~occ & gt_p)) - 1;
            const int prev_h = __shfl_up(Eh, 1, 64);
            const unsigned long long range = (~0ull << p) & ((1ull << g) - 1ull);
            const unsigned long long heads = __ballot(lane == p || Eh != prev_h) & range;
            const unsigned long long recv = (heads & gt_p) | (1ull << g);
            const unsigned long long before = heads & ((1ull << lane) - 1ull);
            const int from = before ? 63 - __clzll((long long)before) : 0;
            const int nt = __shfl(Et, from, 64), nh = __shfl(Eh, from, 64);
            if ((recv >> lane) & 1ull) {
                Et = nt;
                Eh = nh;
            }
        }
        if (lane == p) {
            Et = zt;
            Eh = zh;
        }
    }
    if (lane < c) P.rb_elem[((uint32_t)b + (uint32_t)lane) & mask] = Et;
}

// a cluster longer than a wave (an adversarial cloud: the table is at most half full): one lane replays it in global memory
__device__ void ds_arrange_serial(const DsParams &P, uint32_t mask, int b) {
    int c = 0;
    while ((uint32_t)c <= mask && P.tab[((uint32_t)b + (uint32_t)c) & mask].key != kKeyEmpty) ++c;
    if ((uint32_t)c > mask) {  // no empty bucket at all: the grid is at most half full by construction, so this is a broken table -- say so, do not spin
        atomicOr(P.err, E_TABLE_FULL);
        return;
    }
    for (int j = 0; j < c; ++j) P.rb_elem[((uint32_t)b + (uint32_t)j) & mask] = -1;
    int last = -1;
    for (int step = 0; step < c; ++step) {
        int best = 0x7FFFFFFF, bj = 0;  // the next arrival: the smallest first-point index above the previous one
        for (int j = 0; j < c; ++j) {
            const int t = P.tab[((uint32_t)b + (uint32_t)j) & mask].minidx;
            if (t > last && t < best) {
                best = t;
                bj = j;
            }
        }
        last = best;
        const unsigned long long key = P.tab[((uint32_t)b + (uint32_t)bj) & mask].key;
        int cur_t = best;
        int cur_h = (int)((ref_home(key, mask) - (uint32_t)b) & mask);
        for (int pos = cur_h; pos < c; ++pos) {  // (a walker settles inside its cluster: c buckets hold c elements)
            const uint32_t slot = ((uint32_t)b + (uint32_t)pos) & mask;
            const int et = P.rb_elem[slot];
            if (et < 0) {
                P.rb_elem[slot] = cur_t;
                P.rb_home[slot] = cur_h;
                break;
            }
            const int eh = P.rb_home[slot];
            if (cur_h < eh) {  // the occupant is strictly closer to its home: it gives way and walks on
                P.rb_elem[slot] = cur_t;
                P.rb_home[slot] = cur_h;
                cur_t = et;
                cur_h = eh;
            }
        }
    }
}

__global__ __launch_bounds__(kScanThreads) void k_ds_arrange(DsParams P) {
    __shared__ int wl_b[kScanThreads / 2];  // first buckets of this workgroup's clusters of two or more
    __shared__ int wl_n;
    const int n = count_of(P.n_ptr, P.n_imm);
    const uint32_t mask = ref_grid_mask(n);
    if (n <= 0 || (uint32_t)blockIdx.x * kScanThreads > mask) {  // beyond the grid: nothing here
        if (threadIdx.x == 0) P.blk_counts[blockIdx.x] = 0;
        return;
    }
    if (threadIdx.x == 0) wl_n = 0;
    const int b = blockIdx.x * kScanThreads + threadIdx.x;
    const bool occ = (uint32_t)b <= mask && P.tab[b].key != kKeyEmpty;
    int total;
    block_exclusive_scan(occ, total);  // (contains the barrier that publishes wl_n)
    if (threadIdx.x == 0) P.blk_counts[blockIdx.x] = total;
    if (occ && P.tab[((uint32_t)b - 1u) & mask].key == kKeyEmpty) {  // first bucket of a cluster
        if (P.tab[((uint32_t)b + 1u) & mask].key == kKeyEmpty) P.rb_elem[b] = P.tab[b].minidx;  // alone: nothing to settle
        else wl_b[atomicAdd(&wl_n, 1)] = b;  // (two consecutive buckets cannot both be first buckets: at most 512)
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = wave; w < wl_n; w += kScanThreads / 64) {
        const int cb = wl_b[w];
        const uint32_t slot = ((uint32_t)cb + (uint32_t)lane) & mask;
        const unsigned long long key = P.tab[slot].key;
        const unsigned long long empties = __ballot(key == kKeyEmpty);
        if (empties == 0ull) {  // longer than a wave
            if (lane == 0) ds_arrange_serial(P, mask, cb);
            continue;
        }
        const int c = __ffsll((long long)empties) - 1;
        int t = 0x7FFFFFFF, h = 0;
        if (lane < c) {
            t = P.tab[slot].minidx;
            h = (int)((ref_home(key, mask) - (uint32_t)cb) & mask);  // home relative to the cluster's first bucket: 0 .. c-1
        }
        ds_arrange_wave(P, mask, cb, c, t, h);
    }
}

// bucket-order compaction: the survivor of bucket b goes to position (occupied buckets before b); wipes the table;
// fused, stage A of the next downsample on the emitted point
__global__ __launch_bounds__(kScanThreads) void k_ds_scatter_rb(DsParams P) {
    const int n = count_of(P.n_ptr, P.n_imm);
    const uint32_t mask = ref_grid_mask(n);
    if (blockIdx.x != 0 && (n <= 0 || (uint32_t)blockIdx.x * kScanThreads > mask)) return;  // beyond the grid (its count is 0)
    const int b = blockIdx.x * kScanThreads + threadIdx.x;
    const bool occ = n > 0 && (uint32_t)b <= mask && P.tab[b].key != kKeyEmpty;
    int total, grand;
    const int base = block_base(P.blk_counts, &grand);
    const int j = base + block_exclusive_scan(occ, total);
    if (occ) {
        const int i = KICP_IDX(P.dbg, P.err, P.rb_elem[b], n, 20);
        const double x = P.in[3 * i], y = P.in[3 * i + 1], z = P.in[3 * i + 2];
        P.out[3 * j] = x;
        P.out[3 * j + 1] = y;
        P.out[3 * j + 2] = z;
        P.tab[b].key = kKeyEmpty;
        P.tab[b].minidx = 0x7FFFFFFF;
        if (P.sort_keys) P.sort_keys[j] = tile_key_of(x, y, z, j, P.sort_inv_cell);
        if (P.next_tab) {
            const int vx = voxel_coord(x, P.next_voxel), vy = voxel_coord(y, P.next_voxel),
                      vz = voxel_coord(z, P.next_voxel);
            int s2 = -1;
            if (voxel_in_range(vx, vy, vz)) {
                s2 = ds_claim(P.next_tab, ref_grid_mask(grand), pack_voxel(vx, vy, vz), 1);
                if (s2 >= 0) ds_min_index(P.next_tab, s2, j);
                else atomicOr(P.err, E_TABLE_FULL);
            } else {
                atomicOr(P.err, E_RANGE);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.n_out = grand;
}

// Stage-in of a scan that lies in pinned, device-mapped HOST memory (the zero-copy staging slot): a plain copy into HBM.
// With deskewing the front stages of a frame wait for the previous frame's pose -- they are on the frame's serial chain --
// and k_pre_flags then read the scan over PCIe (30 us of the ~160 us chain, profiles/r03_ac_timeline.txt).  The scan itself
// depends on no pose: this copy runs under the previous registration, and the chain reads HBM.
__global__ __launch_bounds__(256) void k_stage_in(const char *src, char *dst, size_t bytes) {
    const size_t n16 = bytes / 16;
    const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
    uint4 *d16 = reinterpret_cast<uint4 *>(dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d16[i] = s16[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[n16 * 16 + threadIdx.x] = src[n16 * 16 + threadIdx.x];
}
void launch_stage_in(const void *src, void *dst, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    hipLaunchKernelGGL(k_stage_in, dim3(grid_for((long)(bytes / 16 + 1), 256, 512)), dim3(256), 0, s, static_cast<const char *>(src),
                       static_cast<char *>(dst), bytes);
}
void launch_ts_minmax(const double *ts, int n_ts, PrepState *st, hipStream_t s) {
    hipLaunchKernelGGL(k_ts_minmax, dim3(grid_for(n_ts, 256, 512)), dim3(256), 0, s, ts, n_ts, st);
}
void launch_pre_flags(const PreParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_pre_flags, dim3(grid_for(P.n, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_pre_scatter(const PreParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_pre_scatter, dim3(grid_for(P.n, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_ds_claim(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_claim, dim3(grid_for(P.n_max, 256, 2048)), dim3(256), 0, s, P);
}
void launch_ds_flags(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_flags, dim3(grid_for(P.n_max, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_ds_arrange(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_arrange, dim3(grid_for(P.tab_cap, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_ds_scatter_rb(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_scatter_rb, dim3(grid_for(P.tab_cap, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_ds_scatter(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_scatter, dim3(grid_for(P.n_max, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}

}  // namespace kicp
