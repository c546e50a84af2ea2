// kicp_kernels.hip -- hand-written gfx950 kernels of the KISS-ICP registration hot path.
//
//   k_icp              Registration::AlignPointsToMap        core/Registration.cpp:138-167
//                      = TransformPoints (:55-58) + DataAssociation (:60-78) +
//                        VoxelHashMap::GetClosestNeighbor (core/VoxelHashMap.cpp:46-70) +
//                        BuildLinearSystem (:80-121) + LDLT solve / SE3::exp update (:156-163),
//                      the whole <=500-iteration loop in ONE persistent launch.
//   k_closest_neighbor VoxelHashMap::GetClosestNeighbor, batched.
//   k_map_*            VoxelHashMap::AddPoints / RemovePointsFarFromLocation
//                      core/VoxelHashMap.cpp:97-132.
//   k_pre_*, k_ds_*    Preprocessor::Preprocess (core/Preprocessing.cpp:55-95) and
//                      VoxelDownsample (core/VoxelUtils.cpp:7-21), order-preserving compactions.
//
// No MFMA anywhere: the normal equations are a 16-scalar f64 reduction per point
// (~0.4 flop/byte), not a dense contraction.  The work is HBM/L2-latency bound; what matters is
// one aligned 16-byte load per hash probe, contiguous voxel blocks, 32 lanes cooperating on each
// query (27 probe lanes = the 27 neighbour voxels), wave-shuffle + LDS reductions, and no
// host round trip inside the ICP loop.
#include "kicp_launch.hpp"

namespace kicp {

// ------------------------------------------------------------------------------------------
// agent-scope word exchange between workgroups (MI355X: per-XCD L2s are not coherent with each
// other, per-CU L1 is never refreshed by other CUs' stores).  8-byte {tag, value} granules
// written by ONE relaxed agent-scope (sc1, write-through) store and re-read with relaxed
// agent-scope loads until the tag matches: the data is its own flag, no fences.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void granule_store(unsigned long long *g, unsigned tag, unsigned value) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int count_of(const int *n_ptr, int n_imm) { return n_ptr ? *n_ptr : n_imm; }

// ------------------------------------------------------------------------------------------
// voxel hash lookups
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ Slot load_slot(const Slot *p) {
    const int4 v = *reinterpret_cast<const int4 *>(p);
    Slot s;
    s.key = (unsigned long long)(unsigned)v.x | ((unsigned long long)(unsigned)v.y << 32);
    s.block = v.z;
    s.count = v.w;
    return s;
}

// block id (and stored point count) of a voxel, or -1
__device__ __forceinline__ int map_find(const MapView &m, unsigned long long key, int &count) {
    uint32_t s = hash_key(key, m.mask);
    for (uint32_t probes = 0; probes <= m.mask; ++probes) {
        const Slot sl = load_slot(m.slots + s);
        if (sl.key == key) {
            count = sl.count;
            return sl.block;
        }
        if (sl.key == kKeyEmpty) return -1;
        s = (s + 1) & m.mask;
    }
    return -1;
}

// The 27 neighbour shifts in the reference's order (core/VoxelHashMap.cpp:35-41), two bits per
// axis and entry packed into 64-bit immediates so that lane j gets shift j without a table load.
struct ShiftCodes {
    unsigned long long x, y, z;
};
constexpr ShiftCodes make_shift_codes() {
    constexpr int s[27][3] = {
        {0, 0, 0},   {1, 0, 0},   {-1, 0, 0},  {0, 1, 0},   {0, -1, 0},  {0, 0, 1},   {0, 0, -1},
        {1, 1, 0},   {1, -1, 0},  {-1, 1, 0},  {-1, -1, 0}, {1, 0, 1},   {1, 0, -1},  {-1, 0, 1},
        {-1, 0, -1}, {0, 1, 1},   {0, 1, -1},  {0, -1, 1},  {0, -1, -1}, {1, 1, 1},   {1, 1, -1},
        {1, -1, 1},  {1, -1, -1}, {-1, 1, 1},  {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
    ShiftCodes c{0, 0, 0};
    for (int i = 0; i < 27; ++i) {
        c.x |= (unsigned long long)(s[i][0] + 1) << (2 * i);
        c.y |= (unsigned long long)(s[i][1] + 1) << (2 * i);
        c.z |= (unsigned long long)(s[i][2] + 1) << (2 * i);
    }
    return c;
}
constexpr ShiftCodes kShift = make_shift_codes();

// GetClosestNeighbor for one query, cooperatively by a 32-lane group (two groups per wave):
//   1. lane j < 27 probes voxel (v + shift_j): one 16-byte slot load gives block id + point count;
//   2. the hit voxels are visited in shift order, kChunk at a time: for each, lane i < count loads
//      point i (one 16-byte xy load + one 8-byte z load, coalesced over the group); all loads of a
//      chunk are issued before the first distance is computed, so a chunk costs one memory round
//      trip instead of one per point;
//   3. every lane keeps its best (squared distance, shift j, index i); a 5-step xor-shuffle takes
//      the lexicographic minimum = the reference's strict '<' in shift order and, inside a voxel,
//      std::min_element's first minimum.
// Returns the squared distance (DBL_MAX when the neighbourhood is empty), the neighbour, and the
// number of map points examined.
constexpr int kChunk = 9;

//   FILL: additionally stage the candidates, packed in (shift, index) order, into an LDS region
//   {x[cap], y[cap], z[cap]} so later ICP iterations of the same query never leave the CU.
template <bool FILL>
__device__ __forceinline__ double group_closest_neighbor(const MapView &m, double sx, double sy,
                                                         double sz, int lane, double nn[3],
                                                         int &examined, int &range_err,
                                                         double *cand = nullptr, int cap = 0) {
    const int vx = voxel_coord(sx, m.voxel_size);
    const int vy = voxel_coord(sy, m.voxel_size);
    const int vz = voxel_coord(sz, m.voxel_size);
    int blk = -1, cnt = 0;
    if (lane < 27) {
        const int qx = vx + (int)((kShift.x >> (2 * lane)) & 3) - 1;
        const int qy = vy + (int)((kShift.y >> (2 * lane)) & 3) - 1;
        const int qz = vz + (int)((kShift.z >> (2 * lane)) & 3) - 1;
        if (voxel_in_range(qx, qy, qz)) {
            blk = map_find(m, pack_voxel(qx, qy, qz), cnt);
            if (blk < 0) cnt = 0;
        } else {
            range_err = 1;
        }
    }
    // hit mask of this group (the wave holds two groups)
    const unsigned long long ball = __ballot(blk >= 0);
    unsigned hits = (unsigned)(ball >> (threadIdx.x & 32));
    int offs = 0;  // exclusive prefix of the point counts in shift order = candidate base index
    if (FILL) {
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int o = __shfl_up(incl, off, 32);
            if (lane >= off) incl += o;
        }
        offs = incl - cnt;
    }
    double best = DBL_MAX;
    double bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    while (__ballot(hits != 0) != 0ull) {  // wave-uniform trip count
        double2 xy[kChunk];
        double zz[kChunk];
        int jj[kChunk];
        bool ld[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int j = hits ? (__ffs(hits) - 1) : -1;
            hits &= hits - 1;  // (0 & -1) == 0
            const int bj = __shfl(blk, j & 31, 32);
            const int cj = __shfl(cnt, j & 31, 32);
            const int oj = FILL ? __shfl(offs, j & 31, 32) : 0;
            jj[u] = FILL ? oj : j;  // candidate base (FILL) or shift index: both order the voxels
            ld[u] = (j >= 0) && (lane < cj);
            if (ld[u]) {
                xy[u] = block_xy(m, bj)[lane];
                zz[u] = block_z(m, bj)[lane];
            }
        }
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            if (ld[u]) {
                const double dx = xy[u].x - sx, dy = xy[u].y - sy, dz = zz[u] - sz;
                const double d = (dx * dx + dy * dy) + dz * dz;
                if (d < best) {  // shifts arrive in increasing j: strict '<' keeps the earliest
                    best = d;
                    bx = xy[u].x;
                    by = xy[u].y;
                    bz = zz[u];
                    bkey = FILL ? (jj[u] + lane) : ((jj[u] << 12) | lane);
                }
                if (FILL) {
                    const int c = jj[u] + lane;
                    if (c < cap) {
                        cand[c] = xy[u].x;
                        cand[cap + c] = xy[u].y;
                        cand[2 * cap + c] = zz[u];
                    }
                }
            }
        }
    }
    // lexicographic min over (distance, shift, index) inside the 32-lane group
    double gbest = best;
    int gkey = bkey, glane = lane;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(gbest, off, 32);
        const int ok = __shfl_xor(gkey, off, 32);
        const int ol = __shfl_xor(glane, off, 32);
        if (ob < gbest || (ob == gbest && ok < gkey)) {
            gbest = ob;
            gkey = ok;
            glane = ol;
        }
    }
    nn[0] = __shfl(bx, glane, 32);
    nn[1] = __shfl(by, glane, 32);
    nn[2] = __shfl(bz, glane, 32);
    int ex = cnt;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) ex += __shfl_xor(ex, off, 32);
    examined = ex;
    return gbest;
}

// Same search for voxels that hold more than 32 points (max_points_per_voxel > 32): every lane
// strides over the voxel's points.  Rare configuration, kept simple.
__device__ __forceinline__ double group_closest_neighbor_wide(const MapView &m, double sx, double sy,
                                                              double sz, int lane, double nn[3],
                                                              int &examined, int &range_err) {
    const int vx = voxel_coord(sx, m.voxel_size);
    const int vy = voxel_coord(sy, m.voxel_size);
    const int vz = voxel_coord(sz, m.voxel_size);
    double best = DBL_MAX, bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF, cnt = 0;
    if (lane < 27) {
        const int qx = vx + (int)((kShift.x >> (2 * lane)) & 3) - 1;
        const int qy = vy + (int)((kShift.y >> (2 * lane)) & 3) - 1;
        const int qz = vz + (int)((kShift.z >> (2 * lane)) & 3) - 1;
        if (voxel_in_range(qx, qy, qz)) {
            const int b = map_find(m, pack_voxel(qx, qy, qz), cnt);
            if (b >= 0) {
                const double2 *xy = block_xy(m, b);
                const double *z = block_z(m, b);
                for (int k = 0; k < cnt; ++k) {
                    const double dx = xy[k].x - sx, dy = xy[k].y - sy, dz = z[k] - sz;
                    const double d = (dx * dx + dy * dy) + dz * dz;
                    if (d < best) {
                        best = d;
                        bx = xy[k].x;
                        by = xy[k].y;
                        bz = z[k];
                    }
                }
                bkey = lane;
            } else {
                cnt = 0;
            }
        } else {
            range_err = 1;
        }
    }
    double gbest = best;
    int gkey = bkey, glane = lane;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(gbest, off, 32);
        const int ok = __shfl_xor(gkey, off, 32);
        const int ol = __shfl_xor(glane, off, 32);
        if (ob < gbest || (ob == gbest && ok < gkey)) {
            gbest = ob;
            gkey = ok;
            glane = ol;
        }
    }
    nn[0] = __shfl(bx, glane, 32);
    nn[1] = __shfl(by, glane, 32);
    nn[2] = __shfl(bz, glane, 32);
    int ex = cnt;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) ex += __shfl_xor(ex, off, 32);
    examined = ex;
    return gbest;
}

// GetClosestNeighbor over candidates already staged in LDS by a previous iteration (same voxel
// neighbourhood): 32 lanes stride over the packed list; the candidate index is the tie-break key.
__device__ __forceinline__ double group_closest_neighbor_lds(const double *cand, int cap, int E, double sx,
                                                             double sy, double sz, int lane, double nn[3]) {
    double best = DBL_MAX, bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    for (int c = lane; c < E; c += 32) {
        const double x = cand[c], y = cand[cap + c], z = cand[2 * cap + c];
        const double dx = x - sx, dy = y - sy, dz = z - sz;
        const double d = (dx * dx + dy * dy) + dz * dz;
        if (d < best) {
            best = d;
            bx = x;
            by = y;
            bz = z;
            bkey = c;
        }
    }
    double gbest = best;
    int gkey = bkey, glane = lane;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(gbest, off, 32);
        const int ok = __shfl_xor(gkey, off, 32);
        const int ol = __shfl_xor(glane, off, 32);
        if (ob < gbest || (ob == gbest && ok < gkey)) {
            gbest = ob;
            gkey = ok;
            glane = ol;
        }
    }
    nn[0] = __shfl(bx, glane, 32);
    nn[1] = __shfl(by, glane, 32);
    nn[2] = __shfl(bz, glane, 32);
    return gbest;
}

__device__ __forceinline__ double closest_neighbor_any(const MapView &m, double sx, double sy, double sz,
                                                       int lane, double nn[3], int &examined, int &range_err) {
    if (m.max_points <= 32) return group_closest_neighbor<false>(m, sx, sy, sz, lane, nn, examined, range_err);
    return group_closest_neighbor_wide(m, sx, sy, sz, lane, nn, examined, range_err);
}

// ------------------------------------------------------------------------------------------
// k_closest_neighbor: VoxelHashMap::GetClosestNeighbor batched over nq queries
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_closest_neighbor(MapView m, const double *q, int nq,
                                                          double *nn_out, double *dist_out) {
    const int lane = threadIdx.x & 31;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int ngrp = (gridDim.x * blockDim.x) >> 5;
    for (int i = grp; i < nq; i += ngrp) {
        double nn[3];
        int ex, rerr = 0;
        const double d2 = closest_neighbor_any(m, q[3 * i], q[3 * i + 1], q[3 * i + 2], lane, nn, ex, rerr);
        if (lane == 0) {
            const bool found = d2 < DBL_MAX;
            nn_out[3 * i] = found ? nn[0] : 0.0;
            nn_out[3 * i + 1] = found ? nn[1] : 0.0;
            nn_out[3 * i + 2] = found ? nn[2] : 0.0;
            dist_out[i] = found ? sqrt(d2) : DBL_MAX;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_icp: the whole ICP loop of AlignPointsToMap in one persistent launch
//
// grid = G workgroups (all co-resident, G <= 256 = one per CU), 256 threads = 8 groups of 32 lanes.
// Per iteration every group walks its points (fixed assignment, so the running transformed
// source of a point is always re-read by the lane that wrote it):
//   s = est * s            TransformPoints of the previous iteration's estimate (Registration.cpp:159;
//                          est = initial_guess for the first iteration, :147)
//   (nn, d) = closest neighbour among the 27 voxels, keep iff d < max_dist (strict, :72)
//   accumulate the 16 unique scalars of J^T w J and J^T w r with J = [I | -hat(s)], r = s - nn,
//   w = sigma^2 / (sigma + |r|^2)^2 (:81-98).
// Workgroup partials are reduced in LDS in a fixed order, published as tagged granules, gathered
// by EVERY workgroup (one hop, no second broadcast), summed in workgroup order (deterministic),
// and every workgroup solves the same 6x6 system redundantly: dx = LDLT(JTJ).solve(-JTr),
// est = exp(dx), T_icp = est * T_icp, stop when |dx| < convergence_criterion (:156-163).
// ------------------------------------------------------------------------------------------
template <bool PROF>
__global__ __launch_bounds__(kIcpThreads) void k_icp(IcpParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sh_part = reinterpret_cast<double *>(smem);                     // [8][kIcpSums]
    double *sh_tot = sh_part + kIcpGroupsPerBlock * kIcpSums;                // [kIcpSums]
    double *sh_p8 = sh_tot + kIcpSums;                                       // [8][kIcpSums]
    int *sh_failp = reinterpret_cast<int *>(sh_p8 + 8 * kIcpSums);          // [2] (8 bytes)
    unsigned *sh_words = reinterpret_cast<unsigned *>(sh_failp + 2);        // [gridDim.x][2*kIcpSums]
    // (all LDS is carved from the dynamic region: a static __shared__ in front of it would
    // shift its base off 8/16-byte alignment)

    const int tid = threadIdx.x;
    const int lane = tid & 31;
    const int grp = tid >> 5;
    const MapView &m = P.map;
    PipeState *st = P.state;

    const int n = count_of(P.n_ptr, P.n_imm);
    // How many of the launched workgroups take part is decided here, from the actual N_src, so the
    // summation order (hence the result, bit for bit) never depends on host-side hints.
    int G = P.force_blocks > 0 ? P.force_blocks
                               : (n + kIcpGroupsPerBlock * P.points_per_group - 1) / (kIcpGroupsPerBlock * P.points_per_group);
    G = max(1, min(G, (int)gridDim.x));
    if ((int)blockIdx.x >= G) return;
    // per (round, group) candidate regions: meta {s[3] running source point; v[3] voxel; E; valid}
    // + candidates {x[cap], y[cap], z[cap]}.  The exchange scratch is sized by G, the rest of the
    // workgroup's LDS is split over the rounds a group walks per iteration (as many of them as
    // still leave P.cand_target candidates per region).
    const int rounds = (n + G * kIcpGroupsPerBlock - 1) / (G * kIcpGroupsPerBlock);
    char *region_base = smem + icp_fixed_smem(G);
    int cached_rounds = 0, cap_q = 0;
    if (P.cand_target > 0 && m.max_points <= 32) {
        const int budget = P.lds_bytes - (int)icp_fixed_smem(G);
        const int per_region_min = (int)sizeof(IcpRegionMeta) + 24 * P.cand_target;
        cached_rounds = max(0, min(rounds, budget / (kIcpGroupsPerBlock * per_region_min)));
        if (cached_rounds > 0) {
            cap_q = (budget / (kIcpGroupsPerBlock * cached_rounds) - (int)sizeof(IcpRegionMeta)) / 24;
            cap_q = min(cap_q & ~31, 27 * 32);
        }
    }
    const int n_regions = kIcpGroupsPerBlock * cached_rounds;
    const unsigned epoch_base = st->epoch_base;
    unsigned long long t_assoc = 0, t_publish = 0, t_gather = 0, t_solve = 0;
    unsigned gather_passes = 0;

    SE3 guess;
    double max_dist, ks;
    if (P.pipeline_mode) {
        // KissICP.cpp:44-47: sigma = ComputeThreshold(); initial_guess = last_pose * last_delta
        const double sigma = sqrt(st->model_sse / (double)st->num_samples);
        guess = se3_mul(st->last_pose, st->last_delta);
        max_dist = 3.0 * sigma;
        ks = sigma;
    } else {
        guess = st->guess;
        max_dist = P.max_dist;
        ks = P.kernel_scale;
    }
    const bool map_empty = (m.ctr[C_LIVE] == 0);  // Registration.cpp:143

    if (tid == 0) sh_failp[0] = 0;
    __syncthreads();

    SE3 est = guess;
    SE3 T_icp = se3_identity();
    int iterations = 0, converged = 0;
    unsigned long long examined_total = 0, ncorr_total = 0, ncorr_last = 0;
    int range_err = 0;
    bool failed = false;

    const int max_iters = map_empty ? 0 : P.max_iters;
    for (int it = 0; it < max_iters; ++it) {
        const unsigned long long c0 = PROF ? wall_clock64() : 0ull;
        double acc[kIcpSums];
#pragma unroll
        for (int k = 0; k < kIcpSums; ++k) acc[k] = 0.0;
        int round = 0;
        for (int p = blockIdx.x * kIcpGroupsPerBlock + grp; p < n; p += G * kIcpGroupsPerBlock, ++round) {
            const int region = round * kIcpGroupsPerBlock + grp;
            const bool has_region = round < cached_rounds;
            IcpRegionMeta *meta = reinterpret_cast<IcpRegionMeta *>(region_base) + region;
            double *cand = reinterpret_cast<double *>(region_base + (size_t)n_regions * sizeof(IcpRegionMeta)) +
                           (size_t)region * 3 * cap_q;
            double pin[3];
            if (it > 0 && has_region) {  // running source point lives in LDS
                pin[0] = meta->s[0];
                pin[1] = meta->s[1];
                pin[2] = meta->s[2];
            } else {
                const double *src = (it == 0) ? P.frame : P.work;
                pin[0] = src[3 * p];
                pin[1] = src[3 * p + 1];
                pin[2] = src[3 * p + 2];
            }
            double s[3];
            se3_act(est, pin, s);
            const int vx = voxel_coord(s[0], m.voxel_size), vy = voxel_coord(s[1], m.voxel_size),
                      vz = voxel_coord(s[2], m.voxel_size);
            bool cached = false;
            int E = 0;
            if (has_region && it > 0) {
                cached = meta->valid && meta->v[0] == vx && meta->v[1] == vy && meta->v[2] == vz;
                E = meta->E;
            }
            if (lane == 0) {
                if (has_region) {
                    meta->s[0] = s[0];
                    meta->s[1] = s[1];
                    meta->s[2] = s[2];
                } else {
                    P.work[3 * p] = s[0];
                    P.work[3 * p + 1] = s[1];
                    P.work[3 * p + 2] = s[2];
                }
            }
            double nn[3];
            int ex;
            double d2;
            if (cached) {
                d2 = group_closest_neighbor_lds(cand, cap_q, E, s[0], s[1], s[2], lane, nn);
                ex = E;
            } else if (has_region) {
                d2 = group_closest_neighbor<true>(m, s[0], s[1], s[2], lane, nn, ex, range_err, cand, cap_q);
                if (lane == 0) {
                    meta->v[0] = vx;
                    meta->v[1] = vy;
                    meta->v[2] = vz;
                    meta->E = ex;
                    meta->valid = (ex <= cap_q);
                }
            } else {
                d2 = closest_neighbor_any(m, s[0], s[1], s[2], lane, nn, ex, range_err);
            }
            if (lane == 0) {
                acc[17] += (double)ex;
                if (d2 < DBL_MAX && sqrt(d2) < max_dist) {
                    const double rx = s[0] - nn[0], ry = s[1] - nn[1], rz = s[2] - nn[2];
                    const double r2 = (rx * rx + ry * ry) + rz * rz;
                    const double w = (ks * ks) / ((ks + r2) * (ks + r2));
                    acc[0] += w;
                    acc[1] += w * s[0];
                    acc[2] += w * s[1];
                    acc[3] += w * s[2];
                    // w * hat(s)^T hat(s) = w * (|s|^2 I - s s^T), upper triangle
                    acc[4] += w * (s[1] * s[1] + s[2] * s[2]);
                    acc[5] += w * (-(s[0] * s[1]));
                    acc[6] += w * (-(s[0] * s[2]));
                    acc[7] += w * (s[0] * s[0] + s[2] * s[2]);
                    acc[8] += w * (-(s[1] * s[2]));
                    acc[9] += w * (s[0] * s[0] + s[1] * s[1]);
                    acc[10] += w * rx;
                    acc[11] += w * ry;
                    acc[12] += w * rz;
                    // w * (s x r)
                    acc[13] += w * (s[1] * rz - s[2] * ry);
                    acc[14] += w * (s[2] * rx - s[0] * rz);
                    acc[15] += w * (s[0] * ry - s[1] * rx);
                    acc[16] += 1.0;
                }
            }
        }
        // ---- workgroup reduction (fixed order) ----------------------------------------------
        const unsigned long long c1 = PROF ? wall_clock64() : 0ull;
        acc[kIcpTickSlot] = (double)(c1 - c0);  // this group's association time (profiling, max-reduced)
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < kIcpSums; ++k) sh_part[grp * kIcpSums + k] = acc[k];
        }
        __syncthreads();
        const unsigned epoch = epoch_base + (unsigned)it + 1u;
        unsigned long long *gran = P.granules + (size_t)(it & 1) * G * (2 * kIcpSums);
        if (tid < 2 * kIcpSums) {
            const int k = tid >> 1;
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < kIcpGroupsPerBlock; ++g) {
                const double pv = sh_part[g * kIcpSums + k];
                v = (k == kIcpTickSlot) ? fmax(v, pv) : v + pv;
            }
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            granule_store(gran + (size_t)blockIdx.x * (2 * kIcpSums) + tid, epoch, half);
        }
        // ---- gather every workgroup's partial (bounded spin) --------------------------------
        const unsigned long long c2 = PROF ? wall_clock64() : 0ull;
        {
            // every thread fetches its own words, kGatherChunk loads in flight, and re-polls only
            // the ones whose tag has not arrived yet (bounded)
            constexpr int kGatherChunk = 8;
            const int nwords = G * 2 * kIcpSums;
            bool fail = false;
            for (int w0 = tid; w0 < nwords && !fail; w0 += kGatherChunk * kIcpThreads) {
                unsigned long long x[kGatherChunk];
#pragma unroll
                for (int u = 0; u < kGatherChunk; ++u) {
                    const int w = w0 + u * kIcpThreads;
                    x[u] = (w < nwords) ? granule_load(gran + w) : ((unsigned long long)epoch << 32);
                }
#pragma unroll
                for (int u = 0; u < kGatherChunk; ++u) {
                    const int w = w0 + u * kIcpThreads;
                    unsigned spins = 0;
                    while ((unsigned)(x[u] >> 32) != epoch) {
                        if (PROF) ++gather_passes;
                        if (++spins > P.spin_limit ||
                            ((spins & 255u) == 0 &&
                             (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & E_TIMEOUT))) {
                            fail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        x[u] = granule_load(gran + w);
                    }
                    if (w < nwords) sh_words[w] = (unsigned)x[u];
                }
            }
            if (fail) sh_failp[0] = 1;
        }
        __syncthreads();
        if (sh_failp[0]) {
            failed = true;
            break;
        }
        // sum over workgroups: 8 contiguous chunks per scalar, then the 8 chunk sums, both in order
        if (tid < 8 * kIcpSums) {
            const int k = tid % kIcpSums, part = tid / kIcpSums;
            const int b0 = (G * part) / 8, b1 = (G * (part + 1)) / 8;
            double v = 0.0;
            for (int b = b0; b < b1; ++b) {
                const unsigned lo = sh_words[(b * kIcpSums + k) * 2];
                const unsigned hi = sh_words[(b * kIcpSums + k) * 2 + 1];
                const double pv = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                v = (k == kIcpTickSlot) ? fmax(v, pv) : v + pv;
            }
            sh_p8[part * kIcpSums + k] = v;
        }
        __syncthreads();
        if (tid < kIcpSums) {
            double v = 0.0;
#pragma unroll
            for (int part = 0; part < 8; ++part) {
                const double pv = sh_p8[part * kIcpSums + tid];
                v = (tid == kIcpTickSlot) ? fmax(v, pv) : v + pv;
            }
            sh_tot[tid] = v;
        }
        __syncthreads();
        // ---- every thread solves the same system (uniform, no broadcast needed) ------------
        const unsigned long long c3 = PROF ? wall_clock64() : 0ull;
        double S[kIcpSums];
#pragma unroll
        for (int k = 0; k < kIcpSums; ++k) S[k] = sh_tot[k];
        double JTJ[36], nb[6], dx[6];
#pragma unroll
        for (int i = 0; i < 36; ++i) JTJ[i] = 0.0;
        JTJ[0] = JTJ[7] = JTJ[14] = S[0];
        // top-right block sum w * (-hat(s)) and its transpose
        JTJ[0 * 6 + 4] = S[3];
        JTJ[0 * 6 + 5] = -S[2];
        JTJ[1 * 6 + 3] = -S[3];
        JTJ[1 * 6 + 5] = S[1];
        JTJ[2 * 6 + 3] = S[2];
        JTJ[2 * 6 + 4] = -S[1];
        JTJ[4 * 6 + 0] = S[3];
        JTJ[5 * 6 + 0] = -S[2];
        JTJ[3 * 6 + 1] = -S[3];
        JTJ[5 * 6 + 1] = S[1];
        JTJ[3 * 6 + 2] = S[2];
        JTJ[4 * 6 + 2] = -S[1];
        JTJ[3 * 6 + 3] = S[4];
        JTJ[3 * 6 + 4] = JTJ[4 * 6 + 3] = S[5];
        JTJ[3 * 6 + 5] = JTJ[5 * 6 + 3] = S[6];
        JTJ[4 * 6 + 4] = S[7];
        JTJ[4 * 6 + 5] = JTJ[5 * 6 + 4] = S[8];
        JTJ[5 * 6 + 5] = S[9];
#pragma unroll
        for (int i = 0; i < 6; ++i) nb[i] = -S[10 + i];
        ldlt6_solve(JTJ, nb, dx);
        est = se3_exp(dx);
        T_icp = se3_mul(est, T_icp);
        iterations = it + 1;
        ncorr_last = (unsigned long long)S[16];
        ncorr_total += ncorr_last;
        examined_total += (unsigned long long)S[17];
        double nrm = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) nrm += dx[i] * dx[i];
        const unsigned long long c4 = PROF ? wall_clock64() : 0ull;
        t_assoc += c1 - c0;
        t_publish += c2 - c1;
        t_gather += c3 - c2;
        t_solve += c4 - c3;
        if (PROF && blockIdx.x == 0 && tid == 0 && it < kIcpProfIters) {
            unsigned *r = st->prof_iter[it];
            r[0] = (unsigned)(c1 - c0);
            r[1] = (unsigned)(c2 - c1);
            r[2] = (unsigned)(c3 - c2);
            r[3] = (unsigned)(c4 - c3);
            r[4] = (unsigned)S[kIcpTickSlot];  // slowest group's association time, any workgroup
            r[5] = gather_passes;
        }
        gather_passes = 0;
        if (sqrt(nrm) < P.conv) {
            converged = 1;
            break;
        }
    }

    if (range_err) atomicOr(&st->err, E_RANGE);
    if (failed && tid == 0) atomicOr(&st->err, E_TIMEOUT);

    if (blockIdx.x == 0 && tid == 0) {
        const SE3 new_pose = se3_mul(T_icp, guess);  // Registration.cpp:166
        st->new_pose = new_pose;
        st->guess = guess;
        st->icp_iterations = iterations;
        st->icp_converged = converged;
        st->icp_examined = examined_total;
        st->icp_ncorr_last = ncorr_last;
        st->icp_ncorr_total = ncorr_total;
        st->n_src = n;
        st->icp_blocks_used = G;
        st->prof[0] = t_assoc;
        st->prof[1] = t_publish;
        st->prof[2] = t_gather;
        st->prof[3] = t_solve;
        st->epoch_base = epoch_base + (unsigned)P.max_iters + 2u;
        if (P.pipeline_mode) {
            st->sigma = ks;
            // KissICP.cpp:57-63 + Threshold.cpp:38-49
            const SE3 dev = se3_mul(se3_inverse(guess), new_pose);
            const double theta = rotation_angle(dev.q);
            const double delta_rot = 2.0 * m.max_distance * sin(theta / 2.0);
            const double delta_trans = sqrt(sqnorm3(dev.t[0], dev.t[1], dev.t[2]));
            const double model_error = delta_trans + delta_rot;
            if (model_error > P.min_motion_th) {
                st->model_sse += model_error * model_error;
                st->num_samples += 1;
            }
            st->last_delta = se3_mul(se3_inverse(st->last_pose), new_pose);
            st->last_pose = new_pose;
        }
    }
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
__global__ __launch_bounds__(256) void k_ts_minmax(const double *ts, int n_ts, PipeState *st) {
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
        double p[3] = {P.xyz[3 * i], P.xyz[3 * i + 1], P.xyz[3 * i + 2]};
        if (P.deskew) {
            const double mn = f64_from_order_bits(P.state->tmin_bits);
            const double mx = f64_from_order_bits(P.state->tmax_bits);
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

// find-or-claim the slot of a voxel in the downsample scratch table
__device__ __forceinline__ int ds_claim(DsSlot *tab, uint32_t mask, unsigned long long key) {
    // Read before CAS: ~15 scan points share a voxel, so most arrivals find their key already
    // there and never touch the atomic unit.  Keys are stable for the lifetime of a claim phase,
    // so a (possibly L1-stale) plain read can only cost an extra CAS, never a wrong answer.
    uint32_t s = hash_key(key, mask);
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

// scatter the range-cropped cloud (order preserving) and, fused, stage A of the first
// VoxelDownsample: claim the voxel and atomicMin the (new) point index into it
__global__ __launch_bounds__(kScanThreads) void k_pre_scatter(PreParams P) {
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
        if (P.ds_tab) {
            const int vx = voxel_coord(p[0], P.ds_voxel), vy = voxel_coord(p[1], P.ds_voxel),
                      vz = voxel_coord(p[2], P.ds_voxel);
            int s = -1;
            if (voxel_in_range(vx, vy, vz)) {
                s = ds_claim(P.ds_tab, P.ds_mask, pack_voxel(vx, vy, vz));
                if (s >= 0) ds_min_index(P.ds_tab, s, j);
                else atomicOr(P.err, E_TABLE_FULL);
            } else {
                atomicOr(P.err, E_RANGE);
            }
            P.ds_slot_of[j] = s;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.n_out = grand;
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
            s = ds_claim(P.tab, P.mask, pack_voxel(vx, vy, vz));
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

// ------------------------------------------------------------------------------------------
// VoxelHashMap::AddPoints (VoxelHashMap.cpp:97-119), made deterministic on the device:
//   k_map_link   every new point finds or claims its voxel's slot (CAS on the packed key) and
//                pushes itself on the slot's per-frame list;
//   k_map_apply  the point that opened a list walks it in ascending point index and applies the
//                reference's sequential acceptance rule (voxel full? closer than map_resolution
//                to a stored point? else append) -- the same result as the serial loop.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_link(MapView m, const double *in, const int *n_ptr,
                                                  int n_imm, const PipeState *state, int use_pose,
                                                  double *world, int *slot_of, int *next) {
    const int n = count_of(n_ptr, n_imm);
    SE3 pose;
    if (use_pose) pose = state->new_pose;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double p[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]};
        if (use_pose) {  // VoxelHashMap.cpp:90-92
            double o[3];
            se3_act(pose, p, o);
            p[0] = o[0];
            p[1] = o[1];
            p[2] = o[2];
        }
        world[3 * i] = p[0];
        world[3 * i + 1] = p[1];
        world[3 * i + 2] = p[2];
        const int vx = voxel_coord(p[0], m.voxel_size), vy = voxel_coord(p[1], m.voxel_size),
                  vz = voxel_coord(p[2], m.voxel_size);
        int slot = -1;
        if (voxel_in_range(vx, vy, vz)) {
            const unsigned long long key = pack_voxel(vx, vy, vz);
            uint32_t s = hash_key(key, m.mask);
            for (uint32_t probes = 0; probes <= m.mask; ++probes) {
                unsigned long long old = m.slots[s].key;  // read before CAS (keys are stable here)
                if (old == key) {
                    slot = (int)s;
                    break;
                }
                if (old != kKeyEmpty) {
                    s = (s + 1) & m.mask;  // other key or tombstone
                    continue;
                }
                old = atomicCAS(&m.slots[s].key, kKeyEmpty, key);
                if (old == kKeyEmpty) {
                    atomicAdd(&m.ctr[C_USED], 1);
                    slot = (int)s;
                    break;
                }
                if (old == key) {
                    slot = (int)s;
                    break;
                }
                s = (s + 1) & m.mask;  // other key or tombstone
            }
            if (slot < 0) atomicOr(&m.ctr[C_ERR], E_TABLE_FULL);
        } else {
            atomicOr(&m.ctr[C_ERR], E_RANGE);
        }
        slot_of[i] = slot;
        next[i] = (slot >= 0) ? atomicExch(&m.heads[slot], i) : -2;
    }
}

__device__ __forceinline__ int pool_alloc(const MapView &m) {
    int nf = m.ctr[C_NFREE];
    while (nf > 0) {
        const int seen = atomicCAS(&m.ctr[C_NFREE], nf, nf - 1);
        if (seen == nf) return m.free_ids[nf - 1];
        nf = seen;
    }
    const int b = atomicAdd(&m.ctr[C_BUMP], 1);
    return (b < m.blocks_cap) ? b : -1;
}

// serial application of one voxel's list by a single lane (voxels that hold or receive more
// points than a 32-lane group can keep in registers)
__device__ void map_apply_voxel_serial(const MapView &m, int slot, int head, int b, const double *world,
                                       const int *next) {
    Slot *sl = m.slots + slot;
    BlockHdr *hdr = block_hdr(m, b);
    double2 *pxy = block_xy(m, b);
    double *pz = block_z(m, b);
    int cnt = hdr->count;
    int last = -1;
    while (cnt < m.max_points) {  // :104 a full voxel rejects the rest
        int cur = 0x7FFFFFFF;    // next list entry in ascending point index (= arrival order)
        for (int j = head; j >= 0; j = next[j])
            if (j > last && j < cur) cur = j;
        if (cur == 0x7FFFFFFF) break;
        last = cur;
        const double px = world[3 * cur], py = world[3 * cur + 1], pz_new = world[3 * cur + 2];
        bool too_close = false;
        for (int k = 0; k < cnt; ++k) {  // :105-108 (norm < map_resolution, strict)
            const double dx = pxy[k].x - px, dy = pxy[k].y - py, dz = pz[k] - pz_new;
            if (sqrt((dx * dx + dy * dy) + dz * dz) < m.map_resolution) {
                too_close = true;
                break;
            }
        }
        if (!too_close) {
            pxy[cnt] = make_double2(px, py);
            pz[cnt] = pz_new;
            ++cnt;
        }
    }
    hdr->count = cnt;
    sl->count = cnt;
}

// k_map_apply: a 32-lane group owns a tile of kApplyTile consecutive new points and serves the
// voxel lists opened by them (the opener of a list is its tail: next == -1).  Per voxel: lane k
// holds stored point k in registers, the new points of the list are ranked by point index
// (= the reference's arrival order) and offered one after the other; a point is appended (to lane
// `count`) iff the voxel is not full and no stored point -- including the ones appended a moment
// ago -- is closer than map_resolution (VoxelHashMap.cpp:103-110).  One memory round trip for the
// stored points, one for the new points, instead of a dependent load per comparison.
constexpr int kApplyTile = 8;
__global__ __launch_bounds__(256) void k_map_apply(MapView m, const int *n_ptr, int n_imm,
                                                   const double *world, const int *slot_of,
                                                   const int *next) {
    const int n = count_of(n_ptr, n_imm);
    const int lane = threadIdx.x & 31;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int ngrp = (gridDim.x * blockDim.x) >> 5;
    const int ntiles = (n + kApplyTile - 1) / kApplyTile;
    const int half_shift = threadIdx.x & 32;  // this group's half of the 64-bit wave ballot
    // wave-uniform trip count: the two groups of a wave walk tiles grp and grp + 1 in lock step
    for (int tile = grp; (tile & ~1) < ntiles; tile += ngrp) {
        const int i = tile * kApplyTile + lane;
        int my_slot = -1;
        if (lane < kApplyTile && i < n && next[i] == -1) my_slot = slot_of[i];
        unsigned leaders = (unsigned)(__ballot(my_slot >= 0) >> half_shift);
        while (__ballot(leaders != 0) != 0ull) {
            const bool active = leaders != 0;
            const int l = active ? (__ffs(leaders) - 1) : 0;
            leaders &= leaders - 1;
            const int slot = __shfl(my_slot, l, 32);
            if (!active) continue;  // (the other group of the wave still has voxels to serve)
            Slot *sl = m.slots + slot;
            const int head = m.heads[slot];
            int b = sl->block;
            int cnt = 0;
            if (b < 0) {  // new voxel (VoxelHashMap.cpp:112-116)
                if (lane == 0) {
                    b = pool_alloc(m);
                    if (b >= 0) {
                        BlockHdr *hdr = block_hdr(m, b);
                        hdr->key = sl->key;
                        hdr->slot = slot;
                        hdr->count = 0;
                        sl->block = b;
                        atomicAdd(&m.ctr[C_LIVE], 1);
                    } else {
                        atomicOr(&m.ctr[C_ERR], E_POOL_FULL);
                    }
                }
                b = __shfl(b, 0, 32);
            } else {
                cnt = sl->count;
            }
            if (lane == 0) m.heads[slot] = -1;
            if (b < 0) continue;
            // the list, one entry per lane (walked by every lane: uniform loads)
            int my_idx = 0x7FFFFFFF, L = 0, walk = head;
            for (; walk >= 0 && L < 32; walk = next[walk], ++L)
                if (lane == L) my_idx = walk;
            if (walk >= 0 || m.max_points > 32) {  // long list or wide voxel: serial fallback
                if (lane == 0) map_apply_voxel_serial(m, slot, head, b, world, next);
                continue;
            }
            double2 *pxy = block_xy(m, b);
            double *pz = block_z(m, b);
            double ex = 0.0, ey = 0.0, ez = 0.0;  // stored point `lane`
            if (lane < cnt) {
                const double2 xy = pxy[lane];
                ex = xy.x;
                ey = xy.y;
                ez = pz[lane];
            }
            double nx = 0.0, ny = 0.0, nz = 0.0;  // new point held by this lane
            if (lane < L) {
                nx = world[3 * my_idx];
                ny = world[3 * my_idx + 1];
                nz = world[3 * my_idx + 2];
            }
            int rank = 0;  // position of this lane's point in ascending point index
            for (int j = 0; j < L; ++j) rank += (__shfl(my_idx, j, 32) < my_idx) ? 1 : 0;
            const int cnt0 = cnt;
            for (int r = 0; r < L && cnt < m.max_points; ++r) {  // :104 a full voxel rejects the rest
                const unsigned who = (unsigned)(__ballot(lane < L && rank == r) >> half_shift);
                const int src = __ffs(who) - 1;
                const double qx = __shfl(nx, src, 32), qy = __shfl(ny, src, 32), qz = __shfl(nz, src, 32);
                bool close = false;
                if (lane < cnt) {  // :105-108 (norm < map_resolution, strict)
                    const double dx = ex - qx, dy = ey - qy, dz = ez - qz;
                    close = sqrt((dx * dx + dy * dy) + dz * dz) < m.map_resolution;
                }
                if (((unsigned)(__ballot(close) >> half_shift)) == 0u) {
                    if (lane == cnt) {
                        ex = qx;
                        ey = qy;
                        ez = qz;
                    }
                    ++cnt;
                }
            }
            if (lane >= cnt0 && lane < cnt) {
                pxy[lane] = make_double2(ex, ey);
                pz[lane] = ez;
            }
            if (lane == 0) {
                block_hdr(m, b)->count = cnt;
                sl->count = cnt;
            }
        }
    }
}

// VoxelHashMap::RemovePointsFarFromLocation (VoxelHashMap.cpp:121-132): a voxel dies iff its
// FIRST point is >= max_distance from the origin.  Tombstone the slot, recycle the block.
__global__ __launch_bounds__(256) void k_map_prune(MapView m, const PipeState *state,
                                                   int use_state_origin, double ox, double oy,
                                                   double oz, PipeState *reset_state) {
    if (use_state_origin) {
        ox = state->new_pose.t[0];
        oy = state->new_pose.t[1];
        oz = state->new_pose.t[2];
    }
    const double md2 = m.max_distance * m.max_distance;
    const int nb = min(m.ctr[C_BUMP], m.blocks_cap);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        BlockHdr *hdr = block_hdr(m, b);
        if (hdr->count <= 0) continue;
        const double2 p0 = block_xy(m, b)[0];
        const double dx = p0.x - ox, dy = p0.y - oy, dz = block_z(m, b)[0] - oz;
        if ((dx * dx + dy * dy) + dz * dz >= md2) {
            Slot *sl = m.slots + hdr->slot;
            sl->key = kKeyTomb;
            sl->block = -1;
            sl->count = 0;
            hdr->count = 0;
            const int k = atomicAdd(&m.ctr[C_NFREE], 1);
            m.free_ids[k] = b;
            atomicSub(&m.ctr[C_LIVE], 1);
            atomicAdd(&m.ctr[C_TOMB], 1);
        }
    }
    if (reset_state && blockIdx.x == 0 && threadIdx.x == 0) {  // re-arm the per-frame words
        reset_state->tmin_bits = ~0ull;
        reset_state->tmax_bits = 0ull;
    }
}

// rebuild the slot array from the live blocks (after growth, or to drop tombstones)
__global__ __launch_bounds__(256) void k_map_rehash(MapView m) {
    const int nb = min(m.ctr[C_BUMP], m.blocks_cap);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        BlockHdr *hdr = block_hdr(m, b);
        if (hdr->count <= 0) continue;
        const unsigned long long key = hdr->key;
        uint32_t s = hash_key(key, m.mask);
        for (;;) {
            const unsigned long long old = atomicCAS(&m.slots[s].key, kKeyEmpty, key);
            if (old == kKeyEmpty) break;
            s = (s + 1) & m.mask;
        }
        m.slots[s].block = b;
        m.slots[s].count = hdr->count;
        hdr->slot = (int)s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        m.ctr[C_TOMB] = 0;
        m.ctr[C_USED] = m.ctr[C_LIVE];
    }
}

// total number of stored points (for Pointcloud sizing)
__global__ __launch_bounds__(256) void k_map_count_points(MapView m) {
    const int nb = min(m.ctr[C_BUMP], m.blocks_cap);
    int s = 0;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x)
        s += max(block_hdr(m, b)->count, 0);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&m.ctr[C_NPTS], s);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline int grid_for(long n, int threads, int cap) {
    long g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

size_t icp_granule_words(int G) { return (size_t)2 * G * 2 * kIcpSums; }

constexpr int kIcpLdsBytes = 160 * 1024;  // one workgroup per CU owns the whole LDS

int icp_prepare() {
    // opt in to the full 160 KiB of LDS (dynamic regions above 64 KiB need the attribute)
    static bool done = false;
    if (done) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_icp<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kIcpLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_icp<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kIcpLdsBytes);
    if (e != hipSuccess) return (int)e;
    done = true;
    return 0;
}
void launch_icp(IcpParams P, int G, bool profile, hipStream_t s) {
    P.lds_bytes = kIcpLdsBytes;
    if (profile)
        hipLaunchKernelGGL(k_icp<true>, dim3(G), dim3(kIcpThreads), kIcpLdsBytes, s, P);
    else
        hipLaunchKernelGGL(k_icp<false>, dim3(G), dim3(kIcpThreads), kIcpLdsBytes, s, P);
}
void launch_closest_neighbor(const MapView &m, const double *q, int nq, double *nn, double *dist,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_closest_neighbor, dim3(grid_for((long)nq * 32, 256, 2048)), dim3(256), 0, s, m, q,
                       nq, nn, dist);
}
void launch_ts_minmax(const double *ts, int n_ts, PipeState *st, hipStream_t s) {
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
void launch_ds_scatter(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_scatter, dim3(grid_for(P.n_max, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_map_link(const MapView &m, const double *in, const int *n_ptr, int n_imm, int n_max,
                     const PipeState *state, int use_pose, double *world, int *slot_of, int *next,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_map_link, dim3(grid_for(n_max, 256, 2048)), dim3(256), 0, s, m, in, n_ptr, n_imm,
                       state, use_pose, world, slot_of, next);
}
void launch_map_apply(const MapView &m, const int *n_ptr, int n_imm, int n_max, const double *world,
                      const int *slot_of, const int *next, hipStream_t s) {
    // one 32-lane group per tile of kApplyTile points
    hipLaunchKernelGGL(k_map_apply, dim3(grid_for(((long)n_max + kApplyTile - 1) / kApplyTile * 32, 256, 2048)), dim3(256), 0,
                       s, m, n_ptr, n_imm, world, slot_of, next);
}
void launch_map_prune(const MapView &m, long bump_ub, const PipeState *state, int use_state_origin,
                      const double origin[3], PipeState *reset_state, hipStream_t s) {
    hipLaunchKernelGGL(k_map_prune, dim3(grid_for(bump_ub, 256, 2048)), dim3(256), 0, s, m, state,
                       use_state_origin, origin ? origin[0] : 0.0, origin ? origin[1] : 0.0,
                       origin ? origin[2] : 0.0, reset_state);
}
void launch_map_rehash(const MapView &m, long bump_ub, hipStream_t s) {
    hipLaunchKernelGGL(k_map_rehash, dim3(grid_for(bump_ub, 256, 2048)), dim3(256), 0, s, m);
}
void launch_map_count_points(const MapView &m, long bump_ub, hipStream_t s) {
    hipLaunchKernelGGL(k_map_count_points, dim3(grid_for(bump_ub, 256, 1024)), dim3(256), 0, s, m);
}

}  // namespace kicp
