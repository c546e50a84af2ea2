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

// Inverse of the table above: position of the shift (ox, oy, oz) in [-1, 1]^3 in the reference's
// order, indexed by code = (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1); 5 bits per entry, 12 per word.
struct ShiftOrder {
    unsigned long long w[3];
};
constexpr ShiftOrder make_shift_order() {
    ShiftOrder o{{0, 0, 0}};
    for (int i = 0; i < 27; ++i) {
        const int ox = (int)((kShift.x >> (2 * i)) & 3) - 1, oy = (int)((kShift.y >> (2 * i)) & 3) - 1,
                  oz = (int)((kShift.z >> (2 * i)) & 3) - 1;
        const int code = (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1);
        o.w[code / 12] |= (unsigned long long)i << (5 * (code % 12));
    }
    return o;
}
constexpr ShiftOrder kOrder = make_shift_order();
__device__ __forceinline__ int shift_order(int code) {
    const unsigned long long w = code < 12 ? kOrder.w[0] : (code < 24 ? kOrder.w[1] : kOrder.w[2]);
    const int k = code < 12 ? code : (code < 24 ? code - 12 : code - 24);
    return (int)((w >> (5 * k)) & 31);
}

// GetClosestNeighbor for one query, cooperatively by a 32-lane group (two groups per wave):
//   1. probe27: lane j < 27 probes voxel (v + shift_j): one 16-byte slot load gives block id + point
//      count; the exclusive prefix of the counts in shift order numbers the candidates;
//   2. scan_hits: the hit voxels are visited in shift order, kChunk at a time: for each, lane
//      i < count loads point i (one 16-byte xy load + one 8-byte z load, coalesced over the group);
//      all loads of a chunk are issued before the first distance is computed, so a chunk costs one
//      memory round trip instead of one per point;
//   3. every lane keeps its best (squared distance, candidate number); a 5-step xor-shuffle takes
//      the lexicographic minimum = the reference's strict '<' in shift order and, inside a voxel,
//      std::min_element's first minimum.
constexpr int kChunk = 6;

struct Probe {
    int blk;   // block id of this lane's voxel or -1
    int cnt;   // points stored in it
    int offs;  // candidates in front of it (shift order)
    int E;     // candidates in the whole neighbourhood (uniform over the group)
};

__device__ __forceinline__ Probe probe27(const MapView &m, double sx, double sy, double sz, int lane,
                                         int &range_err) {
    const int vx = voxel_coord(sx, m.voxel_size);
    const int vy = voxel_coord(sy, m.voxel_size);
    const int vz = voxel_coord(sz, m.voxel_size);
    Probe pr;
    pr.blk = -1;
    pr.cnt = 0;
    if (lane < 27) {
        const int qx = vx + (int)((kShift.x >> (2 * lane)) & 3) - 1;
        const int qy = vy + (int)((kShift.y >> (2 * lane)) & 3) - 1;
        const int qz = vz + (int)((kShift.z >> (2 * lane)) & 3) - 1;
        if (voxel_in_range(qx, qy, qz)) {
            pr.blk = map_find(m, pack_voxel(qx, qy, qz), pr.cnt);
            if (pr.blk < 0) pr.cnt = 0;
        } else {
            range_err = 1;
        }
    }
    int incl = pr.cnt;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int o = __shfl_up(incl, off, 32);
        if (lane >= off) incl += o;
    }
    pr.offs = incl - pr.cnt;
    pr.E = __shfl(incl, 31, 32);
    return pr;
}

// ------------------------------------------------------------------------------------------
// lane exchanges inside a 32-lane group without an LDS round trip: DPP row operations for the
// 2-, 4-, 8- and 16-lane steps (quad_perm / row_half_mirror / row_mirror: ~2 issue cycles instead
// of a ds_bpermute's ~100-cycle trip), one ds_swizzle SWAP16 for the last step.  The partner
// pattern is not an xor butterfly for the mirror steps, but every step still merges two disjoint
// lane sets whose members already agree, which is all an all-reduce needs.
// ------------------------------------------------------------------------------------------
template <int STEP>
__device__ __forceinline__ int group_xchg(int v) {
    if (STEP == 4) return __builtin_amdgcn_ds_swizzle(v, 0x401F);  // swizzle(SWAP, 16)
    constexpr int kCtrl = STEP == 0 ? 0xB1 /* quad_perm [1,0,3,2] */
                          : STEP == 1 ? 0x4E /* quad_perm [2,3,0,1] */
                          : STEP == 2 ? 0x141 /* row_half_mirror */ : 0x140 /* row_mirror */;
    return __builtin_amdgcn_update_dpp(v, v, kCtrl, 0xF, 0xF, false);
}
template <int STEP>
__device__ __forceinline__ double group_xchg(double v) {
    return __hiloint2double(group_xchg<STEP>(__double2hiint(v)), group_xchg<STEP>(__double2loint(v)));
}
// lexicographic minimum of (distance, key) over the 32 lanes of a group, carrying a payload;
// every lane ends with the winner
template <int STEP>
__device__ __forceinline__ void group_min_step(double &best, int &key, int &payload) {
    const double ob = group_xchg<STEP>(best);
    const int ok = group_xchg<STEP>(key);
    const int op = group_xchg<STEP>(payload);
    if (ob < best || (ob == best && ok < key)) {
        best = ob;
        key = ok;
        payload = op;
    }
}
__device__ __forceinline__ void group_min(double &best, int &key, int &payload) {
    group_min_step<0>(best, key, payload);
    group_min_step<1>(best, key, payload);
    group_min_step<2>(best, key, payload);
    group_min_step<3>(best, key, payload);
    group_min_step<4>(best, key, payload);
}

// lexicographic min over (distance, candidate number) inside the 32-lane group; returns the
// squared distance and the winner's coordinates in nn
__device__ __forceinline__ double group_argmin(double best, int bkey, double bx, double by, double bz, int lane,
                                               double nn[3]) {
    double gbest = best;
    int gkey = bkey, glane = lane;
    group_min(gbest, gkey, glane);
    nn[0] = __shfl(bx, glane, 32);
    nn[1] = __shfl(by, glane, 32);
    nn[2] = __shfl(bz, glane, 32);
    return gbest;
}

//   FILL: additionally stage the candidates, packed in (shift, index) order, into an LDS region
//   {x[stride], y[stride], z[stride]} so later ICP iterations of the same query never leave the CU.
// Returns the squared distance (DBL_MAX when the neighbourhood is empty) and the neighbour.
template <bool FILL>
__device__ __forceinline__ double scan_hits(const MapView &m, const Probe &pr, double sx, double sy, double sz,
                                            int lane, double nn[3], double *cand = nullptr, int stride = 0) {
    // hit mask of this group (the wave holds two groups)
    const unsigned long long ball = __ballot(pr.blk >= 0);
    unsigned hits = (unsigned)(ball >> (threadIdx.x & 32));
    double best = DBL_MAX;
    double bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    while (__ballot(hits != 0) != 0ull) {  // wave-uniform trip count
        double2 xy[kChunk];
        double zz[kChunk];
        int cb[kChunk];
        bool ld[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int j = hits ? (__ffs(hits) - 1) : -1;
            hits &= hits - 1;  // (0 & -1) == 0
            const int bj = __shfl(pr.blk, j & 31, 32);
            const int cj = __shfl(pr.cnt, j & 31, 32);
            cb[u] = __shfl(pr.offs, j & 31, 32);  // candidate number of the voxel's first point
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
                const int c = cb[u] + lane;
                if (d < best) {  // voxels arrive in shift order: strict '<' keeps the earliest
                    best = d;
                    bx = xy[u].x;
                    by = xy[u].y;
                    bz = zz[u];
                    bkey = c;
                }
                if (FILL) {
                    cand[c] = xy[u].x;
                    cand[stride + c] = xy[u].y;
                    cand[2 * stride + c] = zz[u];
                }
            }
        }
    }
    return group_argmin(best, bkey, bx, by, bz, lane, nn);
}

// Same search for voxels that hold more than 32 points (max_points_per_voxel > 32): every probe
// lane strides over its voxel's points.  Rare configuration, kept simple.
__device__ __forceinline__ double scan_hits_wide(const MapView &m, const Probe &pr, double sx, double sy,
                                                 double sz, int lane, double nn[3]) {
    double best = DBL_MAX, bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    if (pr.blk >= 0) {
        const double2 *xy = block_xy(m, pr.blk);
        const double *z = block_z(m, pr.blk);
        for (int k = 0; k < pr.cnt; ++k) {
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
    }
    return group_argmin(best, bkey, bx, by, bz, lane, nn);
}

// GetClosestNeighbor over candidates already staged in LDS by a previous iteration (same voxel
// neighbourhood): 32 lanes stride over the packed list; the candidate number is the tie-break key.
__device__ __forceinline__ double scan_lds(const double *cand, int stride, int E, double sx, double sy,
                                           double sz, int lane, double nn[3]) {
    double best = DBL_MAX, bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    for (int c = lane; c < E; c += 32) {
        const double x = cand[c], y = cand[stride + c], z = cand[2 * stride + c];
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
    return group_argmin(best, bkey, bx, by, bz, lane, nn);
}

// ------------------------------------------------------------------------------------------
// LDS-staged neighbourhoods of the ICP kernel.
//
// A query's 27-voxel neighbourhood is copied once into an LDS region and reused by the following
// ICP iterations (the map does not change during AlignPointsToMap).  The source point moves a
// little every iteration and sooner or later crosses a voxel face; re-fetching from HBM then costs
// three dependent memory round trips, and with thousands of queries SOME query crosses in nearly
// every iteration -- and every workgroup waits for the slowest one.  So the staged window is
// widened by one voxel layer on every side the query is close to (within kWindowMargin of the
// face): 3..4 voxels per axis.  Each staged point carries a tag {voxel offset from the window's
// centre voxel, index inside its voxel}; a scan for a query now in voxel v' visits exactly the
// candidates whose voxel lies in [v'-1, v'+1]^3 -- the reference's 27 voxels, no more -- and breaks
// distance ties by (position of the voxel in the reference's shift table, index in the voxel) like
// the reference's nested strict '<' loops (VoxelHashMap.cpp:46-70).
// ------------------------------------------------------------------------------------------
#ifndef KICP_WINDOW_MARGIN
#define KICP_WINDOW_MARGIN 0.125
#endif
constexpr double kWindowMargin = KICP_WINDOW_MARGIN;  // fraction of a voxel
constexpr int kFillChunk = 12;            // voxels whose points are in flight together during a fill

// first probe of two independent keys issued together, then each chain resolved
__device__ __forceinline__ void map_find_pair(const MapView &m, bool ok0, unsigned long long key0, bool ok1,
                                              unsigned long long key1, int &blk0, int &cnt0, int &blk1,
                                              int &cnt1) {
    uint32_t s0 = hash_key(key0, m.mask), s1 = hash_key(key1, m.mask);
    Slot a, b;
    a.key = b.key = kKeyEmpty;
    a.block = b.block = -1;
    a.count = b.count = 0;
    if (ok0) a = load_slot(m.slots + s0);
    if (ok1) b = load_slot(m.slots + s1);
    blk0 = blk1 = -1;
    cnt0 = cnt1 = 0;
    for (uint32_t probes = 0; ok0 && probes <= m.mask; ++probes) {
        if (a.key == key0) {
            blk0 = a.block;
            cnt0 = a.count;
            break;
        }
        if (a.key == kKeyEmpty) break;
        s0 = (s0 + 1) & m.mask;
        a = load_slot(m.slots + s0);
    }
    for (uint32_t probes = 0; ok1 && probes <= m.mask; ++probes) {
        if (b.key == key1) {
            blk1 = b.block;
            cnt1 = b.count;
            break;
        }
        if (b.key == kKeyEmpty) break;
        s1 = (s1 + 1) & m.mask;
        b = load_slot(m.slots + s1);
    }
    if (blk0 < 0) cnt0 = 0;
    if (blk1 < 0) cnt1 = 0;
}

__device__ __forceinline__ void group_lds_sync() {
    // the 32 lanes of a group are half a wave: LDS traffic between them needs ordering, not a barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int region_doubles(int E) { return 3 * E + (E + 3) / 4; }

// Stage the (widened) neighbourhood of the query s (voxel v) into an LDS region described by *meta;
// `cells` is this group's scratch of 64 int2.  Returns false when the workgroup's pool is exhausted
// (the caller then searches HBM directly).  Needs max_points_per_voxel <= 32.
__device__ __forceinline__ bool window_fill(const MapView &m, const double s[3], const int v[3], int lane,
                                            int2 *cells, double *pool, int pool_doubles, int *bump,
                                            IcpRegionMeta *meta, int &range_err) {
    int lo[3], nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f = s[a] / m.voxel_size - (double)v[a];  // position inside the voxel, [0, 1)
        lo[a] = (f < kWindowMargin) ? -2 : -1;
        const int hi = (f > 1.0 - kWindowMargin) ? 2 : 1;
        nn[a] = hi - lo[a] + 1;
    }
    const int W = nn[0] * nn[1] * nn[2];  // <= 64
    bool ok[2];
    unsigned long long key[2];
    int code[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int w = lane + 32 * h;
        ok[h] = false;
        key[h] = 0;
        code[h] = 0;
        if (w < W) {
            const int iz = w % nn[2], t = w / nn[2], iy = t % nn[1], ix = t / nn[1];
            const int ox = lo[0] + ix, oy = lo[1] + iy, oz = lo[2] + iz;
            const int qx = v[0] + ox, qy = v[1] + oy, qz = v[2] + oz;
            code[h] = w;  // cell number in window order (x-major, z fastest)
            if (voxel_in_range(qx, qy, qz)) {
                ok[h] = true;
                key[h] = pack_voxel(qx, qy, qz);
            } else if (ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1) {
                range_err = 1;
            }
        }
    }
    int blk[2], cnt[2];
    map_find_pair(m, ok[0], key[0], ok[1], key[1], blk[0], cnt[0], blk[1], cnt[1]);
    // candidate numbering: window order, cells 0..31 first
    int incl0 = cnt[0], incl1 = cnt[1];
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int o0 = __shfl_up(incl0, off, 32), o1 = __shfl_up(incl1, off, 32);
        if (lane >= off) {
            incl0 += o0;
            incl1 += o1;
        }
    }
    const int tot0 = __shfl(incl0, 31, 32);
    const int E = tot0 + __shfl(incl1, 31, 32);
    cells[lane] = make_int2(blk[0], cnt[0] | ((incl0 - cnt[0]) << 6) | (code[0] << 18));
    cells[lane + 32] = make_int2(blk[1], cnt[1] | ((tot0 + incl1 - cnt[1]) << 6) | (code[1] << 18));
    // a region of exactly E candidates: reuse the old allocation when it is large enough,
    // otherwise take a new one from the workgroup's pool (never freed within a launch)
    const int need = region_doubles(E);
    int base = meta->base, cap = meta->cap;
    if (need > cap) {
        int nb = -1;
        if (lane == 0) {
            nb = atomicAdd(bump, need);
            if (nb + need > pool_doubles) {
                atomicAdd(bump, -need);
                nb = -1;
            }
        }
        nb = __shfl(nb, 0, 32);
        if (nb < 0) {  // pool exhausted: this query searches HBM directly from now on
            if (lane == 0) {
                meta->valid = 0;
                meta->cap = -1;
            }
            return false;
        }
        base = nb;
        cap = need;
    }
    double *X = pool + base, *Y = X + E, *Z = Y + E;
    unsigned short *T = reinterpret_cast<unsigned short *>(Z + E);
    group_lds_sync();  // cells[] visible to the whole group
    const int half_shift = threadIdx.x & 32;
    unsigned long long hits = (unsigned long long)(unsigned)(__ballot(blk[0] >= 0) >> half_shift) |
                              ((unsigned long long)(unsigned)(__ballot(blk[1] >= 0) >> half_shift) << 32);
    while (hits) {
        double2 xy[kFillChunk];
        double zz[kFillChunk];
        int info[kFillChunk];
#pragma unroll
        for (int u = 0; u < kFillChunk; ++u) {
            info[u] = -1;
            if (hits) {
                const int j = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                const int2 c = cells[j];
                if (lane < (c.y & 63)) {
                    info[u] = c.y;
                    xy[u] = block_xy(m, c.x)[lane];
                    zz[u] = block_z(m, c.x)[lane];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kFillChunk; ++u) {
            if (info[u] >= 0) {
                const int c = ((info[u] >> 6) & 4095) + lane;
                X[c] = xy[u].x;
                Y[c] = xy[u].y;
                Z[c] = zz[u];
                T[c] = (unsigned short)(((info[u] >> 18) & 63) | (lane << 6));  // {cell, index in voxel}
            }
        }
    }
    if (lane == 0) {
        meta->v[0] = v[0];
        meta->v[1] = v[1];
        meta->v[2] = v[2];
        meta->lo[0] = (signed char)lo[0];
        meta->lo[1] = (signed char)lo[1];
        meta->lo[2] = (signed char)lo[2];
        meta->hi[0] = (signed char)(lo[0] + nn[0] - 1);
        meta->hi[1] = (signed char)(lo[1] + nn[1] - 1);
        meta->hi[2] = (signed char)(lo[2] + nn[2] - 1);
        meta->E = E;
        meta->base = base;
        meta->cap = cap;
        meta->valid = 1;
    }
    group_lds_sync();  // candidates and meta visible to the whole group
    return true;
}

// ---- scan of a staged window ---------------------------------------------------------------------
// A staged point carries the tag {cell of its voxel in the window (6 bits), index inside the voxel
// (5 bits)}.  The query sits in the voxel at offset d = (dx, dy, dz) from the window's centre voxel;
// its candidates are the points whose cell lies in [d-1, d+1]^3 -- the reference's 27 voxels.
struct WindowGeom {
    int lo0, lo1, lo2;  // window extent (voxels relative to the centre voxel), low corner
    int n1, n2;         // cells along y and z (3 or 4)
    int dx, dy, dz;     // query voxel relative to the centre voxel
};
__device__ __forceinline__ int div34(int w, int n) { return n == 4 ? (w >> 2) : ((w * 43) >> 7); }  // w < 128
// position of cell w's voxel in the reference's shift table (VoxelHashMap.cpp:35-41), seen from the query
__device__ __forceinline__ int cell_shift_order(int w, const WindowGeom &g) {
    const int t = div34(w, g.n2), iz = w - t * g.n2;
    const int ix = div34(t, g.n1), iy = t - ix * g.n1;
    const int ox = g.lo0 + ix - g.dx, oy = g.lo1 + iy - g.dy, oz = g.lo2 + iz - g.dz;
    return shift_order((ox + 1) * 9 + (oy + 1) * 3 + (oz + 1));
}
__device__ __forceinline__ int tag_order_key(int tag, const WindowGeom &g) {
    return (cell_shift_order(tag & 63, g) << 5) | ((tag >> 6) & 31);
}

// Returns the squared distance (DBL_MAX: no candidate), the neighbour, and the number of map points
// in the query's 27-voxel neighbourhood (= points the reference examines).  Four candidates per
// lane are in flight per trip (all LDS reads issued before the first use), no divergent control
// flow on the common path: the reference's tie rules (strict '<' in shift order, then
// std::min_element's first minimum) only cost anything when two distances are EQUAL.
//   FILTER = false: the window is exactly the query's 27 voxels (no widened side, d = 0).
template <bool FILTER>
__device__ __forceinline__ double scan_window(const double *region, int E, int W, const WindowGeom &g, double sx,
                                              double sy, double sz, int lane, double nn[3], int &examined) {
    constexpr int U = 4;
    const double *X = region, *Y = X + E, *Z = Y + E;
    const unsigned short *T = reinterpret_cast<const unsigned short *>(Z + E);
    const int half_shift = threadIdx.x & 32;
    unsigned long long inmask = ~0ull;
    if (FILTER) {  // which cells of the window belong to the query's 27 voxels: lane j answers for cells j, j + 32
        bool in[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int w = lane + 32 * h;
            const int t = div34(w, g.n2), iz = w - t * g.n2;
            const int ix = div34(t, g.n1), iy = t - ix * g.n1;
            const int ox = g.lo0 + ix - g.dx, oy = g.lo1 + iy - g.dy, oz = g.lo2 + iz - g.dz;
            in[h] = w < W && (unsigned)(ox + 1) < 3u && (unsigned)(oy + 1) < 3u && (unsigned)(oz + 1) < 3u;
        }
        inmask = (unsigned long long)(unsigned)(__ballot(in[0]) >> half_shift) |
                 ((unsigned long long)(unsigned)(__ballot(in[1]) >> half_shift) << 32);
    }
    double best = DBL_MAX;
    int bc = -1, btag = 0, inside = 0;
    for (int c0 = lane; __ballot(c0 < E) != 0ull; c0 += 32 * U) {  // wave-uniform trip count (ballots inside)
        int tag[U];
        double x[U], y[U], z[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 32 * u;
            ok[u] = c < E;
            const int cc = ok[u] ? c : 0;
            tag[u] = ok[u] ? (int)T[cc] : 0;
            x[u] = X[cc];
            y[u] = Y[cc];
            z[u] = Z[cc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool v = ok[u] && (!FILTER || ((inmask >> (tag[u] & 63)) & 1ull));
            if (FILTER) inside += __popc((unsigned)(__ballot(v) >> half_shift));
            const double ex = x[u] - sx, ey = y[u] - sy, ez = z[u] - sz;
            const double d = (ex * ex + ey * ey) + ez * ez;
            if (v && d <= best) {
                bool take = d < best;
                if (!take) take = tag_order_key(tag[u], g) < tag_order_key(btag, g);  // exact tie (rare)
                if (take) {
                    best = d;
                    bc = c0 + 32 * u;
                    btag = tag[u];
                }
            }
        }
    }
    examined = FILTER ? inside : E;
    // lexicographic min over (distance, reference order) across the 32 lanes; the winner's
    // coordinates are then read back from LDS by every lane (same address: broadcast)
    int bkey = bc >= 0 ? tag_order_key(btag, g) : 0x7FFFFFFF;
    group_min(best, bkey, bc);
    const int rc = bc >= 0 ? bc : 0;
    nn[0] = E > 0 ? X[rc] : 0.0;
    nn[1] = E > 0 ? Y[rc] : 0.0;
    nn[2] = E > 0 ? Z[rc] : 0.0;
    return best;
}

__device__ __forceinline__ double closest_neighbor_any(const MapView &m, double sx, double sy, double sz,
                                                       int lane, double nn[3], int &examined, int &range_err) {
    const Probe pr = probe27(m, sx, sy, sz, lane, range_err);
    examined = pr.E;
    if (m.max_points <= 32) return scan_hits<false>(m, pr, sx, sy, sz, lane, nn);
    return scan_hits_wide(m, pr, sx, sy, sz, lane, nn);
}

// ==========================================================================================
// 16-lane groups (one DPP row per group, four groups per wave).  With 32-lane groups the
// association phase is issue-bound: a wave spends ~600 instructions per source point on work that
// does not get shorter with more lanes (transform, voxel, window test, arg-min, weight and sums),
// and two waves share each SIMD.  Half the lanes per point = half the waves per point: the 16
// points of a workgroup fit in four waves, one per SIMD; the other four waves only take part in
// the exchange.  Same staged layout, same tie rules, same results as the 32-lane functions.
// ==========================================================================================
template <int N>
__device__ __forceinline__ int row_shr_or_zero(int v) {  // lane l of a 16-lane row reads lane l-N, or 0
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ int row_inclusive_scan(int v) {
    v += row_shr_or_zero<1>(v);
    v += row_shr_or_zero<2>(v);
    v += row_shr_or_zero<4>(v);
    v += row_shr_or_zero<8>(v);
    return v;
}
__device__ __forceinline__ int row_sum(int v) {  // every lane of the row ends with the row's sum
    v += group_xchg<0>(v);
    v += group_xchg<1>(v);
    v += group_xchg<2>(v);
    v += group_xchg<3>(v);
    return v;
}
__device__ __forceinline__ void group16_min(double &best, int &key, int &payload) {
    group_min_step<0>(best, key, payload);
    group_min_step<1>(best, key, payload);
    group_min_step<2>(best, key, payload);
    group_min_step<3>(best, key, payload);
}
// this group's 16 bits of a wave ballot
__device__ __forceinline__ unsigned row_ballot(bool pred) {
    return (unsigned)(__ballot(pred) >> (threadIdx.x & 48)) & 0xFFFFu;
}

constexpr int kFillChunk16 = 5;  // voxels in flight per trip (a lane holds up to two points of each)

// window_fill for a 16-lane group: lane l answers for the window cells l, l+16, l+32, l+48
__device__ __forceinline__ bool window_fill16(const MapView &m, const double s[3], const int v[3], int lane,
                                              int2 *cells, double *pool, int pool_doubles, int *bump,
                                              IcpRegionMeta *meta, int &range_err) {
    int lo[3], nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f = s[a] / m.voxel_size - (double)v[a];  // position inside the voxel, [0, 1)
        lo[a] = (f < kWindowMargin) ? -2 : -1;
        const int hi = (f > 1.0 - kWindowMargin) ? 2 : 1;
        nn[a] = hi - lo[a] + 1;
    }
    const int W = nn[0] * nn[1] * nn[2];  // <= 64
    bool ok[4];
    unsigned long long key[4];
    uint32_t slot[4];
    Slot pr[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int w = lane + 16 * h;
        ok[h] = false;
        key[h] = 0;
        if (w < W) {
            const int t = div34(w, nn[2]), iz = w - t * nn[2];
            const int ix = div34(t, nn[1]), iy = t - ix * nn[1];
            const int ox = lo[0] + ix, oy = lo[1] + iy, oz = lo[2] + iz;
            const int qx = v[0] + ox, qy = v[1] + oy, qz = v[2] + oz;
            if (voxel_in_range(qx, qy, qz)) {
                ok[h] = true;
                key[h] = pack_voxel(qx, qy, qz);
            } else if (ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1) {
                range_err = 1;
            }
        }
        slot[h] = hash_key(key[h], m.mask);
        pr[h].key = kKeyEmpty;
        pr[h].block = -1;
        pr[h].count = 0;
    }
    // the first probe of all four keys is in flight together, then each chain is resolved
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (ok[h]) pr[h] = load_slot(m.slots + slot[h]);
    int blk[4], cnt[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        blk[h] = -1;
        cnt[h] = 0;
        for (uint32_t probes = 0; ok[h] && probes <= m.mask; ++probes) {
            if (pr[h].key == key[h]) {
                blk[h] = pr[h].block;
                cnt[h] = pr[h].count;
                break;
            }
            if (pr[h].key == kKeyEmpty) break;
            slot[h] = (slot[h] + 1) & m.mask;
            pr[h] = load_slot(m.slots + slot[h]);
        }
        if (blk[h] < 0) cnt[h] = 0;
    }
    // candidate numbering in window order: row h (cells 16h .. 16h+15) comes after rows 0 .. h-1
    int E = 0;
    unsigned long long hits = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int offs = E + row_inclusive_scan(cnt[h]) - cnt[h];
        cells[lane + 16 * h] = make_int2(blk[h], cnt[h] | (offs << 6) | ((lane + 16 * h) << 18));
        E += row_sum(cnt[h]);
        hits |= (unsigned long long)row_ballot(blk[h] >= 0) << (16 * h);
    }
    // a region of exactly E candidates: reuse the old allocation when it is large enough,
    // otherwise take a new one from the workgroup's pool (never freed within a launch)
    const int need = region_doubles(E);
    int base = meta->base, cap = meta->cap;
    if (need > cap) {
        int nb = -1;
        if (lane == 0) {
            nb = atomicAdd(bump, need);
            if (nb + need > pool_doubles) {
                atomicAdd(bump, -need);
                nb = -1;
            }
        }
        nb = row_sum(lane == 0 ? nb + 1 : 0) - 1;  // broadcast lane 0's answer
        if (nb < 0) {  // pool exhausted: this query searches HBM directly from now on
            if (lane == 0) {
                meta->valid = 0;
                meta->cap = -1;
            }
            return false;
        }
        base = nb;
        cap = need;
    }
    double *X = pool + base, *Y = X + E, *Z = Y + E;
    unsigned short *T = reinterpret_cast<unsigned short *>(Z + E);
    group_lds_sync();  // cells[] visible to the whole group
    while (hits) {
        double2 xy[kFillChunk16][2];
        double zz[kFillChunk16][2];
        int info[kFillChunk16];
#pragma unroll
        for (int u = 0; u < kFillChunk16; ++u) {
            info[u] = -1;
            if (hits) {
                const int j = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                const int2 c = cells[j];
                info[u] = c.y;
                const int n = c.y & 63;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (lane + 16 * r < n) {
                        xy[u][r] = block_xy(m, c.x)[lane + 16 * r];
                        zz[u][r] = block_z(m, c.x)[lane + 16 * r];
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < kFillChunk16; ++u) {
            if (info[u] >= 0) {
                const int n = info[u] & 63;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (lane + 16 * r < n) {
                        const int i = lane + 16 * r;
                        const int c = ((info[u] >> 6) & 4095) + i;
                        X[c] = xy[u][r].x;
                        Y[c] = xy[u][r].y;
                        Z[c] = zz[u][r];
                        T[c] = (unsigned short)(((info[u] >> 18) & 63) | (i << 6));  // {cell, index in voxel}
                    }
            }
        }
    }
    if (lane == 0) {
        meta->v[0] = v[0];
        meta->v[1] = v[1];
        meta->v[2] = v[2];
        meta->lo[0] = (signed char)lo[0];
        meta->lo[1] = (signed char)lo[1];
        meta->lo[2] = (signed char)lo[2];
        meta->hi[0] = (signed char)(lo[0] + nn[0] - 1);
        meta->hi[1] = (signed char)(lo[1] + nn[1] - 1);
        meta->hi[2] = (signed char)(lo[2] + nn[2] - 1);
        meta->E = E;
        meta->base = base;
        meta->cap = cap;
        meta->valid = 1;
    }
    group_lds_sync();  // candidates and meta visible to the whole group
    return true;
}

// scan_window for a 16-lane group (see scan_window): 4 candidates per lane in flight per trip
__device__ __forceinline__ double scan_window16(const double *region, int E, int W, const WindowGeom &g, double sx,
                                                double sy, double sz, int lane, double nn[3], int &examined) {
    constexpr int U = 4;
    const double *X = region, *Y = X + E, *Z = Y + E;
    const unsigned short *T = reinterpret_cast<const unsigned short *>(Z + E);
    unsigned long long inmask = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {  // lane l answers for the cells l, l+16, l+32, l+48
        const int w = lane + 16 * h;
        const int t = div34(w, g.n2), iz = w - t * g.n2;
        const int ix = div34(t, g.n1), iy = t - ix * g.n1;
        const int ox = g.lo0 + ix - g.dx, oy = g.lo1 + iy - g.dy, oz = g.lo2 + iz - g.dz;
        const bool in = w < W && (unsigned)(ox + 1) < 3u && (unsigned)(oy + 1) < 3u && (unsigned)(oz + 1) < 3u;
        inmask |= (unsigned long long)row_ballot(in) << (16 * h);
    }
    double best = DBL_MAX;
    int bc = -1, btag = 0, inside = 0;
    for (int c0 = lane; __ballot(c0 < E) != 0ull; c0 += 16 * U) {  // wave-uniform trip count (ballots inside)
        int tag[U];
        double x[U], y[U], z[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 16 * u;
            ok[u] = c < E;
            const int cc = ok[u] ? c : 0;
            tag[u] = ok[u] ? (int)T[cc] : 0;
            x[u] = X[cc];
            y[u] = Y[cc];
            z[u] = Z[cc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool v = ok[u] && ((inmask >> (tag[u] & 63)) & 1ull);
            inside += __popc(row_ballot(v));
            const double ex = x[u] - sx, ey = y[u] - sy, ez = z[u] - sz;
            const double d = (ex * ex + ey * ey) + ez * ez;
            if (v && d <= best) {
                bool take = d < best;
                if (!take) take = tag_order_key(tag[u], g) < tag_order_key(btag, g);  // exact tie (rare)
                if (take) {
                    best = d;
                    bc = c0 + 16 * u;
                    btag = tag[u];
                }
            }
        }
    }
    examined = inside;
    int bkey = bc >= 0 ? tag_order_key(btag, g) : 0x7FFFFFFF;
    group16_min(best, bkey, bc);
    const int rc = bc >= 0 ? bc : 0;
    nn[0] = E > 0 ? X[rc] : 0.0;
    nn[1] = E > 0 ? Y[rc] : 0.0;
    nn[2] = E > 0 ? Z[rc] : 0.0;
    return best;
}

// GetClosestNeighbor straight from HBM by a 16-lane group (queries without a staged window: pool
// exhausted, more rounds than cached, max_points_per_voxel > 32, staging disabled): lane l walks
// the voxels l and l+16 of the shift table point by point.  Simple, not fast.
__device__ __forceinline__ double hbm_search16(const MapView &m, double sx, double sy, double sz, int lane,
                                               double nn[3], int &examined, int &range_err) {
    const int vx = voxel_coord(sx, m.voxel_size), vy = voxel_coord(sy, m.voxel_size),
              vz = voxel_coord(sz, m.voxel_size);
    double best = DBL_MAX, bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF, total = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = lane + 16 * h;
        if (j < 27) {
            const int qx = vx + (int)((kShift.x >> (2 * j)) & 3) - 1;
            const int qy = vy + (int)((kShift.y >> (2 * j)) & 3) - 1;
            const int qz = vz + (int)((kShift.z >> (2 * j)) & 3) - 1;
            if (voxel_in_range(qx, qy, qz)) {
                int cnt = 0;
                const int blk = map_find(m, pack_voxel(qx, qy, qz), cnt);
                if (blk >= 0) {
                    total += cnt;
                    const double2 *xy = block_xy(m, blk);
                    const double *z = block_z(m, blk);
                    for (int k = 0; k < cnt; ++k) {
                        const double dx = xy[k].x - sx, dy = xy[k].y - sy, dz = z[k] - sz;
                        const double d = (dx * dx + dy * dy) + dz * dz;
                        if (d < best) {  // j ascends with h, k ascends: the lane's first minimum in reference order
                            best = d;
                            bx = xy[k].x;
                            by = xy[k].y;
                            bz = z[k];
                            bkey = (j << 20) | k;
                        }
                    }
                }
            } else {
                range_err = 1;
            }
        }
    }
    examined = row_sum(total);
    int src = lane;
    group16_min(best, bkey, src);
    nn[0] = __shfl(bx, src, 16);
    nn[1] = __shfl(by, src, 16);
    nn[2] = __shfl(bz, src, 16);
    return best;
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
// grid = G workgroups (all co-resident, G <= 256 = one per CU, each owning its CU's 160 KiB of
// LDS), 512 threads = 16 groups of 32 lanes.  Per iteration every group walks its points (fixed
// assignment, so the running transformed source of a point is always re-read by the group that
// wrote it):
//   s = est * s            TransformPoints of the previous iteration's estimate (Registration.cpp:159;
//                          est = initial_guess for the first iteration, :147)
//   (nn, d) = closest neighbour among the 27 voxels, keep iff d < max_dist (strict, :72).  The
//             first visit of a voxel neighbourhood copies its candidates into an LDS region
//             (bump-allocated from the workgroup's pool, exactly E points); as long as the query
//             stays in the same voxel later iterations scan LDS only.
//   accumulate the 16 unique scalars of J^T w J and J^T w r with J = [I | -hat(s)], r = s - nn,
//   w = sigma^2 / (sigma + |r|^2)^2 (:81-98).
// Workgroup partials are reduced in LDS in a fixed order, published as tagged granules, gathered
// by EVERY workgroup (one hop, no second broadcast; 26 threads per scalar, each summing a
// contiguous range of workgroups, then the 26 range sums in order: deterministic), and every
// workgroup solves the same 6x6 system redundantly on its first four waves:
// dx = LDLT(JTJ).solve(-JTr), est = exp(dx), T_icp = est * T_icp, stop when |dx| <
// convergence_criterion (:156-163).
// ------------------------------------------------------------------------------------------
struct IcpShared {  // head of the dynamic LDS (kIcpFixedLds bytes with the region records)
    double part[kIcpGroupsPerBlock][kIcpSums];
    double range_sum[kIcpParts][kIcpSums];
    double tot[kIcpSums];
    double est[8];  // q[4], t[3], |dx|
    // kept by ONE thread (kIcpBookThread of workgroup 0), off the critical path and out of registers:
    double T_icp[7];  // accumulated update, q[4] t[3]
    double guess[7];
    unsigned long long ncorr_last, ncorr_total, examined_total;
    double pad2;
    int fail;
    int bump;  // doubles handed out from the candidate pool
    int pad[6];
    int2 cells[kIcpGroupsPerBlock][64];  // per-group scratch of window_fill
};
static_assert(sizeof(IcpShared) + kIcpMaxCachedRounds * kIcpGroupsPerBlock * sizeof(IcpRegionMeta) <= kIcpFixedLds,
              "kIcpFixedLds too small");

// low 32 bits of the 100 MHz wall clock (enough for differences inside one launch)
__device__ __forceinline__ unsigned ticks32() { return (unsigned)wall_clock64(); }

// GL = lanes per source point: 32 (two groups per wave, all eight waves associate) or 16 (four
// groups per wave: waves 0..3 associate, one per SIMD; waves 4..7 only gather and wait)
template <bool PROF, int GL>
__global__ __launch_bounds__(kIcpThreads) void k_icp(IcpParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (all LDS is carved from the dynamic region: a static __shared__ in front of it would
    // shift its base off 8/16-byte alignment)
    IcpShared &sh = *reinterpret_cast<IcpShared *>(smem);
    IcpRegionMeta *metas = reinterpret_cast<IcpRegionMeta *>(smem + sizeof(IcpShared));
    double *pool = reinterpret_cast<double *>(smem + kIcpFixedLds);
    constexpr int kPoolDoubles = (int)((kIcpLdsBytes - kIcpFixedLds) / sizeof(double));

    const int tid = threadIdx.x;
    const int lane = tid & (GL - 1);
    const int grp = tid / GL;  // GL == 16: threads 256.. get grp >= 16 and never own a source point
    const MapView &m = P.map;
    PipeState *st = P.state;

    const unsigned long long launch_cyc = clock64(), launch_tick = wall_clock64();
    const int n = count_of(P.n_ptr, P.n_imm);
    // How many of the launched workgroups take part is decided here, from the actual N_src, so the
    // summation order (hence the result, bit for bit) never depends on host-side hints.
    const int groups_used = P.groups_used;  // groups of a workgroup that take source points (experiments: 8 = one wave per SIMD)
    int G = P.force_blocks > 0 ? P.force_blocks
                               : (n + groups_used * P.points_per_group - 1) / (groups_used * P.points_per_group);
    G = max(1, min(G, (int)gridDim.x));
    if ((int)blockIdx.x >= G) return;
    const int cached_rounds = (P.use_lds && m.max_points <= 32) ? kIcpMaxCachedRounds : 0;
    const unsigned epoch_base = st->epoch_base;
    unsigned t_assoc = 0, t_publish = 0, t_gather = 0, t_solve = 0;
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
    const double inv_voxel = 1.0 / m.voxel_size;

    if (tid == 0) {
        sh.fail = 0;
        sh.bump = 0;
        const SE3 id = se3_identity();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sh.T_icp[i] = id.q[i];
            sh.guess[i] = guess.q[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sh.T_icp[4 + i] = id.t[i];
            sh.guess[4 + i] = guess.t[i];
        }
        sh.ncorr_last = sh.ncorr_total = sh.examined_total = 0ull;
    }
    if (tid < kIcpMaxCachedRounds * kIcpGroupsPerBlock) {
        metas[tid].valid = 0;
        metas[tid].cap = 0;
        metas[tid].base = 0;
        metas[tid].E = 0;
    }
    __syncthreads();

    SE3 est = guess;
    int iterations = 0, converged = 0;
    int range_err = 0;
    bool failed = false;

    const int max_iters = map_empty ? 0 : P.max_iters;
    for (int it = 0; it < max_iters; ++it) {
        const unsigned c0 = PROF ? ticks32() : 0u;
        double acc[kIcpSums];
#pragma unroll
        for (int k = 0; k < kIcpSums; ++k) acc[k] = 0.0;
        // consecutive source points (neighbours in the scan, hence similar neighbourhood sizes) go to
        // different workgroups: point p belongs to workgroup p % G, group (p / G) % 16
        int round = 0;
#ifdef KICP_CONTIGUOUS_POINTS
        // experiment: a workgroup takes CONSECUTIVE source points (neighbours in the scan share most of
        // their voxels, so a workgroup fetches each voxel once instead of eight workgroups fetching it)
        for (int p = (grp < groups_used) ? (int)blockIdx.x * groups_used + grp : n; p < n; p += G * groups_used, ++round) {
#else
        for (int p = (grp < groups_used) ? (int)blockIdx.x + G * grp : n; p < n; p += G * groups_used, ++round) {
#endif
            const bool has_meta = round < cached_rounds;
            IcpRegionMeta *meta = metas + (has_meta ? round : 0) * kIcpGroupsPerBlock + grp;
            const unsigned ta = PROF ? ticks32() : 0u;
            double pin[3];
            if (it > 0 && has_meta) {  // running source point lives in LDS
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
            const int vx = voxel_coord_fast(s[0], m.voxel_size, inv_voxel), vy = voxel_coord_fast(s[1], m.voxel_size, inv_voxel),
                      vz = voxel_coord_fast(s[2], m.voxel_size, inv_voxel);
            const int v[3] = {vx, vy, vz};
            bool cached = false;
            if (has_meta && meta->valid)  // is the 27-neighbourhood of (vx, vy, vz) inside the staged window?
                cached = meta->lo[0] <= vx - meta->v[0] - 1 && vx - meta->v[0] + 1 <= meta->hi[0] &&
                         meta->lo[1] <= vy - meta->v[1] - 1 && vy - meta->v[1] + 1 <= meta->hi[1] &&
                         meta->lo[2] <= vz - meta->v[2] - 1 && vz - meta->v[2] + 1 <= meta->hi[2];
            if (lane == 0) {
                if (has_meta) {
                    meta->s[0] = s[0];
                    meta->s[1] = s[1];
                    meta->s[2] = s[2];
                } else {
                    P.work[3 * p] = s[0];
                    P.work[3 * p + 1] = s[1];
                    P.work[3 * p + 2] = s[2];
                }
            }
            int path = cached ? 0 : 3;  // profiling: 0 staged window, 1 .. widened, 2 staged just now, 3 HBM search
            const unsigned tb = PROF ? ticks32() : 0u;
            if (!cached && has_meta && meta->cap >= 0) {
                if (GL == 16)
                    cached = window_fill16(m, s, v, lane, sh.cells[grp], pool, kPoolDoubles, &sh.bump, meta, range_err);
                else
                    cached = window_fill(m, s, v, lane, sh.cells[grp], pool, kPoolDoubles, &sh.bump, meta, range_err);
                if (cached) path = 2;
            }
            const unsigned tc = PROF ? ticks32() : 0u;
            double nn[3];
            double d2;
            int E;
#ifdef KICP_HEAVY_PRIO
            // a wave that holds a large staged window gets issue priority over its SIMD sibling for
            // the scan: the heaviest query sets the pace of the whole iteration, light waves have slack
            if (__ballot(cached && meta->E > KICP_HEAVY_PRIO) != 0ull) __builtin_amdgcn_s_setprio(2);
#endif
            if (cached) {
                WindowGeom g;
                g.lo0 = meta->lo[0];
                g.lo1 = meta->lo[1];
                g.lo2 = meta->lo[2];
                const int n0 = meta->hi[0] - g.lo0 + 1;
                g.n1 = meta->hi[1] - g.lo1 + 1;
                g.n2 = meta->hi[2] - g.lo2 + 1;
                g.dx = vx - meta->v[0];
                g.dy = vy - meta->v[1];
                g.dz = vz - meta->v[2];
                const int W = n0 * g.n1 * g.n2;
                // one code path for exact and widened windows: the two groups of a wave would otherwise
                // run their scans one after the other whenever they differ
                d2 = (GL == 16) ? scan_window16(pool + meta->base, meta->E, W, g, s[0], s[1], s[2], lane, nn, E)
                                : scan_window<true>(pool + meta->base, meta->E, W, g, s[0], s[1], s[2], lane, nn, E);
                if (path == 0 && W != 27) path = 1;
            } else if (GL == 16) {
                d2 = hbm_search16(m, s[0], s[1], s[2], lane, nn, E, range_err);
            } else {
                const Probe pr = probe27(m, s[0], s[1], s[2], lane, range_err);
                E = pr.E;
                d2 = (m.max_points > 32) ? scan_hits_wide(m, pr, s[0], s[1], s[2], lane, nn)
                                         : scan_hits<false>(m, pr, s[0], s[1], s[2], lane, nn);
            }
            const unsigned td = PROF ? ticks32() : 0u;
            if (PROF && P.prof_groups && lane == 0 && it < kIcpProfIters && round == 0) {
                // per-group record of this iteration (10 ns ticks): where the group's time went
                unsigned *r = P.prof_groups + ((size_t)it * (kIcpMaxBlocks * kIcpGroupsPerBlock) +
                                               (size_t)blockIdx.x * kIcpGroupsPerBlock + grp) * 4;
                r[0] = (unsigned)(ta - c0) | ((unsigned)(tb - ta) << 16);   // wait-in, transform + window test
                r[1] = (unsigned)(tc - tb) | ((unsigned)(td - tc) << 16);   // window fill, scan
                r[2] = (unsigned)(has_meta ? meta->E : 0) | ((unsigned)E << 16);  // staged points, examined
                r[3] = (unsigned)path;
            }
            if (lane == 0) {
                acc[17] += (double)E;
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
#ifdef KICP_HEAVY_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- workgroup reduction (fixed order) ----------------------------------------------
        const unsigned c1 = PROF ? ticks32() : 0u;
        acc[kIcpTickSlot] = (double)(c1 - c0);  // this group's association time (profiling, max-reduced)
        if (lane == 0 && grp < kIcpGroupsPerBlock) {
#pragma unroll
            for (int k = 0; k < kIcpSums; ++k) sh.part[grp][k] = acc[k];
        }
        __syncthreads();
        const unsigned epoch = epoch_base + (unsigned)it + 1u;
        unsigned long long *gran = P.granules + (size_t)(it & 1) * G * (2 * kIcpSums);
        if (tid < 2 * kIcpSums) {
            const int k = tid >> 1;
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < kIcpGroupsPerBlock; ++g) {
                const double pv = sh.part[g][k];
                v = (k == kIcpTickSlot) ? fmax(v, pv) : v + pv;
            }
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            granule_store(gran + (size_t)blockIdx.x * (2 * kIcpSums) + tid, epoch, half);
        }
        // ---- gather every workgroup's partial (bounded spin) --------------------------------
        const unsigned c2 = PROF ? ticks32() : 0u;
        if (tid < kIcpParts * kIcpSums) {
            // thread (k, part) sums scalar k over a contiguous range of workgroups, in order.  All the
            // granules of the range (up to kGatherChunk workgroups x 2) are in flight together; every
            // further pass re-polls, again together, exactly the ones whose tag has not arrived yet:
            // one memory round trip per pass, however many granules are late.
            constexpr int kGatherChunk = 10;  // >= ceil(256 / kIcpParts): one chunk per thread for any grid
            const int k = tid % kIcpSums, part = tid / kIcpSums;
            const int b0 = (G * part) / kIcpParts, b1 = (G * (part + 1)) / kIcpParts;
            double v = 0.0;
            bool fail = false;
            for (int b = b0; b < b1 && !fail; b += kGatherChunk) {
                unsigned long long lo[kGatherChunk], hi[kGatherChunk];
                const unsigned long long *g0 = gran + ((size_t)b * kIcpSums + k) * 2;
                unsigned pending = 0;
#pragma unroll
                for (int u = 0; u < kGatherChunk; ++u) {
                    lo[u] = hi[u] = 0ull;
                    if (b + u < b1) {
                        lo[u] = granule_load(g0 + (size_t)u * (2 * kIcpSums));
                        hi[u] = granule_load(g0 + (size_t)u * (2 * kIcpSums) + 1);
                        pending |= 1u << u;
                    }
                }
#pragma unroll
                for (int u = 0; u < kGatherChunk; ++u)
                    if ((unsigned)(lo[u] >> 32) == epoch && (unsigned)(hi[u] >> 32) == epoch) pending &= ~(1u << u);
                unsigned spins = 0;
                while (pending) {
                    if (PROF) ++gather_passes;
                    if (++spins > P.spin_limit ||
                        ((spins & 255u) == 0 &&
                         (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & E_TIMEOUT))) {
                        fail = true;
                        break;
                    }
#ifndef KICP_POLL_NO_SLEEP
                    __builtin_amdgcn_s_sleep(1);
#endif
#pragma unroll
                    for (int u = 0; u < kGatherChunk; ++u)
                        if ((pending >> u) & 1u) {
                            lo[u] = granule_load(g0 + (size_t)u * (2 * kIcpSums));
                            hi[u] = granule_load(g0 + (size_t)u * (2 * kIcpSums) + 1);
                        }
#pragma unroll
                    for (int u = 0; u < kGatherChunk; ++u)
                        if (((pending >> u) & 1u) && (unsigned)(lo[u] >> 32) == epoch && (unsigned)(hi[u] >> 32) == epoch)
                            pending &= ~(1u << u);
                }
                if (fail) break;
#pragma unroll
                for (int u = 0; u < kGatherChunk; ++u)
                    if (b + u < b1) {
                        const double pv = __longlong_as_double(
                            (long long)(((unsigned long long)(unsigned)hi[u] << 32) | (unsigned)lo[u]));
                        v = (k == kIcpTickSlot) ? fmax(v, pv) : v + pv;
                    }
            }
            sh.range_sum[part][k] = v;
            if (fail) sh.fail = 1;
        }
        __syncthreads();
        if (sh.fail) {
            failed = true;
            break;
        }
        if (tid < kIcpSums) {
            double v = 0.0;
#pragma unroll
            for (int part = 0; part < kIcpParts; ++part) {
                const double pv = sh.range_sum[part][tid];
                v = (tid == kIcpTickSlot) ? fmax(v, pv) : v + pv;
            }
            sh.tot[tid] = v;
        }
        __syncthreads();
        // ---- waves 0..3 (one per SIMD) solve the same system; the result goes through LDS -----
        const unsigned c3 = PROF ? ticks32() : 0u;
        double nrm2 = 0.0;
        if (tid < kIcpSolveThreads) {
            double S[kIcpSums];
#pragma unroll
            for (int k = 0; k < kIcpSums; ++k) S[k] = sh.tot[k];
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
#pragma unroll
            for (int i = 0; i < 6; ++i) nrm2 += dx[i] * dx[i];
            if (tid == 0) {
                sh.est[0] = est.q[0];
                sh.est[1] = est.q[1];
                sh.est[2] = est.q[2];
                sh.est[3] = est.q[3];
                sh.est[4] = est.t[0];
                sh.est[5] = est.t[1];
                sh.est[6] = est.t[2];
                sh.est[7] = nrm2;
            }
        }
        __syncthreads();
        if (tid >= kIcpSolveThreads) {
            est.q[0] = sh.est[0];
            est.q[1] = sh.est[1];
            est.q[2] = sh.est[2];
            est.q[3] = sh.est[3];
            est.t[0] = sh.est[4];
            est.t[1] = sh.est[5];
            est.t[2] = sh.est[6];
            nrm2 = sh.est[7];
        }
        if (blockIdx.x == 0 && tid == kIcpBookThread) {
            // T_icp = est * T_icp (Registration.cpp:161) and the statistics, by a thread whose wave is
            // not on the critical path; sh.tot stays valid until the next gather
            SE3 T;
#pragma unroll
            for (int i = 0; i < 4; ++i) T.q[i] = sh.T_icp[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) T.t[i] = sh.T_icp[4 + i];
            T = se3_mul(est, T);
#pragma unroll
            for (int i = 0; i < 4; ++i) sh.T_icp[i] = T.q[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sh.T_icp[4 + i] = T.t[i];
            const unsigned long long nc = (unsigned long long)sh.tot[16];
            sh.ncorr_last = nc;
            sh.ncorr_total += nc;
            sh.examined_total += (unsigned long long)sh.tot[17];
        }
        iterations = it + 1;
        if (it == 0 && blockIdx.x == 0 && tid == 0) st->prof_it0_ticks = wall_clock64() - launch_tick;
        const unsigned c4 = PROF ? ticks32() : 0u;
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
            r[4] = (unsigned)sh.tot[kIcpTickSlot];  // slowest group's association time, any workgroup
            r[5] = gather_passes;
        }
        gather_passes = 0;
        if (sqrt(nrm2) < P.conv) {
            converged = 1;
            break;
        }
    }

    if (range_err) atomicOr(&st->err, E_RANGE);
    if (failed && tid == 0) atomicOr(&st->err, E_TIMEOUT);

    __syncthreads();  // the bookkeeping thread's last update is in LDS
    if (blockIdx.x == 0 && tid == 0) {
        SE3 T_icp, guess;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T_icp.q[i] = sh.T_icp[i];
            guess.q[i] = sh.guess[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T_icp.t[i] = sh.T_icp[4 + i];
            guess.t[i] = sh.guess[4 + i];
        }
        const unsigned long long examined_total = sh.examined_total, ncorr_last = sh.ncorr_last,
                                 ncorr_total = sh.ncorr_total;
        const SE3 new_pose = se3_mul(T_icp, guess);  // Registration.cpp:166
        st->new_pose = new_pose;
        st->guess = guess;
        st->icp_iterations = iterations;
        st->icp_converged = converged;
        st->icp_examined = examined_total;
        st->icp_ncorr_last = ncorr_last;
        st->icp_ncorr_total = ncorr_total;
        st->n_src = n;
        if (P.prep) {
            st->n_pre = P.prep->n_pre;
            st->n_fd = P.prep->n_fd;
        }
        st->icp_blocks_used = G;
        st->prof[0] = t_assoc;
        st->prof[1] = t_publish;
        st->prof[2] = t_gather;
        st->prof[3] = t_solve;
        st->prof_clock[0] = clock64() - launch_cyc;
        st->prof_clock[1] = wall_clock64() - launch_tick;
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
        double p[3] = {P.xyz[3 * i], P.xyz[3 * i + 1], P.xyz[3 * i + 2]};
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
                                                   unsigned long long key, int idx, int *err) {
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
        const int s = ds_claim(tab, mask, k);
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
        const int s = ds_claim_aggregated(agg, P.ds_tab, P.ds_mask, valid, key, j, P.err);
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
//   k_map_link   every new point finds or claims its voxel's slot (CAS on the packed key), opens or
//                joins the voxel's record of this insert and files its index there (plus a chain
//                for voxels that receive more than kRecList points);
//   k_map_apply  one 32-lane group per record applies the reference's sequential acceptance rule
//                (voxel full? closer than map_resolution to a stored point? else append) to the
//                record's points in ascending point index -- the same result as the serial loop.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_link(MapView m, InsertScratch sc, const double *in, const int *n_ptr,
                                                  int n_imm, const PipeState *state, int use_pose) {
    const int n = count_of(n_ptr, n_imm);
    int *touched = &m.ctr[C_TOUCHED0 + sc.parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        m.ctr[C_TOUCHED0 + (sc.parity ^ 1)] = 0;  // re-arm for the next insert
        // free-block queue: undo the pop cursor's overshoot of the previous insert, then admit the
        // blocks recycled since (nothing else touches these words while k_map_link runs)
        const unsigned head = (unsigned)m.ctr[C_FHEAD], tail = (unsigned)m.ctr[C_FTAIL];
        if ((int)(tail - head) < 0) m.ctr[C_FHEAD] = (int)tail;
        m.ctr[C_FTAIL] = m.ctr[C_FPEND];
    }
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
        sc.world[3 * i] = p[0];
        sc.world[3 * i + 1] = p[1];
        sc.world[3 * i + 2] = p[2];
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
        if (slot < 0) {
            sc.next[i] = -2;
            continue;
        }
        // the voxel's record of this insert: the first point to arrive opens it.  (A plain read of
        // heads[] may be stale, but only as "-1": the CAS then returns the true owner.)
        int t = m.heads[slot];
        if (t < 0) {
            const int nt = atomicAdd(touched, 1);
            const int old = atomicCAS(&m.heads[slot], -1, nt);
            if (old == -1) {
                t = nt;
                sc.rec_slot[nt] = slot;
            } else {
                t = old;
                sc.rec_slot[nt] = -1;  // lost the race: the record stays empty
            }
        }
        const int rank = atomicAdd(&sc.rec_count[t], 1);
        if (rank < kRecList) sc.rec_list[t * kRecList + rank] = i;
        sc.next[i] = atomicExch(&sc.rec_head[t], i);
    }
}

// serial application of one voxel's chain by a single lane (voxels that hold or receive more
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
        int cur = 0x7FFFFFFF;    // next chain entry in ascending point index (= arrival order)
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

// k_map_apply: one 32-lane group per voxel record.  Lane k holds stored point k of the voxel in
// registers, lane l the l-th incoming point; the incoming points are ranked by point index (= the
// reference's arrival order) and offered one after the other; a point is appended (to lane
// `count`) iff the voxel is not full and no stored point -- including the ones appended a moment
// ago -- is closer than map_resolution (VoxelHashMap.cpp:103-110).  Every load of a voxel is
// independent of the others: record -> {slot, list} -> {block, points} is three round trips.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_map_apply(MapView m, InsertScratch sc) {
    constexpr int kGroups = THREADS / 32;
    __shared__ int sh_need[kGroups];
    __shared__ int sh_alloc[4];  // queue position, entries available there, bump base, blocks in the pool
    const int lane = threadIdx.x & 31;
    const int g = threadIdx.x >> 5;
    const int half_shift = threadIdx.x & 32;  // this group's half of the 64-bit wave ballot
    const int touched = m.ctr[C_TOUCHED0 + sc.parity];
    // workgroup-uniform trip count: the eight groups of a workgroup allocate their blocks together
    for (int t0 = blockIdx.x * kGroups; t0 < touched; t0 += gridDim.x * kGroups) {
        const int t = t0 + g;
        int slot = -1, L = 0, head = -1, my_idx = 0x7FFFFFFF;
        if (t < touched) {
            slot = sc.rec_slot[t];
            L = sc.rec_count[t];
            head = sc.rec_head[t];
            if (lane < L && lane < kRecList) my_idx = sc.rec_list[t * kRecList + lane];
            if (lane == 0) {  // leave the record idle for the next insert
                sc.rec_count[t] = 0;
                sc.rec_head[t] = -1;
            }
        }
        Slot *sl = m.slots + max(slot, 0);
        Slot cur;
        cur.key = kKeyEmpty;
        cur.block = -1;
        cur.count = 0;
        if (slot >= 0) {
            cur = load_slot(sl);
            if (lane == 0) m.heads[slot] = -1;
        }
        int b = cur.block;
        int cnt = (b >= 0) ? cur.count : 0;
        const bool need = slot >= 0 && b < 0;  // new voxel (VoxelHashMap.cpp:112-116)
#ifdef KICP_APPLY_PREFETCH
        // experiment: the stored points of an existing voxel and this record's incoming points do not
        // depend on the allocation below -- have them in flight while its atomics round-trip
        double pf_ex = 0.0, pf_ey = 0.0, pf_ez = 0.0, pf_nx = 0.0, pf_ny = 0.0, pf_nz = 0.0;
        const bool pf_ok = slot >= 0 && L <= kRecList && m.max_points <= 32;
        if (pf_ok && b >= 0 && lane < cnt) {
            const double2 xy = block_xy(m, b)[lane];
            pf_ex = xy.x;
            pf_ey = xy.y;
            pf_ez = block_z(m, b)[lane];
        }
        if (pf_ok && lane < L) {
            pf_nx = sc.world[3 * my_idx];
            pf_ny = sc.world[3 * my_idx + 1];
            pf_nz = sc.world[3 * my_idx + 2];
        }
#endif
        // ---- one allocation per workgroup: recycled blocks first, then fresh ones ------------------
        if (lane == 0) sh_need[g] = need ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
#pragma unroll
            for (int k = 0; k < kGroups; ++k) total += sh_need[k];
            int h = 0, avail = 0, bb = 0;
            if (total > 0) {
                const unsigned hh = (unsigned)atomicAdd(&m.ctr[C_FHEAD], total);
                avail = min(max((int)((unsigned)m.ctr[C_FTAIL] - hh), 0), total);
                h = (int)(hh % (unsigned)m.free_cap);
                if (total > avail) bb = atomicAdd(&m.ctr[C_BUMP], total - avail);
                const int fresh_ok = min(max(m.blocks_cap - bb, 0), total - avail);
                atomicAdd(&m.ctr[C_LIVE], avail + fresh_ok);
                if (fresh_ok < total - avail) atomicOr(&m.ctr[C_ERR], E_POOL_FULL);
            }
            sh_alloc[0] = h;
            sh_alloc[1] = avail;
            sh_alloc[2] = bb;
        }
        __syncthreads();
        if (need) {
            int k = 0;
            for (int q = 0; q < g; ++q) k += sh_need[q];
            if (k < sh_alloc[1]) {
                b = m.free_ids[(sh_alloc[0] + k) % m.free_cap];
            } else {
                b = sh_alloc[2] + (k - sh_alloc[1]);
                if (b >= m.blocks_cap) b = -1;
            }
            if (b >= 0 && lane == 0) {
                BlockHdr *hdr = block_hdr(m, b);
                hdr->key = cur.key;
                hdr->slot = slot;
                hdr->count = 0;
                sl->block = b;
            }
        }
        __syncthreads();  // sh_need / sh_alloc are reused by the next trip
        if (slot < 0 || b < 0) continue;  // idle group, a record that lost its race, or pool exhausted
        if (L > kRecList || m.max_points > 32) {  // long list or wide voxel: serial fallback over the chain
            if (lane == 0) map_apply_voxel_serial(m, slot, head, b, sc.world, sc.next);
            continue;
        }
        double2 *pxy = block_xy(m, b);
        double *pz = block_z(m, b);
#ifdef KICP_APPLY_PREFETCH
        double ex = pf_ex, ey = pf_ey, ez = pf_ez;  // stored point `lane` (a new voxel has none)
        double nx = pf_nx, ny = pf_ny, nz = pf_nz;  // incoming point held by this lane
#else
        double ex = 0.0, ey = 0.0, ez = 0.0;  // stored point `lane`
        if (lane < cnt) {
            const double2 xy = pxy[lane];
            ex = xy.x;
            ey = xy.y;
            ez = pz[lane];
        }
        double nx = 0.0, ny = 0.0, nz = 0.0;  // incoming point held by this lane
        if (lane < L) {
            nx = sc.world[3 * my_idx];
            ny = sc.world[3 * my_idx + 1];
            nz = sc.world[3 * my_idx + 2];
        }
#endif
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

// VoxelHashMap::RemovePointsFarFromLocation (VoxelHashMap.cpp:121-132): a voxel dies iff its
// FIRST point is >= max_distance from the origin.  Tombstone the slot, recycle the block.
// When host_rec is given (pipeline mode: this is the last kernel of a frame) the workgroup that
// finishes last copies the map counters and the PipeState behind them -- rec_words 32-bit words,
// contiguous in HBM -- straight into the frame's slot of the host-pinned ring: no blit kernel.
__global__ __launch_bounds__(256) void k_map_prune(MapView m, const PipeState *state,
                                                   int use_state_origin, double ox, double oy,
                                                   double oz, unsigned *host_rec, int rec_words) {
    if (use_state_origin) {
        ox = state->new_pose.t[0];
        oy = state->new_pose.t[1];
        oz = state->new_pose.t[2];
    }
    const double md2 = m.max_distance * m.max_distance;
    const int nb = min(m.ctr[C_BUMP], m.blocks_cap);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        BlockHdr *hdr = block_hdr(m, b);
#ifdef KICP_PRUNE_PARALLEL_LOADS
        // experiment: the three loads of a block are independent of each other (every block below the
        // high-water mark is valid memory), so issue them together instead of count -> point
        const int cnt = hdr->count;
        const double2 p0 = block_xy(m, b)[0];
        const double z0 = block_z(m, b)[0];
        if (cnt <= 0) continue;
        const double dx = p0.x - ox, dy = p0.y - oy, dz = z0 - oz;
#else
        if (hdr->count <= 0) continue;
        const double2 p0 = block_xy(m, b)[0];
        const double dx = p0.x - ox, dy = p0.y - oy, dz = block_z(m, b)[0] - oz;
#endif
        if ((dx * dx + dy * dy) + dz * dz >= md2) {
            Slot *sl = m.slots + hdr->slot;
            sl->key = kKeyTomb;
            sl->block = -1;
            sl->count = 0;
            hdr->count = 0;
            const unsigned k = (unsigned)atomicAdd(&m.ctr[C_FPEND], 1);
            m.free_ids[k % (unsigned)m.free_cap] = b;
            atomicSub(&m.ctr[C_LIVE], 1);
            atomicAdd(&m.ctr[C_TOMB], 1);
        }
    }
    if (host_rec) {
        __shared__ int sh_last;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();  // this workgroup's counter updates are out before it signs off
            sh_last = (atomicAdd(&m.ctr[C_DONE], 1) == (int)gridDim.x - 1);
        }
        __syncthreads();
        if (sh_last) {
            __threadfence();
            const unsigned *src = reinterpret_cast<const unsigned *>(m.ctr);
            for (int w = threadIdx.x; w < rec_words; w += blockDim.x) {
                unsigned v = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w == C_DONE) v = 0;
                host_rec[w] = v;
            }
            if (threadIdx.x == 0) m.ctr[C_DONE] = 0;
        }
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

int icp_prepare() {
    // opt in to the full 160 KiB of LDS (dynamic regions above 64 KiB need the attribute)
    static bool done = false;
    if (done) return 0;
    const void *kernels[4] = {reinterpret_cast<const void *>(k_icp<false, 32>), reinterpret_cast<const void *>(k_icp<true, 32>),
                              reinterpret_cast<const void *>(k_icp<false, 16>), reinterpret_cast<const void *>(k_icp<true, 16>)};
    for (const void *k : kernels) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kIcpLdsBytes);
        if (e != hipSuccess) return (int)e;
    }
    done = true;
    return 0;
}
void launch_icp(IcpParams P, int G, bool profile, hipStream_t s) {
    const bool lanes16 = options().icp_group_lanes == 16;
    if (profile && lanes16)
        hipLaunchKernelGGL((k_icp<true, 16>), dim3(G), dim3(kIcpThreads), kIcpLdsBytes, s, P);
    else if (profile)
        hipLaunchKernelGGL((k_icp<true, 32>), dim3(G), dim3(kIcpThreads), kIcpLdsBytes, s, P);
    else if (lanes16)
        hipLaunchKernelGGL((k_icp<false, 16>), dim3(G), dim3(kIcpThreads), kIcpLdsBytes, s, P);
    else
        hipLaunchKernelGGL((k_icp<false, 32>), dim3(G), dim3(kIcpThreads), kIcpLdsBytes, s, P);
}
void launch_closest_neighbor(const MapView &m, const double *q, int nq, double *nn, double *dist,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_closest_neighbor, dim3(grid_for((long)nq * 32, 256, 2048)), dim3(256), 0, s, m, q,
                       nq, nn, dist);
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
void launch_ds_scatter(const DsParams &P, hipStream_t s) {
    hipLaunchKernelGGL(k_ds_scatter, dim3(grid_for(P.n_max, kScanThreads, 1 << 20)), dim3(kScanThreads), 0, s, P);
}
void launch_map_link(const MapView &m, const InsertScratch &sc, const double *in, const int *n_ptr, int n_imm,
                     int n_max, const PipeState *state, int use_pose, hipStream_t s) {
    hipLaunchKernelGGL(k_map_link, dim3(grid_for(n_max, 256, 2048)), dim3(256), 0, s, m, sc, in, n_ptr, n_imm, state,
                       use_pose);
}
void launch_map_apply(const MapView &m, const InsertScratch &sc, int n_max, hipStream_t s) {
    // one 32-lane group per voxel record (at most one record per incoming point).  A workgroup makes
    // ONE allocation (three returning atomics on shared words) per trip for all its groups, so larger
    // workgroups mean fewer serialised atomics.
    const int threads = (int)options().map_apply_threads;
    if (threads >= 1024)
        hipLaunchKernelGGL(k_map_apply<1024>, dim3(grid_for((long)n_max * 32, 1024, 1024)), dim3(1024), 0, s, m, sc);
    else if (threads >= 512)
        hipLaunchKernelGGL(k_map_apply<512>, dim3(grid_for((long)n_max * 32, 512, 2048)), dim3(512), 0, s, m, sc);
    else
        hipLaunchKernelGGL(k_map_apply<256>, dim3(grid_for((long)n_max * 32, 256, 2048)), dim3(256), 0, s, m, sc);
}
void launch_map_prune(const MapView &m, long bump_ub, const PipeState *state, int use_state_origin,
                      const double origin[3], unsigned *host_rec, int rec_words, hipStream_t s) {
    // grid-stride over the blocks; at most 256 workgroups: every one of them signs off with a fenced
    // atomic (frame-record hand-off), which would serialise over thousands of workgroups
    hipLaunchKernelGGL(k_map_prune, dim3(grid_for(bump_ub, 256, 256)), dim3(256), 0, s, m, state,
                       use_state_origin, origin ? origin[0] : 0.0, origin ? origin[1] : 0.0,
                       origin ? origin[2] : 0.0, host_rec, rec_words);
}
void launch_map_rehash(const MapView &m, long bump_ub, hipStream_t s) {
    hipLaunchKernelGGL(k_map_rehash, dim3(grid_for(bump_ub, 256, 2048)), dim3(256), 0, s, m);
}
void launch_map_count_points(const MapView &m, long bump_ub, hipStream_t s) {
    hipLaunchKernelGGL(k_map_count_points, dim3(grid_for(bump_ub, 256, 1024)), dim3(256), 0, s, m);
}

}  // namespace kicp
