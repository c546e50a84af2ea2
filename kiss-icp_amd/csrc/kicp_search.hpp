// kicp_search.hpp -- device-side voxel-hash lookups and the cooperative nearest-neighbour search
// shared by k_icp (kicp_icp.hip) and k_closest_neighbor (kicp_map.hip).
//   VoxelHashMap::GetClosestNeighbor        core/VoxelHashMap.cpp:46-70 (shift table :35-41)
#pragma once

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

// block id (and stored point count) of a voxel, or -1.  Linear probing; the first kProbeAhead slots of the
// chain are loaded TOGETHER (they share one or two cache lines), so that a lookup is one memory round
// trip however the chain happens to fall: with 27..64 independent lookups per query and a wave waiting for
// its slowest lane, dependent probe steps were the longest part of a window fill.
constexpr int kProbeAhead = 4;
__device__ __forceinline__ int map_find(const MapView &m, unsigned long long key, int &count) {
    uint32_t s = hash_key(key, m.mask);
    Slot a[kProbeAhead];
#pragma unroll
    for (int i = 0; i < kProbeAhead; ++i) a[i] = load_slot(m.slots + ((s + i) & m.mask));
#pragma unroll
    for (int i = 0; i < kProbeAhead; ++i) {
        if (a[i].key == key) {
            count = a[i].count;
            return a[i].block;
        }
        if (a[i].key == kKeyEmpty) return -1;
    }
    s = (s + kProbeAhead) & m.mask;
    for (uint32_t probes = kProbeAhead; probes <= m.mask; ++probes) {
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
constexpr double kWindowMargin = 0.125;  // fraction of a voxel
constexpr int kFillChunk = 12;            // voxels whose points are in flight together during a fill

// two independent lookups, the first kProbeAhead slots of both chains in flight together
__device__ __forceinline__ void map_find_pair(const MapView &m, bool ok0, unsigned long long key0, bool ok1,
                                              unsigned long long key1, int &blk0, int &cnt0, int &blk1,
                                              int &cnt1) {
    const uint32_t s0 = hash_key(key0, m.mask), s1 = hash_key(key1, m.mask);
    Slot a[kProbeAhead], b[kProbeAhead];
#pragma unroll
    for (int i = 0; i < kProbeAhead; ++i) {
        a[i].key = b[i].key = kKeyEmpty;
        a[i].block = b[i].block = -1;
        a[i].count = b[i].count = 0;
        if (ok0) a[i] = load_slot(m.slots + ((s0 + i) & m.mask));
        if (ok1) b[i] = load_slot(m.slots + ((s1 + i) & m.mask));
    }
    blk0 = blk1 = -1;
    cnt0 = cnt1 = 0;
    bool done0 = !ok0, done1 = !ok1;
#pragma unroll
    for (int i = 0; i < kProbeAhead; ++i) {
        if (!done0) {
            if (a[i].key == key0) {
                blk0 = a[i].block;
                cnt0 = a[i].count;
                done0 = true;
            } else if (a[i].key == kKeyEmpty) {
                done0 = true;
            }
        }
        if (!done1) {
            if (b[i].key == key1) {
                blk1 = b[i].block;
                cnt1 = b[i].count;
                done1 = true;
            } else if (b[i].key == kKeyEmpty) {
                done1 = true;
            }
        }
    }
    // a chain longer than kProbeAhead (rare at load factor <= 1/2): one slot at a time
    uint32_t s = (s0 + kProbeAhead) & m.mask;
    for (uint32_t probes = kProbeAhead; !done0 && probes <= m.mask; ++probes) {
        const Slot sl = load_slot(m.slots + s);
        if (sl.key == key0) {
            blk0 = sl.block;
            cnt0 = sl.count;
            break;
        }
        if (sl.key == kKeyEmpty) break;
        s = (s + 1) & m.mask;
    }
    s = (s1 + kProbeAhead) & m.mask;
    for (uint32_t probes = kProbeAhead; !done1 && probes <= m.mask; ++probes) {
        const Slot sl = load_slot(m.slots + s);
        if (sl.key == key1) {
            blk1 = sl.block;
            cnt1 = sl.count;
            break;
        }
        if (sl.key == kKeyEmpty) break;
        s = (s + 1) & m.mask;
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

// doubles of a region of E candidates: X[E] Y[E] Z[E], then 16-bit tags T[E] and 16-bit keys K[E]
__device__ __forceinline__ int region_doubles(int E) { return 3 * E + 2 * ((E + 3) / 4); }
__device__ __forceinline__ unsigned short *region_tags(double *region, int E) {
    return reinterpret_cast<unsigned short *>(region + 3 * E);
}
__device__ __forceinline__ unsigned short *region_keys(double *region, int E) {
    return reinterpret_cast<unsigned short *>(region + 3 * E + (E + 3) / 4);
}
constexpr int kKeyOutside = 0xFFFF;  // key of a staged point that is not in the query's 27 voxels

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
    int code[2], so[2];  // cell number in window order; position in the reference's shift table (31: not one of the 27)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int w = lane + 32 * h;
        ok[h] = false;
        key[h] = 0;
        code[h] = 0;
        so[h] = 31;
        if (w < W) {
            const int iz = w % nn[2], t = w / nn[2], iy = t % nn[1], ix = t / nn[1];
            const int ox = lo[0] + ix, oy = lo[1] + iy, oz = lo[2] + iz;
            const int qx = v[0] + ox, qy = v[1] + oy, qz = v[2] + oz;
            code[h] = w;  // cell number in window order (x-major, z fastest)
            const bool core = ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1;
            if (core) so[h] = shift_order((ox + 1) * 9 + (oy + 1) * 3 + (oz + 1));
            if (voxel_in_range(qx, qy, qz)) {
                ok[h] = true;
                key[h] = pack_voxel(qx, qy, qz);
            } else if (core) {
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
    // points the reference examines from here: those of the 27 voxels around v
    int core_pts = (so[0] != 31 ? cnt[0] : 0) + (so[1] != 31 ? cnt[1] : 0);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) core_pts += __shfl_xor(core_pts, off, 32);
    cells[lane] = make_int2(blk[0], cnt[0] | ((incl0 - cnt[0]) << 6) | (code[0] << 18) | (so[0] << 24));
    cells[lane + 32] = make_int2(blk[1], cnt[1] | ((tot0 + incl1 - cnt[1]) << 6) | (code[1] << 18) | (so[1] << 24));
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
    unsigned short *T = region_tags(X, E), *K = region_keys(X, E);
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
                const int so_c = (info[u] >> 24) & 31;  // the query sits in the window's centre voxel right now
                K[c] = (unsigned short)(so_c == 31 ? kKeyOutside : ((so_c << 5) | lane));
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
        meta->d[0] = meta->d[1] = meta->d[2] = 0;
        meta->examined = (unsigned short)core_pts;
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

// The query has moved to another voxel of its (widened) window: recompute every staged point's key for
// the new offset g.d* -- {position of its voxel in the reference's shift table seen from the query, index
// inside the voxel}, or kKeyOutside when the voxel is not one of the query's 27 -- and the number of points
// the reference would examine.  Happens a few times per query and launch; the per-iteration scan then
// needs neither the window geometry nor the tags.
__device__ __forceinline__ void window_rekey(double *region, int E, const WindowGeom &g, int lane, IcpRegionMeta *meta) {
    const unsigned short *T = region_tags(region, E);
    unsigned short *K = region_keys(region, E);
    int inside = 0;
    for (int c = lane; c < E; c += 32) {
        const int tag = T[c];
        const int w = tag & 63;
        const int t = div34(w, g.n2), iz = w - t * g.n2;
        const int ix = div34(t, g.n1), iy = t - ix * g.n1;
        const int ox = g.lo0 + ix - g.dx, oy = g.lo1 + iy - g.dy, oz = g.lo2 + iz - g.dz;
        const bool in = (unsigned)(ox + 1) < 3u && (unsigned)(oy + 1) < 3u && (unsigned)(oz + 1) < 3u;
        K[c] = (unsigned short)(in ? ((shift_order((ox + 1) * 9 + (oy + 1) * 3 + (oz + 1)) << 5) | ((tag >> 6) & 31)) : kKeyOutside);
        inside += in ? 1 : 0;
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) inside += __shfl_xor(inside, off, 32);
    if (lane == 0) {
        meta->d[0] = (signed char)g.dx;
        meta->d[1] = (signed char)g.dy;
        meta->d[2] = (signed char)g.dz;
        meta->examined = (unsigned short)inside;
    }
    group_lds_sync();
}

// GetClosestNeighbor over a staged window whose keys are current: 32 lanes stride over the packed list, four
// candidates per lane in flight per trip, no divergent control flow.  Strict '<' in shift order, then
// std::min_element's first minimum inside a voxel (VoxelHashMap.cpp:55-63) = the lexicographic minimum of
// (squared distance, key); keys only matter when two distances are EQUAL.
__device__ __forceinline__ double scan_keys(const double *region, int E, double sx, double sy, double sz, int lane,
                                            double nn[3]) {
    constexpr int U = 4;
    const double *X = region, *Y = X + E, *Z = Y + E;
    const unsigned short *K = region_keys(const_cast<double *>(region), E);
    double best = DBL_MAX;
    int bkey = kKeyOutside, bc = -1;
    for (int c0 = lane; __ballot(c0 < E) != 0ull; c0 += 32 * U) {  // wave-uniform trip count
        int key[U];
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 32 * u;
            const bool ok = c < E;
            const int cc = ok ? c : 0;
            const int k = (int)K[cc];
            key[u] = ok ? k : kKeyOutside;
            x[u] = X[cc];
            y[u] = Y[cc];
            z[u] = Z[cc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double ex = x[u] - sx, ey = y[u] - sy, ez = z[u] - sz;
            const double d = (ex * ex + ey * ey) + ez * ez;
            const bool take = key[u] != kKeyOutside && (d < best || (d == best && key[u] < bkey));
            best = take ? d : best;
            bkey = take ? key[u] : bkey;
            bc = take ? c0 + 32 * u : bc;
        }
    }
    int gkey = bc >= 0 ? bkey : 0x7FFFFFFF;
    group_min(best, gkey, bc);
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

}  // namespace kicp
