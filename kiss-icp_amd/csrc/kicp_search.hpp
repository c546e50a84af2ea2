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
// resolve a lookup over the slots already loaded: selects only (no early exit, so the loads stay together)
__device__ __forceinline__ bool probe_resolve(const Slot (&a)[kProbeAhead], unsigned long long key, int &blk, int &cnt) {
    bool done = false;
    blk = -1;
    cnt = 0;
#pragma unroll
    for (int i = 0; i < kProbeAhead; ++i) {
        const bool hit = !done && a[i].key == key;
        const bool end = !done && a[i].key == kKeyEmpty;
        blk = hit ? a[i].block : blk;
        cnt = hit ? a[i].count : cnt;
        done = done || hit || end;
    }
    return done;
}
// the rest of a chain longer than kProbeAhead (rare at load factor <= 1/2): one slot at a time
__device__ __forceinline__ void probe_tail(const MapView &m, uint32_t s, unsigned long long key, int &blk, int &cnt) {
    for (uint32_t probes = kProbeAhead; probes <= m.mask; ++probes) {
        const Slot sl = load_slot(m.slots + s);
        if (sl.key == key) {
            blk = sl.block;
            cnt = sl.count;
            return;
        }
        if (sl.key == kKeyEmpty) return;
        s = (s + 1) & m.mask;
    }
}
__device__ __forceinline__ int map_find(const MapView &m, unsigned long long key, int &count) {
    const uint32_t s = hash_key(key, m.mask);
    Slot a[kProbeAhead];
#pragma unroll
    for (int i = 0; i < kProbeAhead; ++i) a[i] = load_slot(m.slots + ((s + i) & m.mask));
    int blk, cnt;
    if (!probe_resolve(a, key, blk, cnt)) probe_tail(m, (s + kProbeAhead) & m.mask, key, blk, cnt);
    count = cnt;
    return blk;
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
    const bool done0 = probe_resolve(a, key0, blk0, cnt0);
    const bool done1 = probe_resolve(b, key1, blk1, cnt1);
    if (!done0) probe_tail(m, (s0 + kProbeAhead) & m.mask, key0, blk0, cnt0);
    if (!done1) probe_tail(m, (s1 + kProbeAhead) & m.mask, key1, blk1, cnt1);
    if (blk0 < 0) cnt0 = 0;
    if (blk1 < 0) cnt1 = 0;
}

__device__ __forceinline__ void group_lds_sync() {
    // the 32 lanes of a group are half a wave: LDS traffic between them needs ordering, not a barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// Staged windows.  A query's region (doubles out of its workgroup's pool) holds
//   P[E]   the E map points of the window's voxels as xyz triples (24 bytes each), voxel after voxel in
//          window order (x-major, z fastest);
//   C[64]  per window cell {first point (bits 6..), points (bits 0..5)};
//   I[..]  the SCAN LIST: positions (into P) of the points of the query's 27 voxels, in exactly the order the
//          reference visits them (shift table VoxelHashMap.cpp:35-41, then index inside the voxel), 16 bits
//          each.  The per-iteration search strides over this list and nothing else: no geometry, no tags, no
//          filtering, and strict '<' alone reproduces the reference's tie rules because the list is in its
//          order.  The list is rebuilt (from C, no HBM access) only when the query moves to another voxel of
//          its window -- a few times per launch.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int region_doubles(int E) { return 3 * E + 32 + (E + 3) / 4; }
__device__ __forceinline__ unsigned *region_cells(double *region, int E) {
    return reinterpret_cast<unsigned *>(region + 3 * E);
}
__device__ __forceinline__ unsigned short *region_list(double *region, int E) {
    return reinterpret_cast<unsigned short *>(region + 3 * E + 32);
}

struct WindowGeom {
    int lo0, lo1, lo2;  // window extent (voxels relative to the centre voxel), low corner
    int n0, n1, n2;     // cells along x, y, z (3 or 4)
    int dx, dy, dz;     // query voxel relative to the centre voxel
};
__device__ __forceinline__ int div34(int w, int n) { return n == 4 ? (w >> 2) : ((w * 43) >> 7); }  // w < 128

// (Re)build the scan list of a region for the query offset g.d*; `scratch` = 32 ints of this group.
// Lane l answers for the window cells l and l + 32.
__device__ __forceinline__ void window_index(double *region, int E, const WindowGeom &g, int lane, int *scratch,
                                             IcpRegionMeta *meta) {
    const unsigned *C = region_cells(region, E);
    unsigned short *I = region_list(region, E);
    const int W = g.n0 * g.n1 * g.n2;
    int so[2], cnt[2], off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int w = lane + 32 * h;
        const unsigned c = C[w];
        cnt[h] = (int)(c & 63u);
        off[h] = (int)(c >> 6);
        const int t = div34(w, g.n2), iz = w - t * g.n2;
        const int ix = div34(t, g.n1), iy = t - ix * g.n1;
        const int ox = g.lo0 + ix - g.dx, oy = g.lo1 + iy - g.dy, oz = g.lo2 + iz - g.dz;
        const bool in = w < W && (unsigned)(ox + 1) < 3u && (unsigned)(oy + 1) < 3u && (unsigned)(oz + 1) < 3u;
        so[h] = in ? shift_order((ox + 1) * 9 + (oy + 1) * 3 + (oz + 1)) : 31;
    }
    // point counts by position in the shift table -> where each voxel's run starts in the list
    scratch[lane] = 0;
    group_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (so[h] != 31) scratch[so[h]] = cnt[h];  // the 27 positions are distinct cells
    group_lds_sync();
    const int mine = scratch[lane];  // entries 27..31 stay 0
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int up = __shfl_up(incl, o, 32);
        if (lane >= o) incl += up;
    }
    const int examined = __shfl(incl, 31, 32);
    group_lds_sync();  // everybody has read its count
    scratch[lane] = incl - mine;
    group_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (so[h] != 31) {
            const int k0 = scratch[so[h]];
            for (int j = 0; j < cnt[h]; ++j) I[k0 + j] = (unsigned short)(off[h] + j);
        }
    if (lane == 0) {
        meta->d[0] = (signed char)g.dx;
        meta->d[1] = (signed char)g.dy;
        meta->d[2] = (signed char)g.dz;
        meta->examined = (unsigned short)examined;
    }
    group_lds_sync();
}

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
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int w = lane + 32 * h;
        ok[h] = false;
        key[h] = 0;
        if (w < W) {
            const int iz = w % nn[2], t = w / nn[2], iy = t % nn[1], ix = t / nn[1];
            const int ox = lo[0] + ix, oy = lo[1] + iy, oz = lo[2] + iz;
            const int qx = v[0] + ox, qy = v[1] + oy, qz = v[2] + oz;
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
    // staging order: window order, cells 0..31 first
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
    const int offs0 = incl0 - cnt[0], offs1 = tot0 + incl1 - cnt[1];
    cells[lane] = make_int2(blk[0], cnt[0] | (offs0 << 6));
    cells[lane + 32] = make_int2(blk[1], cnt[1] | (offs1 << 6));
    // a region of exactly E points: reuse the old allocation when it is large enough,
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
    double *P = pool + base;
    unsigned *C = region_cells(P, E);
    C[lane] = (unsigned)(cnt[0] | (offs0 << 6));
    C[lane + 32] = (unsigned)(cnt[1] | (offs1 << 6));
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
                double *q = P + 3 * ((info[u] >> 6) + lane);
                q[0] = xy[u].x;
                q[1] = xy[u].y;
                q[2] = zz[u];
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
    WindowGeom g;
    g.lo0 = lo[0];
    g.lo1 = lo[1];
    g.lo2 = lo[2];
    g.n0 = nn[0];
    g.n1 = nn[1];
    g.n2 = nn[2];
    g.dx = g.dy = g.dz = 0;  // the query sits in the window's centre voxel right now
    window_index(P, E, g, lane, reinterpret_cast<int *>(cells), meta);  // (ends with a group sync: points, cells, list, meta visible)
    return true;
}

// minimum of (distance, key) over the 32 lanes of a group, lexicographic; every lane ends with the winner.
// Two all-reduces (DPP row operations + one swizzle each): the distance first, then the key among the lanes
// that hold that distance.
template <int STEP>
__device__ __forceinline__ void group_fmin_step(double &v) {
    const double o = group_xchg<STEP>(v);
    v = o < v ? o : v;
}
template <int STEP>
__device__ __forceinline__ void group_imin_step(int &v) {
    const int o = group_xchg<STEP>(v);
    v = o < v ? o : v;
}
__device__ __forceinline__ void group_min_dist_key(double &best, int &key) {
    double g = best;
    group_fmin_step<0>(g);
    group_fmin_step<1>(g);
    group_fmin_step<2>(g);
    group_fmin_step<3>(g);
    group_fmin_step<4>(g);
    int k = (best == g) ? key : 0x7FFFFFFF;
    group_imin_step<0>(k);
    group_imin_step<1>(k);
    group_imin_step<2>(k);
    group_imin_step<3>(k);
    group_imin_step<4>(k);
    best = g;
    key = k;
}

// GetClosestNeighbor over a staged window: 32 lanes stride over the scan list, four candidates per lane in
// flight per trip, no divergent control flow.  A lane meets its candidates in the reference's order, so
// strict '<' keeps the reference's choice among equal distances; across lanes the smaller list position wins.
// Returns the squared distance (DBL_MAX: no candidate) and the neighbour.
__device__ __forceinline__ double scan_list(const double *region, int E, int examined, double sx, double sy, double sz,
                                            int lane, double nn[3]) {
    constexpr int U = 4;
    const double *P = region;
    const unsigned short *I = region_list(const_cast<double *>(region), E);
    double best = DBL_MAX;
    int bi = 0x7FFFFFFF;
    for (int i0 = lane; __ballot(i0 < examined) != 0ull; i0 += 32 * U) {  // wave-uniform trip count
        int pos[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 32 * u;
            pos[u] = (int)I[i < examined ? i : 0];
        }
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double *q = P + 3 * pos[u];
            x[u] = q[0];
            y[u] = q[1];
            z[u] = q[2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 32 * u;
            const double ex = x[u] - sx, ey = y[u] - sy, ez = z[u] - sz;
            const double d = (ex * ex + ey * ey) + ez * ez;
            const bool take = (i < examined) & (d < best);
            best = take ? d : best;
            bi = take ? i : bi;
        }
    }
    group_min_dist_key(best, bi);
    const int p = (int)I[bi != 0x7FFFFFFF ? bi : 0];
    const bool found = bi != 0x7FFFFFFF;
    nn[0] = found ? P[3 * p] : 0.0;
    nn[1] = found ? P[3 * p + 1] : 0.0;
    nn[2] = found ? P[3 * p + 2] : 0.0;
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
