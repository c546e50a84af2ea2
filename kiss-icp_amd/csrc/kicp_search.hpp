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

// Two adjacent granules (the two halves of one double) moved by ONE 16-byte access: half as many entries in the
// memory queue of the gathering CU (where the price of a hand-off sits) and half as many fabric writes on the
// publishing side.  Each 8-byte half still validates itself by its own tag, so nothing depends on the 16 bytes
// travelling together.  Buffer instructions because the builtin takes the cache policy (16 = sc1) and the compiler
// keeps track of the outstanding loads, which it cannot do for inline assembly.
typedef int kicp_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t granule_rsrc(const unsigned long long *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void granule_load_pair(__amdgpu_buffer_rsrc_t r, unsigned byte_offset, unsigned long long &a,
                                                  unsigned long long &b) {
    const kicp_v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 16);
    a = (unsigned long long)(unsigned)v.x | ((unsigned long long)(unsigned)v.y << 32);
    b = (unsigned long long)(unsigned)v.z | ((unsigned long long)(unsigned)v.w << 32);
}
__device__ __forceinline__ void granule_store_pair(__amdgpu_buffer_rsrc_t r, unsigned byte_offset, unsigned tag, unsigned lo, unsigned hi) {
    kicp_v4i v;
    v.x = (int)lo;
    v.y = (int)tag;
    v.z = (int)hi;
    v.w = (int)tag;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_offset, 0, 16);
}

// The sum of the 16 lanes of a DPP row, in every lane of it: four steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) --
// the same tree whatever the lane: a + b on one side is b + a on the other.  All 16 lanes of the row must be active.
template <int CTRL>
__device__ __forceinline__ double row16_dpp(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double row16_sum(double v) {
    v = v + row16_dpp<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v = v + row16_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v = v + row16_dpp<0x141>(v);  // row_half_mirror
    v = v + row16_dpp<0x140>(v);  // row_mirror
    return v;
}

// The thread index through an empty asm statement: the compiler cannot prove two reads equal or loop-invariant, so what is
// derived from it (predicates, lane numbers, LDS addresses) is computed where it is used -- one v_and -- instead of being
// hoisted out of k_icp's iteration loop and kept alive across it (kicp_icp.hip: the loop's own tid, and why).
__device__ __forceinline__ int kicp_tid() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ int count_of(const int *n_ptr, int n_imm) { return n_ptr ? *n_ptr : n_imm; }

// Point index of a sorted tile key {30-bit Morton code | 24-bit index} (kicp_sort.hip).
// THE ROOT CAUSE OF ROUND 3's "memory fault that vanishes with a clamp" (profiles/README.md r05_g): hipcc (ROCm 7.2, gfx950)
// miscompiles  frame[3 * (int)(key & 0xFFFFFF)]  when the product feeds a 64-bit address.  The 24-bit mask lets the backend
// treat the multiplication as a 24-bit one (MUL_U24), whose operands demand only their low 24 bits -- so the AND is deleted
// as redundant -- and then the multiply-add is selected as the FULL 32-bit v_mad_u64_u32 after all:
//     global_load_dword v6, v[8:9], off                   ; low half of the key
//     v_mad_u64_u32 v[12:13], s[18:19], v6, 24, s[20:21]  ; frame + 24 * v6 -- the Morton code's low byte still in bits 24..31
// i.e. a load up to 96 GB beyond the cloud whenever that byte is not zero: a memory fault, or -- where the address happens
// to be mapped -- a garbage point (in the run-weight prologue that changes a weight, hence nothing).  Any use of the masked
// value that is not a multiplication (round 3's min(p, n - 1)) makes the compiler keep the AND, which is why the clamp
// "fixed" it.  Here the masked value goes through an empty asm statement: the compiler cannot see through it, so the AND
// is materialised, at no instruction's cost.  tests/test_codegen_masks.py disassembles the built library and fails if a
// freshly loaded word ever reaches such a multiply-add unmasked again.
__device__ __forceinline__ int key_index(unsigned long long key) {
    int p = (int)((unsigned)key & 0xFFFFFFu);
    asm volatile("" : "+v"(p));
    return p;
}

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
// the rest of a chain longer than kProbeAhead (at load factor <= 1/2 one lookup in ten or twenty): kProbeAhead slots per
// memory round trip again -- a slot at a time, a chain of twelve cost eight dependent round trips, and a workgroup that
// looks up a thousand cells waits for the longest of them
__device__ __forceinline__ void probe_tail(const MapView &m, uint32_t s, unsigned long long key, int &blk, int &cnt) {
    for (uint32_t probes = kProbeAhead; probes <= m.mask; probes += kProbeAhead) {
        Slot a[kProbeAhead];
#pragma unroll
        for (int i = 0; i < kProbeAhead; ++i) a[i] = load_slot(m.slots + ((s + i) & m.mask));
        if (probe_resolve(a, key, blk, cnt)) return;
        s = (s + kProbeAhead) & m.mask;
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

// ------------------------------------------------------------------------------------------
// NORMS, not squared distances.  The reference compares (p - q).norm() with strict '<', inside a voxel (std::min_element)
// and across voxels (VoxelHashMap.cpp:58-63): of all candidates whose ROUNDED SQUARE ROOTS are equal the earliest in
// (shift, index) order wins.  The searches here compare squared distances -- a square root per candidate would double their
// arithmetic --, which picks the same candidate unless two squared distances differ and their roots round to the same
// double: values within a unit or two in the last place of each other (the root halves relative differences), once in
// ~1e10 searches on real clouds.  So every fast search DETECTS that case -- a candidate it dropped, or another lane's best,
// within kNormTie of the minimum (2^-50 relative: squared distances farther apart than that have different roots) -- and
// the query is then searched again by closest_neighbor_exact: the map-direct search below with the reference's own
// comparison, sqrt included.  Cells are skipped by their box bounds only when the bound is beyond kNormTie of the limit.
// ------------------------------------------------------------------------------------------
constexpr double kNormTie = 0x1.0000000000004p+0;  // 1 + 2^-50

// inclusive prefix sum over the 32 lanes of a group, in registers: four row_shr steps inside the rows of 16 lanes (lanes shifted
// in from outside a row read 0), then the first row's total -- its lane 15 -- into the second row (row_bcast:15 on rows 1 and 3 of
// the wave: both groups of a wave at once).  Five VALU operations where five __shfl_up are five trips through the LDS crossbar.
constexpr bool kListScanDpp = true;  // (tile_list_build; false: the shuffles, for A/Bs)
__device__ __forceinline__ int group_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xA, 0xF, true);
    return v;
}
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
// sec: a lower bound of the squared distances of this lane's candidates OTHER than its best (DBL_MAX: none, or not tracked).
// Returns true when the minimum may not be the reference's choice: another lane's best, or a candidate a lane did not keep,
// within kNormTie of the group's minimum g (and not equal to it across lanes: equal squares are settled by the key).
__device__ __forceinline__ bool group_norm_tie(double best, double g, double sec) {
    const double lim = g * kNormTie;
    const bool mine = g < DBL_MAX && (sec <= lim || (best <= lim && best != g));
    return (unsigned)(__ballot(mine) >> (kicp_tid() & 32)) != 0u;
}

//   FILL: additionally stage the candidates, packed in (shift, index) order, into an LDS region
//   {x[stride], y[stride], z[stride]} so later ICP iterations of the same query never leave the CU.
// Returns the squared distance (DBL_MAX when the neighbourhood is empty) and the neighbour.
// tie (when given): set when the answer may differ from the reference's because of a tie in NORM (see above)
template <bool FILL>
__device__ __forceinline__ double scan_hits(const MapView &m, const Probe &pr, double sx, double sy, double sz,
                                            int lane, double nn[3], double *cand = nullptr, int stride = 0, bool *tie = nullptr) {
    // hit mask of this group (the wave holds two groups)
    const unsigned long long ball = __ballot(pr.blk >= 0);
    unsigned hits = (unsigned)(ball >> (kicp_tid() & 32));
    double best = DBL_MAX;
    double bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    double prev = DBL_MAX;  // what this lane's best replaced last (its candidates arrive in the reference's order)
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
                    prev = best;  // (the candidate given up came first: if its norm is the same, the reference keeps it)
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
    if (tie) {
        double g = best;
        group_fmin_step<0>(g);
        group_fmin_step<1>(g);
        group_fmin_step<2>(g);
        group_fmin_step<3>(g);
        group_fmin_step<4>(g);
        *tie = group_norm_tie(best, g, prev);
    }
    return group_argmin(best, bkey, bx, by, bz, lane, nn);
}

// The same search with the REFERENCE'S OWN comparison: norms (sqrt of the squared distance, correctly rounded like
// Eigen's), strict '<' in (shift, index) order -- every lane meets its candidates in that order, the lanes are merged by
// (norm, candidate number).  Twice the arithmetic of scan_hits; run for the queries a fast search has flagged, and by the
// stand-alone GetClosestNeighbor.  Returns the SQUARED distance of the chosen neighbour.
__device__ __forceinline__ double scan_hits_exact(const MapView &m, const Probe &pr, double sx, double sy, double sz, int lane, double nn[3]) {
    const unsigned long long ball = __ballot(pr.blk >= 0);
    unsigned hits = (unsigned)(ball >> (kicp_tid() & 32));
    double bestn = DBL_MAX, best2 = DBL_MAX;
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
            hits &= hits - 1;
            const int bj = __shfl(pr.blk, j & 31, 32);
            const int cj = __shfl(pr.cnt, j & 31, 32);
            cb[u] = __shfl(pr.offs, j & 31, 32);
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
                const double nr = sqrt(d);  // (p - q).norm()
                if (nr < bestn) {
                    bestn = nr;
                    best2 = d;
                    bx = xy[u].x;
                    by = xy[u].y;
                    bz = zz[u];
                    bkey = cb[u] + lane;
                }
            }
        }
    }
    double gn = bestn;
    int gkey = bkey, glane = lane;
    group_min(gn, gkey, glane);  // lexicographic (norm, candidate number)
    nn[0] = __shfl(bx, glane, 32);
    nn[1] = __shfl(by, glane, 32);
    nn[2] = __shfl(bz, glane, 32);
    return __shfl(best2, glane, 32);
}

// Same search for voxels that hold more than 32 points (max_points_per_voxel > 32): every probe
// lane strides over its voxel's points.  Rare configuration, kept simple.
__device__ __forceinline__ double scan_hits_wide(const MapView &m, const Probe &pr, double sx, double sy,
                                                 double sz, int lane, double nn[3]) {
    double best = DBL_MAX, best2 = DBL_MAX, bx = 0.0, by = 0.0, bz = 0.0;
    int bkey = 0x7FFFFFFF;
    if (pr.blk >= 0) {
        const double2 *xy = block_xy(m, pr.blk);
        const double *z = block_z(m, pr.blk);
        for (int k = 0; k < pr.cnt; ++k) {  // (the reference's comparison itself -- norms: see scan_hits_exact)
            const double dx = xy[k].x - sx, dy = xy[k].y - sy, dz = z[k] - sz;
            const double d = (dx * dx + dy * dy) + dz * dz;
            const double nr = sqrt(d);
            if (nr < best) {
                best = nr;
                best2 = d;
                bx = xy[k].x;
                by = xy[k].y;
                bz = z[k];
            }
        }
        bkey = lane;
    }
    int glane = lane;
    group_min(best, bkey, glane);
    nn[0] = __shfl(bx, glane, 32);
    nn[1] = __shfl(by, glane, 32);
    nn[2] = __shfl(bz, glane, 32);
    return __shfl(best2, glane, 32);
}

// ------------------------------------------------------------------------------------------
// The workgroup's voxel TILE.
//
// A workgroup of the ICP kernel serves a spatially compact run of source points.  The map voxels those
// points can reach are copied ONCE into the workgroup's LDS -- a small open-addressed table {relative voxel
// key -> first point, point count} over a store of xyz triples -- and every iteration's nearest-neighbour
// search runs against that tile: the map does not change during AlignPointsToMap, neighbouring queries share
// most of their voxels (so a voxel is fetched once per workgroup, not once per query), and nothing is staged
// per query.  What a query has established is its KNOWN WINDOW: the 27 voxels around the voxel it was in
// when it last looked, widened by one layer on every side it was close to (within kWindowMargin of the
// face) -- every occupied voxel of the window is in the tile, so as long as the query's current 27 voxels lie
// inside the window a table miss MEANS an empty voxel.  A query that leaves its window looks again (global
// probes for the new window's voxels; voxels already in the tile are not fetched again).
// ------------------------------------------------------------------------------------------
constexpr double kWindowMargin = 0.125;  // fraction of a voxel
constexpr int kFillChunk = 8;             // voxels whose points are in flight together during a fill
constexpr unsigned kTileEmpty = 0xFFFFFFFFu;
// table value: bits 0..23 first point in the LDS store, or (kTileGlobal) the voxel's block id in the map;
// bits 24..29 point count; bit 30 kTileGlobal; bit 31 kTileReady
constexpr unsigned kTileReady = 0x80000000u;   // the entry is complete (LDS voxels: the points are in the store)
constexpr unsigned kTileGlobal = 0x40000000u;  // the LDS store was full: the points are read from the map block in HBM / L2
constexpr unsigned kTileOverflow = 0xFFFFFFFEu;  // block id beyond 24 bits: queries that need this voxel search HBM
// relative voxel coordinates of a tile key: 11 bits for x and y, 10 for z -- with 0.1 m voxels 205 m x 205 m x 102 m, the
// whole reach of a 100 m sensor (10 bits per axis until round 4: far-field runs of the 1M-point configuration, a few
// hundred sparse points spread over more than 102 m, left the span and searched the map directly: 285 of 373 queries of
// one workgroup, 45 us per iteration, profiles/r04_f_icp_probe_livox.txt).  x stays below 2047, so no key is kTileEmpty.
constexpr int kTileSpanXY = 2048, kTileSpanZ = 1024;

struct Tile {
    unsigned *keys;  // [slots] relative voxel key (x 11 | y 11 | z 10 bits) or kTileEmpty
    unsigned *vals;  // [slots]
    int slots_mask;  // slots - 1 (2048 slots for runs of at most 64 points, else 4096)
    int hash_shift;  // 32 - log2(slots)
    int load_limit;  // entries beyond which the table counts as full (3/4)
    // One LDS region holds the points (24 bytes each, handed out from the bottom) AND the scan lists (16-bit
    // entries, handed out from the top): a workgroup whose neighbourhood is large uses all of it for points and
    // searches lane-per-voxel, one whose neighbourhood is small has room for the lists that make its searches
    // short.  Points are handed out in the fill phases, lists in the search phases, never at the same time.
    double *points;          // start of the region
    unsigned region_bytes;
    int cap_points;          // region_bytes / 24 (what fits when no list is kept)
    int *count;              // points asked for so far (the demand: keeps counting past what fits)
    int *stored;             // end of the points actually kept
    int *entries;            // occupied table slots
    unsigned short *lists;   // == (unsigned short *)points, indexed from the top; nullptr: this workgroup keeps no lists
    int list_top;            // region_bytes / 2
    int *list_count;         // list entries handed out so far
    int ox, oy, oz;          // voxel with relative coordinates (0, 0, 0)
    const double *far;       // a point at infinity in LDS (what a list's tail reads: tile_scan_list)
};
__device__ __forceinline__ int tile_ref(unsigned val) { return (int)(val & 0xFFFFFFu); }
__device__ __forceinline__ int tile_cnt(unsigned val) { return (int)((val >> 24) & 63u); }
__device__ __forceinline__ bool tile_rel(const Tile &t, int qx, int qy, int qz, unsigned &key) {
    const unsigned rx = (unsigned)(qx - t.ox), ry = (unsigned)(qy - t.oy), rz = (unsigned)(qz - t.oz);
    key = (rx << 21) | (ry << 10) | rz;
    return rx < (unsigned)(kTileSpanXY - 1) && ry < (unsigned)kTileSpanXY && rz < (unsigned)kTileSpanZ;
}
__device__ __forceinline__ unsigned long long tile_unrel(const Tile &t, unsigned key) {  // the map key of a relative key
    return pack_voxel(t.ox + (int)(key >> 21), t.oy + (int)((key >> 10) & 2047u), t.oz + (int)(key & 1023u));
}
__device__ __forceinline__ unsigned tile_hash(const Tile &t, unsigned key) { return (key * 0x9E3779B1u) >> t.hash_shift; }
static_assert(kIcpTileSlots == 4096, "hash_shift is derived from 4096 / 2048 slots");
// slot of a key or -1 (LDS loads; other waves may be inserting: relaxed workgroup-scope atomics keep the
// compiler from caching them)
constexpr int kTileMaxProbes = 32;  // the table is kept at most 3/4 full; a chain this long means "not here"
// (four slots of the chain per round trip, resolved in chain order: a miss in a table that is 3/4 full walks 8 slots on
// average, and the slowest of a group's 27 lanes is what a scan-list build waits for)
constexpr bool kScanPrefetch = true;  // tile_scan_list (a build-time switch for A/Bs; neutral to +0.5 %: profiles/r06_o_*)
constexpr int kTileProbeAhead = 4;
static_assert(kTileMaxProbes % kTileProbeAhead == 0, "tile_find's rounds");
__device__ __forceinline__ int tile_find(const Tile &t, unsigned key) {
    unsigned s = tile_hash(t, key);
    for (int probes = 0; probes < kTileMaxProbes; probes += kTileProbeAhead) {
        unsigned k[kTileProbeAhead];
#pragma unroll
        for (int u = 0; u < kTileProbeAhead; ++u)
            k[u] = __hip_atomic_load(&t.keys[(s + (unsigned)u) & (unsigned)t.slots_mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int res = -2;  // undecided
#pragma unroll
        for (int u = 0; u < kTileProbeAhead; ++u) {
            const int here = k[u] == key ? (int)((s + (unsigned)u) & (unsigned)t.slots_mask) : (k[u] == kTileEmpty ? -1 : -2);
            res = res == -2 ? here : res;
        }
        if (res != -2) return res;
        s = (s + (unsigned)kTileProbeAhead) & (unsigned)t.slots_mask;
    }
    return -1;
}

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

// minimum of (distance, key) over the 32 lanes of a group, lexicographic; every lane ends with the winner.
// Two all-reduces (DPP row operations + one swizzle each): the distance first, then the key among the lanes
// that hold that distance.
__device__ __forceinline__ void group_min_dist_key(double &best, int &key, double sec = DBL_MAX, bool *tie = nullptr) {
    double g = best;
    group_fmin_step<0>(g);
    group_fmin_step<1>(g);
    group_fmin_step<2>(g);
    group_fmin_step<3>(g);
    group_fmin_step<4>(g);
    if (tie) *tie = group_norm_tie(best, g, sec);
    int k = (best == g) ? key : 0x7FFFFFFF;
    group_imin_step<0>(k);
    group_imin_step<1>(k);
    group_imin_step<2>(k);
    group_imin_step<3>(k);
    group_imin_step<4>(k);
    best = g;
    key = k;
}

// Establish the known window of the query s (voxel v): make sure every occupied voxel of the window is in
// the tile.  Lane l answers for the window cells l and l + 32.  Returns false when the query cannot use the
// tile (a voxel outside the tile's coordinate span, table or store full): it then searches HBM directly.
// (Meta: IcpQueryMeta, or the 20-byte record of the thread-per-query variant -- v, lo, hi, valid, list_state)
template <class Meta>
__device__ __forceinline__ bool tile_fill(const MapView &m, const Tile &tile, const double s[3], const int v[3], int lane,
                                          Meta *meta, int &range_err) {
    int lo[3], nn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f = s[a] / m.voxel_size - (double)v[a];  // position inside the voxel, [0, 1)
        lo[a] = (f < kWindowMargin) ? -2 : -1;
        const int hi = (f > 1.0 - kWindowMargin) ? 2 : 1;
        nn[a] = hi - lo[a] + 1;
    }
    const int W = nn[0] * nn[1] * nn[2];  // <= 64
    bool need[2];             // this cell must be looked up in the map (HBM)
    unsigned long long key[2];
    unsigned rkey[2];
    bool fail = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int w = lane + 32 * h;
        need[h] = false;
        key[h] = 0;
        rkey[h] = 0;
        if (w < W) {
            const int iz = w % nn[2], t = w / nn[2], iy = t % nn[1], ix = t / nn[1];
            const int ox = lo[0] + ix, oy = lo[1] + iy, oz = lo[2] + iz;
            const int qx = v[0] + ox, qy = v[1] + oy, qz = v[2] + oz;
            if (voxel_in_range(qx, qy, qz)) {
                if (!tile_rel(tile, qx, qy, qz, rkey[h])) {
                    fail = true;  // outside the span of the relative keys
                } else {
                    const int slot = tile_find(tile, rkey[h]);
                    if (slot < 0) {
                        need[h] = true;
                        key[h] = pack_voxel(qx, qy, qz);
                    } else if (__hip_atomic_load(&tile.vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kTileOverflow) {
                        fail = true;
                    }
                }
            } else if (ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1) {
                range_err = 1;
            }
        }
    }
    const int half_shift = kicp_tid() & 32;
    if ((unsigned)(__ballot(fail) >> half_shift) != 0u) {
        if (lane == 0) meta->valid = -1;  // do not try again
        return false;
    }
    // the voxels the tile does not know yet: one map lookup each, both of a lane's in flight together
    int blk[2], cnt[2];
    map_find_pair(m, need[0], key[0], need[1], key[1], blk[0], cnt[0], blk[1], cnt[1]);
    // occupied ones enter the table; whoever claims the slot also fetches the points (another query of this
    // workgroup may be asking for the same voxel at the same moment)
    bool won[2];
    int slot[2], off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        won[h] = false;
        slot[h] = -1;
        off[h] = 0;
        if (need[h] && blk[h] >= 0 && cnt[h] > 0) {
            unsigned sidx = tile_hash(tile, rkey[h]);
            bool placed = false;
            // (a chain longer than tile_find follows, or a table more than 3/4 full, counts as "table full")
            const bool room = __hip_atomic_load(tile.entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tile.load_limit;
            for (int probes = 0; room && probes < kTileMaxProbes; ++probes) {
                const unsigned old = atomicCAS(&tile.keys[sidx], kTileEmpty, rkey[h]);
                if (old == kTileEmpty) {
                    atomicAdd(tile.entries, 1);
                    won[h] = true;
                    slot[h] = (int)sidx;
                    placed = true;
                    break;
                }
                if (old == rkey[h]) {
                    placed = true;  // somebody else is bringing it in
                    break;
                }
                sidx = (sidx + 1) & (unsigned)tile.slots_mask;
            }
            if (!placed) fail = true;  // table full
            if (won[h]) {
                off[h] = atomicAdd(tile.count, cnt[h]);
                // (no list is handed out during a fill phase: the list counter stands still)
                const unsigned lists_bytes = tile.lists ? 2u * (unsigned)__hip_atomic_load(tile.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                const bool fits = (unsigned)(off[h] + cnt[h]) * 24u + min(lists_bytes, tile.region_bytes) <= tile.region_bytes && off[h] + cnt[h] <= 0xFFFF;
                if (fits) atomicMax(tile.stored, off[h] + cnt[h]);
                if (!fits) {
                    // LDS store full: the table remembers where the voxel is in the map instead (the lookup is
                    // saved, the points are read from HBM / L2 at every search)
                    const unsigned val = (unsigned)blk[h] < 0x1000000u ? ((unsigned)blk[h] | ((unsigned)cnt[h] << 24) | kTileGlobal | kTileReady)
                                                                       : kTileOverflow;
                    __hip_atomic_store(&tile.vals[slot[h]], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    won[h] = false;
                    if (val == kTileOverflow) fail = true;
                }
            }
        }
    }
    // the group copies the voxels its lanes have won: lane i fetches point i (one 16-byte + one 8-byte load,
    // coalesced), kFillChunk voxels in flight
    unsigned long long todo = (unsigned long long)(unsigned)(__ballot(won[0]) >> half_shift) |
                              ((unsigned long long)(unsigned)(__ballot(won[1]) >> half_shift) << 32);
    while (todo) {
        double2 xy[kFillChunk];
        double zz[kFillChunk];
        int dst[kFillChunk];
#pragma unroll
        for (int u = 0; u < kFillChunk; ++u) {
            dst[u] = -1;
            if (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int src = j & 31;
                const int b0 = __shfl(blk[0], src, 32), b1 = __shfl(blk[1], src, 32);
                const int c0 = __shfl(cnt[0], src, 32), c1 = __shfl(cnt[1], src, 32);
                const int o0 = __shfl(off[0], src, 32), o1 = __shfl(off[1], src, 32);
                const int bj = j < 32 ? b0 : b1, cj = j < 32 ? c0 : c1, oj = j < 32 ? o0 : o1;
                if (lane < cj) {
                    dst[u] = oj + lane;
                    xy[u] = block_xy(m, bj)[lane];
                    zz[u] = block_z(m, bj)[lane];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kFillChunk; ++u) {
            if (dst[u] >= 0) {
                double *q = tile.points + 3 * dst[u];
                q[0] = xy[u].x;
                q[1] = xy[u].y;
                q[2] = zz[u];
            }
        }
    }
    group_lds_sync();  // the points are in the store before their table entries say so
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (won[h])
            __hip_atomic_store(&tile.vals[slot[h]], (unsigned)off[h] | ((unsigned)cnt[h] << 24) | kTileReady, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool failed = (unsigned)(__ballot(fail) >> half_shift) != 0u;
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
        meta->valid = failed ? -1 : 1;
        meta->list_state = 0;  // whatever list there was belongs to the old window
    }
    group_lds_sync();
    return !failed;
}

// GetClosestNeighbor against the tile: lane j < 27 takes the j-th voxel of the reference's shift table
// (VoxelHashMap.cpp:35-41), finds it in the table (a miss is an empty voxel: the query is inside its known
// window) and walks its points in order.  A lane meets its points in the reference's order, so strict '<'
// keeps the first minimum inside the voxel; across lanes the smaller shift position wins among equal
// distances -- the reference's nested strict '<' loops.  The work per query is bounded by the fullest
// voxel (max_points_per_voxel steps), not by the size of the neighbourhood.
// Returns the squared distance (DBL_MAX: no candidate), the neighbour, the number of points examined;
// bad = the tile cannot answer: 1 a voxel is still being fetched by a concurrent fill, 2 one did not fit.
// second (want_second): the second smallest squared distance over all 27 cells, as tile_scan_list returns it (the stability test's
// bound, IcpQueryMeta::Lr) -- every cell is walked here, so the runner-up of what was met IS the runner-up of the neighbourhood.
__device__ __forceinline__ double tile_scan(const MapView &m, const Tile &tile, double sx, double sy, double sz, int vx, int vy, int vz,
                                            int lane, double nn[3], int &examined, int &bad, bool *tie = nullptr, bool want_second = false,
                                            double *second = nullptr) {
    constexpr int U = 4;
    // the smallest squared distance this lane has met and not kept: its runner-up.  (The norm-tie detection needs less -- only a
    // candidate that came BEFORE the best can have the best's norm and lose -- but the full runner-up costs one more minimum per
    // candidate and flags a superset of the ties: re-searched exactly, once in 1e10.)
    double prev = DBL_MAX;
    int ref = 0, cnt = 0;
    int mybad = 0;
    bool glob = false;
    if (lane < 27) {
        const int qx = vx + (int)((kShift.x >> (2 * lane)) & 3) - 1;
        const int qy = vy + (int)((kShift.y >> (2 * lane)) & 3) - 1;
        const int qz = vz + (int)((kShift.z >> (2 * lane)) & 3) - 1;
        unsigned rkey;
        if (tile_rel(tile, qx, qy, qz, rkey)) {
            const int slot = tile_find(tile, rkey);
            if (slot >= 0) {
                const unsigned val = __hip_atomic_load(&tile.vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (val == kTileOverflow) {
                    mybad = 2;
                } else if (!(val & kTileReady)) {
                    mybad = 1;
                } else {
                    ref = tile_ref(val);
                    cnt = tile_cnt(val);
                    glob = (val & kTileGlobal) != 0u;
                }
            }
        } else {
            mybad = 2;
        }
    }
    const int half_shift = kicp_tid() & 32;
    bad = (unsigned)(__ballot(mybad == 2) >> half_shift) != 0u ? 2 : ((unsigned)(__ballot(mybad == 1) >> half_shift) != 0u ? 1 : 0);
    if (bad) cnt = 0;
    int tot = cnt;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 32);
    examined = tot;
    // ---- voxels held in LDS: every lane walks its own voxel
    double best = DBL_MAX;
    int bk = 0;
    {
        const double *P = tile.points + 3 * (glob ? 0 : ref);
        const int c = glob ? 0 : cnt;
        for (int k0 = 0; __ballot(k0 < c) != 0ull; k0 += U) {  // wave-uniform trip count
            double x[U], y[U], z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double *q = P + 3 * ((k0 + u < c) ? k0 + u : 0);
                x[u] = q[0];
                y[u] = q[1];
                z[u] = q[2];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double ex = x[u] - sx, ey = y[u] - sy, ez = z[u] - sz;
                const double d = (ex * ex + ey * ey) + ez * ez;
                const bool in = k0 + u < c;
                const bool take = in & (d < best);
                const double loser = take ? best : d;  // (what is not the best after this point)
                prev = (in & (loser < prev)) ? loser : prev;
                best = take ? d : best;
                bk = take ? k0 + u : bk;
            }
        }
    }
    int key = (!glob && cnt > 0 && best < DBL_MAX) ? ((lane << 5) | bk) : 0x7FFFFFFF;
    double bx = 0.0, by = 0.0, bz = 0.0;
    if (key != 0x7FFFFFFF) {
        const double *q = tile.points + 3 * (ref + bk);
        bx = q[0];
        by = q[1];
        bz = q[2];
    }
    // ---- voxels left in the map (HBM / L2): the group reads them together, lane i point i, kChunk voxels in
    // flight (one memory round trip per kChunk voxels instead of one per point)
    unsigned gl = (unsigned)(__ballot(glob && cnt > 0) >> half_shift);
    while (__ballot(gl != 0) != 0ull) {  // wave-uniform trip count
        double2 xy[kChunk];
        double zz[kChunk];
        int kj[kChunk];
        bool ld[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int j = gl ? (__ffs(gl) - 1) : -1;
            gl &= gl - 1;  // (0 & -1) == 0
            const int bj = __shfl(ref, j & 31, 32);
            const int cj = __shfl(cnt, j & 31, 32);
            kj[u] = j;
            ld[u] = (j >= 0) && (lane < cj);
            if (ld[u]) {
                xy[u] = block_xy(m, bj)[lane];
                zz[u] = block_z(m, bj)[lane];
            }
        }
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            if (ld[u]) {
                const double ex = xy[u].x - sx, ey = xy[u].y - sy, ez = zz[u] - sz;
                const double d = (ex * ex + ey * ey) + ez * ez;
                const int k = (kj[u] << 5) | lane;  // {shift position of the voxel, index inside it}
                if (d < best || (d == best && k < key)) {
                    prev = best < prev ? best : prev;
                    best = d;
                    key = k;
                    bx = xy[u].x;
                    by = xy[u].y;
                    bz = zz[u];
                } else if (d < prev) {
                    prev = d;
                }
            }
        }
    }
    if (best == DBL_MAX) key = 0x7FFFFFFF;
    const int mykey = key;
    const double mybest = best;
    group_min_dist_key(best, key, prev, tie);
    const bool found = key != 0x7FFFFFFF;
    // the lane that holds the winner hands its coordinates to the group
    const unsigned who = (unsigned)(__ballot(found && mykey == key) >> half_shift);
    const int wl = who ? (__ffs(who) - 1) : 0;
    if (want_second) {
        double g2 = (found && lane == wl) ? prev : mybest;
        group_fmin_step<0>(g2);
        group_fmin_step<1>(g2);
        group_fmin_step<2>(g2);
        group_fmin_step<3>(g2);
        group_fmin_step<4>(g2);
        *second = g2;
    }
    nn[0] = __shfl(bx, wl, 32);
    nn[1] = __shfl(by, wl, 32);
    nn[2] = __shfl(bz, wl, 32);
    return best;
}

// Scan lists (workgroups with few points, i.e. full-size voxels and long neighbourhoods): the positions, in
// the LDS store, of the points of a query's 27 voxels in exactly the order the reference visits them, 16 bits
// each.  The search then strides over the list with all 32 lanes -- ~3x fewer steps than one lane per voxel
// when voxels hold 20 points -- and strict '<' plus "smaller list position wins" reproduces the reference's
// tie rules.  Built from the table (no HBM access) whenever the query enters another voxel.  Returns false
// when the query cannot have a list (a voxel in the extension or missing from the tile, pool exhausted).
// cur_base / cur_cap: the record's list_base / list_cap as the caller has read them; out_base / out_n: where the new list lies and
// its length (what the record holds afterwards), for a scan that follows at once.
__device__ __forceinline__ bool tile_list_build(const Tile &tile, int vx, int vy, int vz, int lane, IcpQueryMeta *meta, int cur_base, int cur_cap,
                                                int &out_base, int &out_n) {
    int off = 0, cnt = 0;
    bool mybad = false;
    if (lane < 27) {
        const int qx = vx + (int)((kShift.x >> (2 * lane)) & 3) - 1;
        const int qy = vy + (int)((kShift.y >> (2 * lane)) & 3) - 1;
        const int qz = vz + (int)((kShift.z >> (2 * lane)) & 3) - 1;
        unsigned rkey;
        if (tile_rel(tile, qx, qy, qz, rkey)) {
            const int slot = tile_find(tile, rkey);
            if (slot >= 0) {
                const unsigned val = __hip_atomic_load(&tile.vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (val == kTileOverflow || (val & kTileGlobal) || !(val & kTileReady)) mybad = true;
                off = tile_ref(val);
                cnt = tile_cnt(val);
            }
        } else {
            mybad = true;
        }
    }
    const int half_shift = kicp_tid() & 32;
    const int incl = kListScanDpp ? group_incl_scan(cnt) : [&] {
        int v = cnt;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int up = __shfl_up(v, o, 32);
            if (lane >= o) v += up;
        }
        return v;
    }();
    const int total = __shfl(incl, 31, 32);
    bool fail = (unsigned)(__ballot(mybad) >> half_shift) != 0u || total > 0x7FFE;  // (tile_scan_list's keys: 15 bits of list position)
    int base = cur_base;
    if (!fail && total > cur_cap) {  // a longer list than before: new room from the pool (the old one is abandoned)
        int nb = -1;
        const int want = total + 16;
        if (lane == 0) {
            // from the top of the region down, as long as the list stays clear of the points kept below (no point
            // is handed out during a search phase)
            const int used = atomicAdd(tile.list_count, want);
            const int start = tile.list_top - used - want;
            const int stored = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nb = (start >= 0 && (long)start * 2 >= (long)stored * 24) ? start : -1;
        }
        nb = __shfl(nb, 0, 32);
        if (nb < 0) {
            fail = true;
        } else {
            base = nb;
            if (lane == 0) {
                meta->list_base = nb;
                meta->list_cap = (unsigned short)want;
            }
        }
    }
    if (fail) {
        if (lane == 0) meta->list_state = -1;
        group_lds_sync();
        return false;
    }
    unsigned short *I = tile.lists + base + (incl - cnt);
    for (int i = 0; i < cnt; ++i) I[i] = (unsigned short)(off + i);
    if (lane == 0) {
        meta->lv[0] = vx;
        meta->lv[1] = vy;
        meta->lv[2] = vz;
        meta->list_n = (unsigned short)total;
        meta->list_state = 1;
    }
    out_base = base;
    out_n = total;
    group_lds_sync();
    return true;
}

// GetClosestNeighbor over a scan list: 32 lanes stride over it, four candidates per lane in flight per trip,
// no divergent control flow.  Returns the squared distance (DBL_MAX: no candidate) and the neighbour.
// second (when given): the SECOND smallest squared distance over the whole list (+inf / DBL_MAX: the list holds one candidate /
// none) -- the winner's lane contributes the runner-up of what it has met, every other lane its best; a candidate that ties
// with the winner counts as a runner-up.  What the stability test of the next iterations needs (IcpQueryMeta::Lr).
__device__ __forceinline__ double tile_scan_list(const Tile &tile, const unsigned short *I, int n, double sx, double sy, double sz,
                                                 int lane, double nn[3], bool *tie = nullptr, bool want_second = false, double *second = nullptr) {
    constexpr int U = 4;
    const double *P = tile.points;
    // A lane keeps its smallest squared distance AND its second smallest (every candidate that is not kept goes into it): the
    // second is what tells a tie in norm (kicp_search.hpp, kNormTie) -- and costs nothing: v_min / v_max / v_min instead of
    // the two selects of a conditional move, once the list's tail reads a point at infinity (tile.far) instead of being
    // masked out of the comparison.
    double best = DBL_MAX, sec = DBL_MAX;
    int bi = 0x7FFFFFFF;
    // (kScanPrefetch: a trip's list entries are asked for during the trip before it -- one LDS round trip per trip instead of two)
    int pos[U];
    if (kScanPrefetch) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = lane + 32 * u;
            pos[u] = (int)I[i < n ? i : 0];
        }
    }
    for (int i0 = lane; __ballot(i0 < n) != 0ull; i0 += 32 * U) {  // wave-uniform trip count
        if (!kScanPrefetch) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 32 * u;
                pos[u] = (int)I[i < n ? i : 0];
            }
        }
        double x[U], y[U], z[U];
        int cur[U];  // (this trip's store positions: they ride in the low half of a candidate's key, below)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 32 * u;
            cur[u] = pos[u];
            const double *q = i < n ? P + 3 * KICP_IDX((BoundsRec *)nullptr, (int *)nullptr, pos[u], tile.cap_points, 9) : tile.far;
            x[u] = q[0];
            y[u] = q[1];
            z[u] = q[2];
        }
        if (kScanPrefetch) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 32 * U + 32 * u;
                pos[u] = (int)I[i < n ? i : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 32 * u;
            const double ex = x[u] - sx, ey = y[u] - sy, ez = z[u] - sz;
            const double d = (ex * ex + ey * ey) + ez * ez;  // (+inf past the end of the list)
            // strict: a lane meets its candidates in list order and keeps the first.  The key is {list position, store position}:
            // ordered by the list position (the reference's visiting order decides among equal distances), and the winner's
            // coordinates are then one LDS round trip away instead of two (list entry, then point)
            bi = d < best ? ((i << 16) | cur[u]) : bi;
            sec = __builtin_fmin(sec, __builtin_fmax(best, d));
            best = __builtin_fmin(best, d);
        }
    }
    const double mybest = best;
    const int mybi = bi;
    group_min_dist_key(best, bi, sec, tie);
    const bool found = bi != 0x7FFFFFFF;
    if (want_second) {
        double g2 = (found && mybi == bi) ? sec : mybest;
        group_fmin_step<0>(g2);
        group_fmin_step<1>(g2);
        group_fmin_step<2>(g2);
        group_fmin_step<3>(g2);
        group_fmin_step<4>(g2);
        *second = g2;
    }
    const int p = found ? (bi & 0xFFFF) : 0;
    nn[0] = found ? P[3 * p] : 0.0;
    nn[1] = found ? P[3 * p + 1] : 0.0;
    nn[2] = found ? P[3 * p + 2] : 0.0;
    return best;
}

__device__ __forceinline__ double closest_neighbor_any(const MapView &m, double sx, double sy, double sz,
                                                       int lane, double nn[3], int &examined, int &range_err) {
    const Probe pr = probe27(m, sx, sy, sz, lane, range_err);
    examined = pr.E;
    if (m.max_points > 32) return scan_hits_wide(m, pr, sx, sy, sz, lane, nn);
    bool tie = false;
    const double d2 = scan_hits<false>(m, pr, sx, sy, sz, lane, nn, nullptr, 0, &tie);
    if (!tie) return d2;
    return scan_hits_exact(m, pr, sx, sy, sz, lane, nn);  // (a tie in norm: the reference's own comparison decides)
}
// ... with the reference's comparison throughout
__device__ __forceinline__ double closest_neighbor_exact(const MapView &m, double sx, double sy, double sz, int lane, double nn[3], int &examined,
                                                         int &range_err) {
    const Probe pr = probe27(m, sx, sy, sz, lane, range_err);
    examined = pr.E;
    if (m.max_points > 32) return scan_hits_wide(m, pr, sx, sy, sz, lane, nn);
    return scan_hits_exact(m, pr, sx, sy, sz, lane, nn);
}

}  // namespace kicp
