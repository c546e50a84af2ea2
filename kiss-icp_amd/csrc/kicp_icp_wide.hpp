// kicp_icp_wide.hpp -- the association phases of k_icp<PROF, WIDE = true>: one THREAD per source point.
//   DataAssociation                         core/Registration.cpp:60-78
//   VoxelHashMap::GetClosestNeighbor        core/VoxelHashMap.cpp:46-70 (shift table :35-41)
// Included by kicp_icp.hip (after IcpShared); nothing else includes it.
//
// k_icp's first form gives every query a 32-lane group: right when a workgroup has a few dozen queries whose
// neighbourhoods hold hundreds of points (full-size voxels).  With small voxels and a large cloud (the 1M-point /
// 0.1 m configuration: ~400 queries per workgroup, 27 voxels of one to five points each) a group spends its time in
// dependent LDS / L2 round trips with two or three of its 32 lanes busy, 16 queries in flight per CU:
// 2 .. 7 us per query and group, 171 us per iteration (profiles/r03_ac_icp_probe_livox.txt).  Here a THREAD owns a
// query for the whole launch -- 512 queries in flight per CU, the running source point, the last neighbour and the
// verdicts in registers -- and the search is arranged so that lanes of a wave rarely wait for each other:
//   1. the 27 table lookups of a query are unrolled: the first two probe slots of all 27 chains are loaded together
//      (54 independent LDS loads), values of the hits together (27 more); what is still open after two probes (a
//      few per cent) is finished by one short loop.  Result: which cells are occupied (LDS / map resident), and E,
//      the number of points the reference examines (VoxelHashMap.cpp:58-61) -- no point has been read yet;
//   2. a voxel that CANNOT hold the answer is never read: the squared distance from the query to the voxel's box is a
//      lower bound of every distance the reference would compute there (built from the same operations, see
//      wide_gaps), and a voxel whose bound exceeds what is already in hand loses every strict '<' of
//      VoxelHashMap.cpp:58-63.  "In hand" = the best of this search, the correspondence threshold (a neighbour beyond it
//      is dropped by Registration.cpp:72 whatever it is), and -- from the second iteration on -- the distance to the
//      PREVIOUS iteration's neighbour when that point is still inside the 27 voxels (the map does not change during
//      AlignPointsToMap): a converged query reads one voxel or two instead of fifteen.  The result (neighbour, distance,
//      ties) is the reference's, bit for bit: reading fewer voxels only removes comparisons that are lost anyway, and
//      the order among equals is kept by the key {shift position, index in the voxel} exactly as in tile_scan;
//   3. the thread walks voxels itself while that is cheap: the first occupied one in shift order (its own voxel, else
//      a face neighbour) always -- it gives a distance to skip by --, then at most four voxels / 24 points in all.  What
//      survives beyond that (in dense surroundings a query that floats beside a surface keeps ten voxels of twenty
//      points; its 63 neighbours in the wave would wait for it: 52 us per iteration in such workgroups,
//      profiles/r04_d_icp_probe_livox.txt) is filed as {query, voxel} items in two queues in LDS -- voxels in the LDS
//      store, and voxels the store had no room for (kTileGlobal: read from the map in HBM / L2) -- and served by the
//      whole workgroup, a thread per POINT of all queued voxels (wide_serve_flat; a 32-lane group per item, six in
//      flight, was 22 us per round); the owners merge the answers;
//   4. most iterations most queries need NO search: the last search left a lower bound of the distance to every point
//      but the neighbour (the runner-up, or the bound of a cell that was not read), the query has moved by so much
//      since, and while the neighbour's new distance stays strictly below the difference it is still the reference's
//      answer (WideQuery::Lr).  The 10 - 30 queries of a workgroup that do need one are then searched by a 32-lane
//      group each, with the same skipping (wide_group_scan): the queues' fixed costs are for the first iteration or two.
// Queries the tile cannot serve (outside the key span, table full) and queries of runs longer than one chunk go
// through a small queue served by the 32-lane groups with the map-direct search of the first form (closest_neighbor_any).
// The partition of the cloud (runs), the order in which products are added (phase C) and the exchange are those of
// the first form, so both forms give the same sums and the same pose bit for bit (tests/test_gpu_paths.py::
// test_thread_per_query_form_is_bitwise_the_group_form): which form a launch uses is a matter of speed only.
#pragma once

namespace kicp {

constexpr int kWideChunk = kIcpThreads;  // queries a workgroup carries through an iteration at a time: one per thread
constexpr int kWideQueue = kIcpChunk;    // records of the slow-path queue (sh.pts)
constexpr int kWideRoundLimit = 1 << 14; // rounds of the item queues a chunk may take (512 queries x 27 cells, at least one item per round)
constexpr int kWideTermRows = 128;       // phase C: products of this many points in LDS together (sh.terms continued into sh.pts)
static_assert(offsetof(IcpShared, pts) == offsetof(IcpShared, terms) + sizeof(double) * kIcpTermChunk * kIcpTerms,
              "phase C of the thread-per-query form uses terms and pts as one array");
static_assert(sizeof(double) * kWideTermRows * kIcpTerms <= sizeof(double) * kIcpTermChunk * kIcpTerms + sizeof(IcpPoint) * kIcpChunk, "term rows");
static_assert(kWideTermRows % kIcpGroupsPerBlock == 0 && kIcpTermChunk % kIcpGroupsPerBlock == 0 && kIcpChunk % kIcpGroupsPerBlock == 0,
              "the order of additions (points j = g, g + 16, ... per adder) must not depend on the chunking");
constexpr int kWideJobs = (int)((sizeof(double) * kIcpTermChunk * kIcpTerms + sizeof(IcpPoint) * kIcpChunk) / 8);  // point-fetch jobs of the window phase

// LDS record of a query: its known window (kicp_search.hpp).  The running point, the last neighbour and the flags
// live in the owning thread's registers.
struct WideMeta {
    int v[3];                  // voxel the known window is centred on
    signed char lo[3], hi[3];  // extent of the window per axis, in voxels relative to v
    signed char valid;         // 1: every occupied voxel of the window is in the tile; 0: not looked yet; -1: cannot use the tile
    signed char list_state;    // 2 while the query takes part in a workgroup-wide window phase (otherwise unused)
};
static_assert(sizeof(WideMeta) == 20, "WideMeta layout");

struct WideQuery {  // registers of the owning thread, from iteration to iteration
    double s[3];   // transformed source point
    double nn[3];  // closest map point (have_nn)
    double d2;     // its squared distance (DBL_MAX: no neighbour)
    int v[3];      // voxel of s
    int E;         // map points the reference examines for this query
    int flag;      // 0 window valid, 1 window must be (re)established, 2 map-direct search
    bool have_nn;  // nn comes from a tile search of this launch, made while the query was in voxel v
    // OCCUPANCY: which of the 27 cells around v are occupied (and E, their population): what the table lookups of a search
    // find out.  It stays true as long as the query stays in its voxel (the map does not change during
    // AlignPointsToMap), so a query that has not left its voxel since its last search skips the lookups -- and a wave
    // all of whose queries stayed skips that part of the code altogether.  occ_valid: occ / E / nn belong to voxel v.
    unsigned occ;
    bool occ_valid;
    // STABILITY.  Lr: a lower bound (a distance, not squared, shaved by 2^-30) of the distance from the query to every map
    // point of its 27 cells EXCEPT nn -- the second smallest distance the last full search computed, or the smallest
    // box bound of a cell it did not read.  The query moves by |s' - s| per iteration, so every one of those points stays
    // at least Lr - |s' - s| away (triangle inequality); as long as the query stays in its voxel (same 27 cells) and nn's
    // new distance is strictly below that, nn is still the unique minimum the reference's strict '<' loops would find
    // (VoxelHashMap.cpp:55-63) and E is unchanged: the iteration needs no lookup and no point -- registers only.
    // Anything else (voxel left, margin used up, no neighbour yet) takes the full search, which renews Lr.
    double Lr;
    bool lr_valid;
};

// one full search: what goes in, what comes out (the neighbour itself comes back in WideBest)
struct WideJob {
    double s[3];
    int v[3];
    unsigned occ;  // in (cached) / out
    int E;         // in (cached) / out
    bool cached;   // in: occ / E are valid for v
    double d2, Lr; // out
    int bkey;      // out: {shift position, index} of the neighbour
};

// a query on its way to / from the lanes that run the full searches of an iteration (compaction: a few per cent of a
// workgroup's queries need one, a few iterations in; gathered into the first lanes they cost one wave instead of all)
struct WideRec {
    double s[3];    // in: the query;            out: nn
    double limit;   // in: limit0;               out: d2
    int v[3];       // in: its voxel;            out: {bkey, -, bad}
    unsigned occ;   // in / out: occupancy of the 27 cells (in: when `cached`)
    int E;          // in / out: their population
    int cached;     // in: occ / E are valid for v
    double Lr;      // out
};
static_assert(sizeof(WideRec) == 64, "WideRec layout");
static_assert(offsetof(IcpShared, range_sum) == offsetof(IcpShared, part) + sizeof(double) * kIcpGroupsPerBlock * kIcpSums, "the records use part and range_sum as one array");
constexpr int kWideRecsRoom = (int)((sizeof(double) * kIcpGroupsPerBlock * kIcpSums + sizeof(double) * kIcpSumRows * kIcpSums) / sizeof(WideRec));
constexpr int kWideRecs = kWideRecsRoom < 128 ? kWideRecsRoom : 128;  // (16 lanes of each of the 8 waves take one)
constexpr int kWideGroupRecs = (int)((sizeof(double) * kIcpTermChunk * kIcpTerms + sizeof(IcpPoint) * kIcpChunk) / sizeof(WideRec));  // records of the searches by groups (sh.terms + sh.pts)

// ---- lower bounds of the distances to the neighbouring voxel layers --------------------------------------------------
// A point p stored in voxel c satisfies floor(fl(p / vs)) == c (PointToVoxel, VoxelUtils.hpp:33-37), hence
// c vs (1 - 2^-52) <= p < (c + 1) vs (1 + 2^-52) in real arithmetic.  For the layer above the query's voxel v:
// p - s >= (v + 1) vs - s - slack, for the layer below: s - p > s - v vs - slack, the slack (2^-48 of the magnitudes
// involved) covering the rounding of the products, of the differences and of the voxel assignment at the face.  The
// scan computes ex = fl(p - s) and d = fl(fl(fl(ex^2) + fl(ey^2)) + fl(ez^2)); rounding is monotone, so with
// g <= |ex| the same expression over the g's is <= d.  m2 / p2: squared bounds for the layers v - 1 / v + 1.
struct WideGaps {
    double m2[3], p2[3];
};
__device__ __forceinline__ WideGaps wide_gaps(const double s[3], const int v[3], double vs) {
    WideGaps g;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f0 = (double)v[a] * vs, f1 = (double)(v[a] + 1) * vs;
        const double slack = (fabs(f0) + fabs(f1) + fabs(s[a])) * 0x1p-48 + DBL_MIN;
        double gm = (s[a] - f0) - slack, gp = (f1 - s[a]) - slack;
        gm = gm > 0.0 ? gm : 0.0;
        gp = gp > 0.0 ? gp : 0.0;
        g.m2[a] = gm * gm;
        g.p2[a] = gp * gp;
    }
    return g;
}
// which of the 27 cells can still matter: bound <= limit (sums of three of the nine squared gaps, picked at compile time).
// The reference compares norms (kicp_search.hpp): a cell is given up only when its bound lies beyond kNormTie of the limit.
__device__ __forceinline__ unsigned wide_keep_mask(const WideGaps &gaps_in, double limit) {
    unsigned keep = 0u;
    // (the six squared gaps go through empty asm statements: the 27 sums below depend on the query only, so the compiler hoisted
    // them out of every loop this is called in and kept 54 registers alive across the searches and the queue rounds -- a third
    // of the kernel's spills in round 5.  Recomputed per call they are 54 additions.)
    WideGaps gaps = gaps_in;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        asm volatile("" : "+v"(gaps.m2[a]));
        asm volatile("" : "+v"(gaps.p2[a]));
    }
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        const int cx = (int)((kShift.x >> (2 * j)) & 3), cy = (int)((kShift.y >> (2 * j)) & 3), cz = (int)((kShift.z >> (2 * j)) & 3);
        const double bx = cx == 0 ? gaps.m2[0] : (cx == 2 ? gaps.p2[0] : 0.0);
        const double by = cy == 0 ? gaps.m2[1] : (cy == 2 ? gaps.p2[1] : 0.0);
        const double bz = cz == 0 ? gaps.m2[2] : (cz == 2 ? gaps.p2[2] : 0.0);
        if (!((bx + by) + bz > limit * kNormTie)) keep |= 1u << j;  // (kNormTie: a cell just beyond the limit may still hold a point of the same NORM)
    }
    return keep;
}

struct WideCounters {  // profiling build
    unsigned visited_lds, visited_map;
    unsigned t_lookup, t_chains, t_walk;  // 10 ns ticks: the unrolled lookups, chains longer than two slots, the voxels in LDS
};

// a map-resident voxel some query still has to look at (the queue lives in sh.terms + sh.pts)
struct WideItem {
    double s[3];       // in: the query; out: the voxel's point closest to it
    double d2;         // out: its squared distance
    unsigned blk_cnt;  // in: block id (LDS queue: position in the LDS store) | point count << 24; out: the voxel's SECOND smallest squared
                       // distance as float bits, rounded down (FLT_MAX: the voxel has one point) -- a bound, for the stability test
    unsigned short slot;  // the voxel's slot in the tile's table
    unsigned char j, k;   // shift position of the voxel; out: index of the point in it
};
static_assert(sizeof(WideItem) == 40, "WideItem layout");
constexpr int kWideItems = (int)((sizeof(double) * kIcpTermChunk * kIcpTerms + sizeof(IcpPoint) * kIcpChunk) / sizeof(WideItem));
// Two queues in that memory: voxels in the LDS store (a trip of the serving loop is an LDS round trip) and voxels in the
// map (a trip is an HBM / L2 round trip) -- mixed, every trip of every group waited for a map voxel
// (17 us per round of 486 items, profiles/r04_f_icp_probe_livox.txt).
constexpr int kWideItemsLds = 120, kWideItemsMap = kWideItems - kWideItemsLds;
// what a thread walks itself before it leaves the rest of its voxels to the queue (the first voxel is always walked)
constexpr int kWideWalkVoxels = 4, kWideWalkPoints = 24;

struct WideBest {  // a search in progress (between the LDS part and the map part)
    double best, bx, by, bz;
    double limit;  // nothing at a distance above this can be, or tie with, the answer
    int bkey;
    unsigned m_lds, m_map;  // occupied cells that can still matter and have not been read: points in the LDS store / in the map only
    double sec;     // the smallest squared distance computed for any point other than the best one (DBL_MAX: none)
    unsigned seen;  // cells whose points have been read
};
__device__ __forceinline__ void wide_take(WideBest &b, double sx, double sy, double sz, double x, double y, double z, int key, bool valid) {
    const double ex = x - sx, ey = y - sy, ez = z - sz;
    const double d = (ex * ex + ey * ey) + ez * ez;
    if (valid) {
        if (d < b.best || (d == b.best && key < b.bkey)) {
            b.sec = b.best;  // (the best so far is not above anything seen before)
            b.best = d;
            b.bkey = key;
            b.bx = x;
            b.by = y;
            b.bz = z;
        } else {
            b.sec = d < b.sec ? d : b.sec;
        }
    }
}

// the table entry of cell j of the 27 (any chain length); 0: not in the table.  (Plain loads here and in the search:
// in this form nobody writes the table while it is searched -- fills and searches are separated by barriers.)
__device__ __forceinline__ unsigned wide_entry(const Tile &tile, int vx, int vy, int vz, int j, unsigned *slot_out = nullptr) {
    const int qx = vx + (int)((kShift.x >> (2 * j)) & 3) - 1, qy = vy + (int)((kShift.y >> (2 * j)) & 3) - 1, qz = vz + (int)((kShift.z >> (2 * j)) & 3) - 1;
    unsigned rkey;
    if (!tile_rel(tile, qx, qy, qz, rkey)) return 0u;
    unsigned s = tile_hash(tile, rkey);
    for (int probes = 0; probes < kTileMaxProbes; probes += 4) {  // (four slots per round trip, resolved in chain order: tile_find)
        unsigned k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = tile.keys[(s + (unsigned)u) & (unsigned)tile.slots_mask];
        int res = -2;  // undecided; -1: not in the table; else the slot
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int here = k[u] == rkey ? (int)((s + (unsigned)u) & (unsigned)tile.slots_mask) : (k[u] == kTileEmpty ? -1 : -2);
            res = res == -2 ? here : res;
        }
        if (res >= 0) {
            if (slot_out) *slot_out = (unsigned)res;
            return tile.vals[res];
        }
        if (res == -1) return 0u;
        s = (s + 4u) & (unsigned)tile.slots_mask;
    }
    return 0u;
}

// GetClosestNeighbor for the query of THIS thread against the workgroup's tile.  limit0: an upper bound of every
// distance that can still matter (see the head of this file); prune: skip voxels by their bound (off: every occupied
// voxel is visited -- same result, for the tests and the measurements).  bad: the tile cannot answer (a voxel outside
// the key span or one the table has no entry for): the caller sends the query to the map-direct search.
// (first part: the table lookups and the voxels in LDS; the voxels in the map are left in b.m_map)
template <bool PROF>
__device__ __forceinline__ void wide_search_lds(const MapView &m, const Tile &tile, WideJob &q, double limit0, bool prune, int &bad, WideCounters &ctr, WideBest &b) {
    b.best = DBL_MAX;
    b.bx = b.by = b.bz = 0.0;
    b.bkey = 0x7FFFFFFF;
    b.limit = limit0;
    b.m_map = 0u;
    b.sec = DBL_MAX;
    b.seen = 0u;
    const unsigned tp0 = PROF ? ticks32() : 0u;
    const int vx = q.v[0], vy = q.v[1], vz = q.v[2];
    const double sx = q.s[0], sy = q.s[1], sz = q.s[2];
    // ---- 1: which of the 27 cells are occupied (three batches of nine: 18 + 9 loads in flight, ~45 registers) -----------
    bad = 0;
    q.d2 = DBL_MAX;
    const bool cached = q.cached;
    unsigned m_lds = 0u, m_map = 0u, open = 0u;
    int E = 0;
    unsigned tp1 = tp0;
    if (cached) {
        m_lds = q.occ;  // (which of them are in the LDS store is found out when they are visited: voxels move there as they are used)
        E = q.E;
    } else {
    {
        bool span_ok = true;
#pragma unroll
        for (int c = 0; c < 8; ++c) {  // the corners of the 3 x 3 x 3 block decide for all 27 cells
            unsigned rkey;
            span_ok = tile_rel(tile, vx + ((c & 1) ? 1 : -1), vy + ((c & 2) ? 1 : -1), vz + ((c & 4) ? 1 : -1), rkey) && span_ok;
        }
        if (!span_ok) {
            bad = 2;
            return;
        }
    }
    auto classify = [&](unsigned v, int j) {
        if (v == 0u) return;
        if (v == kTileOverflow || !(v & kTileReady)) {
            bad = 2;
            return;
        }
        E += tile_cnt(v);
        if (v & kTileGlobal)
            m_map |= 1u << j;
        else
            m_lds |= 1u << j;
    };
#pragma unroll
    for (int jb = 0; jb < 27; jb += 9) {
        unsigned rk[9], h0[9], ka[9], kb[9], val[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int j = jb + u;
            const int qx = vx + (int)((kShift.x >> (2 * j)) & 3) - 1, qy = vy + (int)((kShift.y >> (2 * j)) & 3) - 1, qz = vz + (int)((kShift.z >> (2 * j)) & 3) - 1;
            (void)tile_rel(tile, qx, qy, qz, rk[u]);
            h0[u] = tile_hash(tile, rk[u]);
            ka[u] = tile.keys[h0[u]];
            kb[u] = tile.keys[(h0[u] + 1u) & (unsigned)tile.slots_mask];
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const bool hit_a = ka[u] == rk[u];
            const bool go_b = !hit_a && ka[u] != kTileEmpty;
            const bool hit_b = go_b && kb[u] == rk[u];
            if (go_b && !hit_b && kb[u] != kTileEmpty) open |= 1u << (jb + u);
            val[u] = 0u;
            if (hit_a || hit_b) val[u] = tile.vals[hit_a ? h0[u] : ((h0[u] + 1u) & (unsigned)tile.slots_mask)];
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) classify(val[u], jb + u);
    }
    tp1 = PROF ? ticks32() : 0u;
    while (open) {  // chains longer than two slots
        const int j = __ffs(open) - 1;
        open &= open - 1u;
        // (the key and its home slot are recomputed from j: no dynamically indexed register array)
        const int qx = vx + (int)((kShift.x >> (2 * j)) & 3) - 1, qy = vy + (int)((kShift.y >> (2 * j)) & 3) - 1, qz = vz + (int)((kShift.z >> (2 * j)) & 3) - 1;
        unsigned rkey;
        (void)tile_rel(tile, qx, qy, qz, rkey);
        unsigned s = (tile_hash(tile, rkey) + 2u) & (unsigned)tile.slots_mask;
        // (four slots of the chain per round trip, resolved in chain order: a miss in a table 5/8 full walks four slots on average
        // and thirty at worst, and a wave waits for its slowest lane -- 3 us on average, 36 at worst in the first iteration of the
        // 1M-point configuration, profiles/r06_final_icp_probe_livox100.txt.  No key lies beyond kTileMaxProbes slots from
        // its home: the two the last round looks at beyond that can only end the chain.)
        for (int probes = 2; probes < kTileMaxProbes; probes += 4) {
            unsigned k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = tile.keys[(s + (unsigned)u) & (unsigned)tile.slots_mask];
            int res = -2;  // undecided; -1: not in the table; else the slot
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int here = k[u] == rkey ? (int)((s + (unsigned)u) & (unsigned)tile.slots_mask) : (k[u] == kTileEmpty ? -1 : -2);
                res = res == -2 ? here : res;
            }
            if (res >= 0) classify(tile.vals[res], j);
            if (res != -2) break;
            s = (s + 4u) & (unsigned)tile.slots_mask;
        }
    }
    if (bad) return;
    q.occ = m_lds | m_map;
    q.cached = true;
    }
    q.E = E;
    const unsigned tp2 = PROF ? ticks32() : 0u;
    // ---- 2: this thread walks voxels itself, in shift order (its own voxel, the faces, ...), as long as that is cheap: the
    // first one always, then up to kWideWalkVoxels / kWideWalkPoints in all -- after every walk the bounds of all 27
    // cells (sums of three of the nine squared gaps, picked at compile time) are held against what is now in hand.
    // What survives beyond the budget (dense surroundings) is left in b.m_lds for the queue.
    const WideGaps gaps = wide_gaps(q.s, q.v, m.voxel_size);
    auto keep_mask = [&]() { return wide_keep_mask(gaps, b.limit); };
    if (prune) m_lds &= keep_mask();
    int walked_points = 0;
    for (int visits = 0; m_lds != 0u && visits < kWideWalkVoxels;) {
        const int j = __ffs(m_lds) - 1;
        const unsigned v = wide_entry(tile, vx, vy, vz, j);
        if (v & kTileGlobal) {  // (cached occupancy: this one is in the map)
            m_lds &= m_lds - 1u;
            m_map |= 1u << j;
            continue;
        }
        const int ref = tile_ref(v), cnt = tile_cnt(v);
        if (visits > 0 && walked_points + cnt > kWideWalkPoints) break;
        ++visits;
        m_lds &= m_lds - 1u;
        b.seen |= 1u << j;
        const double *P = tile.points + 3 * ref;
        for (int k0 = 0; k0 < cnt; k0 += 2) {
            const int k1 = k0 + 1 < cnt ? k0 + 1 : k0;
            const double x0 = P[3 * k0], y0 = P[3 * k0 + 1], z0 = P[3 * k0 + 2];
            const double x1 = P[3 * k1], y1 = P[3 * k1 + 1], z1 = P[3 * k1 + 2];
            wide_take(b, sx, sy, sz, x0, y0, z0, (j << 5) | k0, true);
            wide_take(b, sx, sy, sz, x1, y1, z1, (j << 5) | k1, k1 != k0);
        }
        walked_points += cnt;
        b.limit = b.best < b.limit ? b.best : b.limit;
        if (prune) m_lds &= keep_mask();
        if (PROF) ++ctr.visited_lds;
    }
    if (cached) {  // what is left for the queues: which queue
        unsigned todo = m_lds;
        while (todo) {
            const int j = __ffs(todo) - 1;
            todo &= todo - 1u;
            if (wide_entry(tile, vx, vy, vz, j) & kTileGlobal) {
                m_lds &= ~(1u << j);
                m_map |= 1u << j;
            }
        }
    }
    if (prune) m_map &= keep_mask();
    b.m_lds = m_lds;
    b.m_map = m_map;
    if (PROF) {
        ctr.t_lookup = tp1 - tp0;
        ctr.t_chains = tp2 - tp1;
        ctr.t_walk = ticks32() - tp2;
    }
}

// the queue of {query, voxel} items, served by the 32-lane groups: lane i reads point i of the voxel (from the LDS store,
// or from the map: one 16-byte and one 8-byte load, coalesced), kChunk voxels of a group in flight; the closest point of
// the voxel -- the smaller index among equals, std::min_element's first minimum (VoxelHashMap.cpp:58-61) -- goes back
// into the item
template <bool LDS>
// runner_up: also find the voxel's second smallest distance (the stability test's margin); without it the item reports the
// smallest again, i.e. no margin -- right for the first iteration, after which nearly every query is searched again anyway.
__device__ __forceinline__ void wide_serve_items(const MapView &m, const Tile &tile, WideItem *items, int n_items, int grp, int lane, bool promote, bool runner_up) {
    constexpr int kFly = LDS ? kChunk : 6;  // voxels of a group in flight (map voxels: twelve -- a trip is an HBM / L2 round trip)
    for (int e0 = grp; __ballot(e0 < n_items) != 0ull; e0 += kIcpGroupsPerBlock * kFly) {  // wave-uniform trip count
        double2 xy[kFly];
        double zz[kFly];
        bool ld[kFly], valid[kFly];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            const int e = e0 + kIcpGroupsPerBlock * u;
            valid[u] = e < n_items;
            const WideItem &it = items[valid[u] ? e : 0];
            const int blk = (int)(it.blk_cnt & 0xFFFFFFu), cnt = (int)((it.blk_cnt >> 24) & 63u);
            ld[u] = valid[u] && lane < cnt;
            if (ld[u]) {
                if (LDS) {
                    const double *P = tile.points + 3 * (blk + lane);
                    xy[u].x = P[0];
                    xy[u].y = P[1];
                    zz[u] = P[2];
                } else {
                    xy[u] = block_xy(m, blk)[lane];
                    zz[u] = block_z(m, blk)[lane];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
            if (!LDS && promote && valid[u]) {
                // PROMOTION: a voxel that is read from the map is one some query really looks at -- and will look at again in
                // the next iteration (the search is bounded by the last neighbour).  While the store has room the group leaves
                // the points it has just read there and the table entry says so from now on: the store turns into a cache of
                // the voxels that ARE visited, whatever the window phase guessed.  (Nobody searches during this phase; two
                // groups promoting the same voxel in the same round leave two copies, one of them unused.)
                WideItem &it = items[e0 + kIcpGroupsPerBlock * u];
                const int cnt = (int)((it.blk_cnt >> 24) & 63u);
                int off = -1;
                if (lane == 0 && (tile.vals[it.slot] & kTileGlobal)) {
                    const int o = atomicAdd(tile.count, cnt);
                    if ((unsigned)(o + cnt) * 24u <= tile.region_bytes && o + cnt <= 0xFFFF) {
                        atomicMax(tile.stored, o + cnt);
                        off = o;
                    }
                }
                off = __shfl(off, 0, 32);
                if (off >= 0) {
                    if (ld[u]) {
                        double *q = tile.points + 3 * (off + lane);
                        q[0] = xy[u].x;
                        q[1] = xy[u].y;
                        q[2] = zz[u];
                    }
                    if (lane == 0) tile.vals[it.slot] = (unsigned)off | ((unsigned)cnt << 24) | kTileReady;
                }
            }
            double d = DBL_MAX;
            if (ld[u]) {
                const WideItem &it = items[e0 + kIcpGroupsPerBlock * u];
                const double ex = xy[u].x - it.s[0], ey = xy[u].y - it.s[1], ez = zz[u] - it.s[2];
                d = (ex * ex + ey * ey) + ez * ez;
            }
            double gd = d;
            int gk = ld[u] ? lane : 0x7FFFFFFF;
            group_min_dist_key(gd, gk);
            double g2 = gd;
            if (runner_up) {  // (the whole workgroup alike)
                g2 = (ld[u] && gk != lane) ? d : DBL_MAX;
                group_fmin_step<0>(g2);
                group_fmin_step<1>(g2);
                group_fmin_step<2>(g2);
                group_fmin_step<3>(g2);
                group_fmin_step<4>(g2);
            }
            if (valid[u] && ld[u] && gk == lane) {
                WideItem &it = items[e0 + kIcpGroupsPerBlock * u];
                it.s[0] = xy[u].x;
                it.s[1] = xy[u].y;
                it.s[2] = zz[u];
                it.d2 = gd;
                it.k = (unsigned char)lane;
                it.blk_cnt = __float_as_uint(g2 < (double)FLT_MAX ? __double2float_rd(g2) : FLT_MAX);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same service by the WHOLE WORKGROUP, a thread per POINT (option icp_wide_flat).  A group serving items one after
// the other keeps six voxels in flight and spends a round trip to L2 / HBM on each such batch whatever the voxels hold --
// 22 us per round of ~350 items, 180 us of the first iteration of a workgroup of the 1M-point configuration
// (profiles/r04_final_icp_probe_livox100.txt), with a handful of points in most voxels.  Here the points of all items of a
// round form one sequence (prefix sums of the items' counts), thread t takes points t, t + 512, ... (four in flight: 2048
// points per pass, every load of a pass issued before the first is used), finds the item of each by bisection of the
// prefix sums, and the items' minima are settled by LDS atomics in the item records themselves:
//   pass 1  umin of the distance's bit pattern (non-negative doubles order like their bits) into WideItem::d2;
//   pass 2  the points AT the minimum: umin of {slot | j << 16 | index << 24} -- the smallest index wins, as in the reference's
//           strict '<' walk (VoxelHashMap.cpp:58-63); every other point: umin of its distance (float, rounded down) into
//           WideItem::blk_cnt -- the runner-up bound;
//   pass 3  the winner leaves its coordinates; a point that ties with it counts as a runner-up.
// PROMOTION is settled before the loads: the leader of an item claims the table entry (compare-and-swap of the entry's map
// form against the same without kTileReady: one claim per voxel however many queries ask for it in this round -- the
// group-wise service left a copy per asking group in the store), takes room in the store and enters the new position;
// the threads of that item's points store what they have loaded.  Nobody reads the table or the store during a service.
// scr: 2 KiB of LDS nobody uses during a service (kicp_icp.hip: the records of the compacted searches).
// ------------------------------------------------------------------------------------------
constexpr int kWideFlatPer = 4;
constexpr size_t kWideFlatScratchBytes = sizeof(unsigned short) * 2 * (kWideItems + 2) + sizeof(int) * (kIcpThreads / 64);
static_assert(kWideItems < kIcpThreads, "a thread per item leads it");
template <bool LDS>
__device__ __forceinline__ void wide_serve_flat(const MapView &m, const Tile &tile, WideItem *items, int n, void *scr, bool promote) {
    const int tid = kicp_tid(), lane64 = tid & 63, wave = tid >> 6;
    unsigned short *start = reinterpret_cast<unsigned short *>(scr);  // [n + 1]: first point of item e in the round's sequence
    unsigned short *offs = start + (kWideItems + 2);                  // [n]: where item e's voxel goes in the LDS store (0xFFFF: nowhere)
    int *wsum = reinterpret_cast<int *>(offs + (kWideItems + 2));     // [8]: points per wave of leaders
    // (1) item e is led by thread e: its count, the prefix sums, the promotion
    unsigned bc = 0u;
    int cnt = 0;
    if (tid < n) {
        bc = items[tid].blk_cnt;
        cnt = (int)((bc >> 24) & 63u);
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane64 >= d) incl += t;
    }
    if (lane64 == 63) wsum[wave] = incl;
    if (tid < n) {
        unsigned short o16 = 0xFFFFu;
        if (!LDS && promote && cnt > 0) {
            const unsigned slot = items[tid].slot;
            const unsigned in_map = bc | kTileGlobal | kTileReady;  // the entry of a voxel that is read from the map
            if (atomicCAS(&tile.vals[slot], in_map, bc | kTileGlobal) == in_map) {  // claimed: this leader alone deals with the voxel
                const int o = atomicAdd(tile.count, cnt);
                if ((unsigned)(o + cnt) * 24u <= tile.region_bytes && o + cnt <= 0xFFFF) {
                    atomicMax(tile.stored, o + cnt);
                    o16 = (unsigned short)o;
                    __hip_atomic_store(&tile.vals[slot], (unsigned)o | ((unsigned)cnt << 24) | kTileReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    __hip_atomic_store(&tile.vals[slot], in_map, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // no room: it stays where it is
                }
            }
        }
        offs[tid] = o16;
    }
    __syncthreads();
    {
        int before = 0;
#pragma unroll
        for (int w = 0; w < kIcpThreads / 64; ++w) before += w < wave ? wsum[w] : 0;
        if (tid <= n) start[tid] = (unsigned short)(before + incl - cnt);  // (thread n: the total)
    }
    __syncthreads();
    // (2) passes of at most kWideFlatPer * 512 points (whole items; every thread computes the same bounds)
    for (int lo = 0; lo < n;) {
        const int p_lo = (int)start[lo];
        int hi = lo + 1;  // (an item has at most 63 points)
#pragma unroll
        for (int step = 256; step; step >>= 1) {
            const int c = hi + step;
            if (c <= n && (int)start[c] - p_lo <= kWideFlatPer * kIcpThreads) hi = c;
        }
        const int np = (int)start[hi] - p_lo;
        int e[kWideFlatPer], idx[kWideFlatPer];
        double2 xy[kWideFlatPer];
        double zz[kWideFlatPer], d[kWideFlatPer];
        bool ok[kWideFlatPer];
#pragma unroll
        for (int u = 0; u < kWideFlatPer; ++u) {
            const int r = tid + u * kIcpThreads, p = p_lo + r;
            ok[u] = r < np;
            int x = lo;  // the last item of [lo, hi) that starts at or before p (items without points are passed over)
#pragma unroll
            for (int step = 256; step; step >>= 1) {
                const int c = x + step;
                if (c < hi && (int)start[c] <= p) x = c;
            }
            e[u] = x;
            idx[u] = p - (int)start[x];
            d[u] = DBL_MAX;
            if (ok[u]) {
                const int blk = (int)(items[x].blk_cnt & 0xFFFFFFu);
                if (LDS) {
                    const double *P = tile.points + 3 * (blk + idx[u]);
                    xy[u].x = P[0];
                    xy[u].y = P[1];
                    zz[u] = P[2];
                } else {
                    xy[u] = block_xy(m, blk)[idx[u]];
                    zz[u] = block_z(m, blk)[idx[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kWideFlatPer; ++u)
            if (ok[u]) {
                WideItem &it = items[e[u]];
                const double ex = xy[u].x - it.s[0], ey = xy[u].y - it.s[1], ez = zz[u] - it.s[2];
                d[u] = (ex * ex + ey * ey) + ez * ez;
                __hip_atomic_fetch_min(reinterpret_cast<unsigned long long *>(&it.d2), (unsigned long long)__double_as_longlong(d[u]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!LDS) {
                    const unsigned o = offs[e[u]];
                    if (o != 0xFFFFu) {
                        double *q = tile.points + 3 * ((int)o + idx[u]);
                        q[0] = xy[u].x;
                        q[1] = xy[u].y;
                        q[2] = zz[u];
                    }
                }
            }
        __syncthreads();  // every item's minimum is settled; its count and its query have been read
        if (tid < hi - lo) {
            WideItem &it = items[lo + tid];
            it.blk_cnt = __float_as_uint(FLT_MAX);
            it.k = 0xFFu;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kWideFlatPer; ++u)
            if (ok[u]) {
                WideItem &it = items[e[u]];
                if (d[u] == it.d2) {
                    unsigned *w = reinterpret_cast<unsigned *>(&it.slot);  // {slot, j, k}: k is the top byte
                    __hip_atomic_fetch_min(w, ((unsigned)it.slot | ((unsigned)it.j << 16)) | ((unsigned)idx[u] << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    __hip_atomic_fetch_min(&it.blk_cnt, __float_as_uint(d[u] < (double)FLT_MAX ? __double2float_rd(d[u]) : FLT_MAX), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kWideFlatPer; ++u)
            if (ok[u]) {
                WideItem &it = items[e[u]];
                if (d[u] == it.d2) {
                    if ((int)it.k == idx[u]) {
                        it.s[0] = xy[u].x;
                        it.s[1] = xy[u].y;
                        it.s[2] = zz[u];
                    } else {
                        __hip_atomic_fetch_min(&it.blk_cnt, __float_as_uint(d[u] < (double)FLT_MAX ? __double2float_rd(d[u]) : FLT_MAX), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        lo = hi;  // (the next pass touches other items; the caller's barrier ends the service)
    }
}

// ------------------------------------------------------------------------------------------
// A FEW full searches (the later iterations: 10 - 30 of a workgroup's 400 queries, profiles/r04_ap_icp_probe_livox100.txt)
// are not worth the queues: filing, two flat services and the merge have fixed costs of ~25 us however few items there
// are.  A 32-lane group takes a query instead, as in the first form (tile_scan, kicp_search.hpp): lane j < 27 looks up
// cell j of the reference's shift table and walks its points in the LDS store, the cells whose points are in the map are
// read by the group together (lane i point i, kChunk voxels in flight) -- the reference's order, its strict '<'.  Unlike
// tile_scan this one skips cells by their box bounds (wide_gaps: exact, as in the thread's own search) -- with the last
// neighbour's distance as the first limit a query reads one to four cells, one trip to the map instead of five --, and it
// returns what the stability test needs: which cells are occupied and a lower bound of the distance to every point but
// the neighbour: the SECOND smallest distance over the points read (a lane keeps the runner-up of what it has seen; over
// the group the winner's lane contributes its runner-up, every other lane its best) or the bound of an occupied cell that
// was not read.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wide_group_scan(const MapView &m, const Tile &tile, const double s[3], const int v[3], double limit0, int lane, double nn[3],
                                                  int &examined, unsigned &occ, double &second, int &bad) {
    constexpr int U = 4;
    const double sx = s[0], sy = s[1], sz = s[2];
    int ref = 0, cnt = 0;
    int mybad = 0;
    bool glob = false;
    double bd = DBL_MAX;  // lower bound of the distance to any point of this lane's cell (wide_gaps)
    if (lane < 27) {
        const int cx = (int)((kShift.x >> (2 * lane)) & 3), cy = (int)((kShift.y >> (2 * lane)) & 3), cz = (int)((kShift.z >> (2 * lane)) & 3);
        const WideGaps gaps = wide_gaps(s, v, m.voxel_size);
        const double bx = cx == 0 ? gaps.m2[0] : (cx == 2 ? gaps.p2[0] : 0.0);
        const double by = cy == 0 ? gaps.m2[1] : (cy == 2 ? gaps.p2[1] : 0.0);
        const double bz = cz == 0 ? gaps.m2[2] : (cz == 2 ? gaps.p2[2] : 0.0);
        bd = (bx + by) + bz;
        unsigned rkey;
        if (tile_rel(tile, v[0] + cx - 1, v[1] + cy - 1, v[2] + cz - 1, rkey)) {
            const int slot = tile_find(tile, rkey);
            if (slot >= 0) {
                const unsigned val = tile.vals[slot];
                if (val == kTileOverflow || !(val & kTileReady)) {
                    mybad = 2;
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
    bad = (unsigned)(__ballot(mybad != 0) >> half_shift) != 0u ? 2 : 0;
    if (bad) cnt = 0;
    occ = (unsigned)(__ballot(cnt > 0) >> half_shift);
    int tot = cnt;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 32);
    examined = tot;
    // ---- cells in the LDS store: every lane walks its own, unless nothing in it can matter (bound above limit0:
    // beyond the correspondence threshold, or farther than the last neighbour, which is still there)
    double best = DBL_MAX, sec = DBL_MAX;
    int bk = 0;
    bool read = !glob && cnt > 0 && !(bd > limit0 * kNormTie);
    {
        const double *P = tile.points + 3 * (read ? ref : 0);
        const int c = read ? cnt : 0;
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
                sec = (in & (loser < sec)) ? loser : sec;
                best = take ? d : best;
                bk = take ? k0 + u : bk;
            }
        }
    }
    int key = (read && best < DBL_MAX) ? ((lane << 5) | bk) : 0x7FFFFFFF;
    double bx = 0.0, by = 0.0, bz = 0.0;
    if (key != 0x7FFFFFFF) {
        const double *q = tile.points + 3 * (ref + bk);
        bx = q[0];
        by = q[1];
        bz = q[2];
    }
    // ---- cells whose points are in the map: the group reads them together, kChunk cells per trip, and after every trip
    // the cells left are held against what is in hand (a cell is skipped only when its bound is STRICTLY above it)
    double limit = limit0;
    {
        double g = best;
        group_fmin_step<0>(g);
        group_fmin_step<1>(g);
        group_fmin_step<2>(g);
        group_fmin_step<3>(g);
        group_fmin_step<4>(g);
        limit = g < limit ? g : limit;
    }
    unsigned gl = (unsigned)(__ballot(glob && cnt > 0 && !(bd > limit * kNormTie)) >> half_shift);
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
            if (j == lane) read = true;
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
                    sec = best < sec ? best : sec;
                    best = d;
                    key = k;
                    bx = xy[u].x;
                    by = xy[u].y;
                    bz = zz[u];
                } else {
                    sec = d < sec ? d : sec;
                }
            }
        }
        if (__ballot(gl != 0) != 0ull) {  // (more to come: what of it can still matter)
            double g = best;
            group_fmin_step<0>(g);
            group_fmin_step<1>(g);
            group_fmin_step<2>(g);
            group_fmin_step<3>(g);
            group_fmin_step<4>(g);
            limit = g < limit ? g : limit;
            gl &= (unsigned)(__ballot(!(bd > limit * kNormTie)) >> half_shift);
        }
    }
    if (best == DBL_MAX) key = 0x7FFFFFFF;
    const int mykey = key;
    const double mybest = best;
    group_min_dist_key(best, key);
    const bool found = key != 0x7FFFFFFF;
    const unsigned who = (unsigned)(__ballot(found && mykey == key) >> half_shift);
    const int wl = who ? (__ffs(who) - 1) : 0;
    nn[0] = __shfl(bx, wl, 32);
    nn[1] = __shfl(by, wl, 32);
    nn[2] = __shfl(bz, wl, 32);
    // the runner-up: of the points read, and no closer than that can any point of an occupied cell be that was not read
    double g2 = (found && lane == wl) ? sec : mybest;
    if (cnt > 0 && !read) g2 = bd < g2 ? bd : g2;
    group_fmin_step<0>(g2);
    group_fmin_step<1>(g2);
    group_fmin_step<2>(g2);
    group_fmin_step<3>(g2);
    group_fmin_step<4>(g2);
    second = g2;
    return best;
}

// the end of a full search: its distance, and the bound that lets the next iterations do without a search (WideQuery::Lr)
// -- the second smallest distance computed, or the smallest box bound of an occupied cell that was not read
__device__ __forceinline__ void wide_finish(const MapView &m, WideJob &q, const WideBest &b) {
    q.d2 = b.best;
    q.bkey = b.bkey;
    double L = b.sec;
    const unsigned unread = q.occ & ~b.seen;
    if (unread) {
        const WideGaps gaps = wide_gaps(q.s, q.v, m.voxel_size);
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            const int cx = (int)((kShift.x >> (2 * j)) & 3), cy = (int)((kShift.y >> (2 * j)) & 3), cz = (int)((kShift.z >> (2 * j)) & 3);
            const double bx = cx == 0 ? gaps.m2[0] : (cx == 2 ? gaps.p2[0] : 0.0);
            const double by = cy == 0 ? gaps.m2[1] : (cy == 2 ? gaps.p2[1] : 0.0);
            const double bz = cz == 0 ? gaps.m2[2] : (cz == 2 ? gaps.p2[2] : 0.0);
            const double bd = (bx + by) + bz;
            if (((unread >> j) & 1u) && bd < L) L = bd;
        }
    }
    q.Lr = sqrt(L) * (1.0 - 0x1p-30);
}

// ------------------------------------------------------------------------------------------
// The first iteration's window phase for up to 512 queries at once (the same four phases as tile_fill_bulk,
// kicp_icp.hip; differences: the windows come from the owning threads' registers; the set of distinct cells has up
// to 16384 slots -- 400 queries of the 1M-point configuration have five to eight thousand distinct cells, and a set
// as crowded as 4096 slots were cost 45 us of probing in some workgroups (profiles/r03_ac_icp_probe_livox.txt); the
// fetch jobs live in sh.terms + sh.pts).  NEAR CELLS FIRST: the LDS store holds ~3.8 k points, the windows of a
// workgroup of the 1M-point configuration ask for 6 .. 10 k, and what does not fit is read from the map by single
// threads at every search (60 - 70 us per iteration in such workgroups against 8 where everything fits,
// profiles/r04_a_icp_probe_livox.txt).  But a search that skips voxels by their distance visits the query's own voxel
// and a face neighbour or two: so the cells within one step of some query's voxel (a flag bit in the set) are looked up
// and given room in the store first, edges / corners / window extensions take what is left.
// mine: this thread's query takes part (flag 1).  Leaves the verdict in metas[tid].valid (1 / -1).  false: no room
// for the scratch (nothing was changed): the caller establishes the windows one by one.
// ------------------------------------------------------------------------------------------
// Returns 1 done, 0 no room for the scratch, 2 more distinct cells than the member list holds (nothing was entered; the
// caller comes again with fewer queries: 128 queries have at most 8192 cells).  Queries q_lo <= q < q_hi take part.
// prefill_eighths: how much of the store the window phase may fill with points (0: none -- the table only; 8: all of it).
__device__ __forceinline__ int wide_fill_bulk(const MapView &m, const Tile &tile, IcpShared *shp, int q_lo, int q_hi, WideMeta *metas, bool mine, const double s[3], const int v[3],
                                              int *range_err_out, bool prof, int prefill_eighths) {
    const int cn = q_hi - q_lo;
    IcpShared &sh = *shp;
    const int tid = kicp_tid();
    unsigned tk = prof ? ticks32() : 0u;
    auto stamp = [&](int ph) {
        if (prof && tid == 0) {
            const unsigned now = ticks32();
            sh.bulk_ticks[ph] = now - tk;
            tk = now;
        }
    };
    unsigned *jobs = reinterpret_cast<unsigned *>(sh.terms);  // {block id | count << 24, store offset | table slot << 16}
    int range_err = 0;
    const int s0 = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // the store's end before this phase
    const unsigned free_top = tile.region_bytes & ~15u;  // (this form keeps no scan lists: the region is points only)
    const unsigned prefill_top = prefill_eighths > 0 ? max((unsigned)s0 * 24u, (free_top / 8u) * (unsigned)prefill_eighths) : 0u;
    // the set of distinct cells (u32 relative keys), a bit per slot ("near"), and the list of the set's members (u16 slot
    // numbers: near cells from the bottom, the others from the top): the largest power of two of slots, 64 per query at
    // most, that leaves the list 3/4 of the slots (at least 1024 entries) above the points
    const unsigned room = free_top - min(free_top, (unsigned)s0 * 24u);
    int set_log2 = 14;
    while (set_log2 > 10 && ((1 << set_log2) > 64 * max(cn, 16) || (4u << set_log2) + (1u << (set_log2 - 3)) + (6u << (set_log2 - 2)) > room)) --set_log2;
    if ((4u << set_log2) + (1u << (set_log2 - 3)) + 2048u > room) return 0;
    const int S = 1 << set_log2;
    const int Lcap = (int)min((unsigned)S, ((room - (4u << set_log2) - (1u << (set_log2 - 3))) / 2u) & ~7u);
    unsigned *set = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(tile.points) + free_top) - S;
    unsigned *near_bits = set - S / 32;
    unsigned short *cells = reinterpret_cast<unsigned short *>(near_bits) - Lcap;
    // ---- 1: the windows; the set is cleared ---------------------------------------------------------------------------
    for (int i = tid; i < S; i += kIcpThreads) set[i] = kTileEmpty;
    for (int i = tid; i < S / 32; i += kIcpThreads) near_bits[i] = 0u;
    if (tid == 0) sh.job_count = sh.bulk_failed = sh.cell_count = sh.list_entries = 0;  // (cell_count: near cells, list_entries: the others)
    if (mine) {
        WideMeta *meta = metas + tid;  // (the chunk's queries are the threads': query tid, q_lo <= tid < q_hi)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double f = s[a] / m.voxel_size - (double)v[a];  // position inside the voxel, [0, 1)
            meta->v[a] = v[a];
            meta->lo[a] = (signed char)((f < kWindowMargin) ? -2 : -1);
            meta->hi[a] = (signed char)((f > 1.0 - kWindowMargin) ? 2 : 1);
        }
        meta->valid = 0;       // pending; -1 as soon as any of its cells cannot be served from the tile
        meta->list_state = 2;  // takes part
    }
    __syncthreads();
    // ---- 2a: the DISTINCT cells of all windows that the table does not know yet ----------------------------------------
    auto for_each_cell = [&](auto &&fn) {  // fn(query, relative key, within one step of the query's voxel) for every in-range cell of every window taking part
        for (int idx = tid; idx < cn * 64; idx += kIcpThreads) {
            const int qt = q_lo + (idx >> 6);
            // (the query's record -- five words, the same for the 64 threads of a wave -- in one round trip, in front of the test on it:
            // thirty-two rounds of this loop per batch of 256 queries, each a chain of dependent LDS round trips)
            WideMeta Mq;
            __builtin_memcpy(&Mq, __builtin_assume_aligned(metas + qt, 4), sizeof Mq);
            const WideMeta *meta = &Mq;
            if (meta->list_state != 2) continue;
            const int ny = meta->hi[1] - meta->lo[1] + 1, nz = meta->hi[2] - meta->lo[2] + 1, nx = meta->hi[0] - meta->lo[0] + 1;
            const int w = idx & 63;
            if (w >= nx * ny * nz) continue;
            // (window sides are 3 or 4 cells, w < 64: a shift or a multiplication instead of integer divisions)
            const int t2 = nz == 4 ? w >> 2 : (w * 43) >> 7, iz = w - t2 * nz;
            const int ix = ny == 4 ? t2 >> 2 : (t2 * 43) >> 7, iy = t2 - ix * ny;
            const int ox = meta->lo[0] + ix, oy = meta->lo[1] + iy, oz = meta->lo[2] + iz;
            const int qx = meta->v[0] + ox, qy = meta->v[1] + oy, qz = meta->v[2] + oz;
            if (!voxel_in_range(qx, qy, qz)) {
                if (ox >= -1 && ox <= 1 && oy >= -1 && oy <= 1 && oz >= -1 && oz <= 1) range_err = 1;
                continue;
            }
            unsigned rkey;
            if (!tile_rel(tile, qx, qy, qz, rkey)) {
                metas[qt].valid = -1;  // outside the span of the relative keys
                continue;
            }
            fn(qt, rkey, abs(ox) + abs(oy) + abs(oz) <= 1);
        }
    };
    const bool fresh = __hip_atomic_load(tile.entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
    const unsigned set_mask = (unsigned)(S - 1);
    for_each_cell([&](int qt, unsigned rkey, bool near) {
        const int slot = fresh ? -1 : tile_find(tile, rkey);
        if (slot >= 0) {
            if (__hip_atomic_load(&tile.vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kTileOverflow) metas[qt].valid = -1;
            return;
        }
        unsigned si = (rkey * 0x9E3779B1u) >> (32 - set_log2);
        bool done = false;
        for (int probes = 0; probes < 64; ++probes) {
            const unsigned old = atomicCAS(&set[si], kTileEmpty, rkey);
            if (old == kTileEmpty || old == rkey) {
                if (near) atomicOr(&near_bits[si >> 5], 1u << (si & 31u));
                done = true;
                break;
            }
            si = (si + 1u) & set_mask;
        }
        if (!done) metas[qt].list_state = 3;  // (a set this crowded: this query's window is established on its own afterwards)
    });
    __syncthreads();
    // the members: near cells from the bottom of the list, the others from its top (one LDS atomic per wave and class)
    for (int i = tid; i < S; i += kIcpThreads) {  // (S is a multiple of the workgroup size: every wave takes every trip)
        const unsigned e = set[i];
        const bool occ = e != kTileEmpty, nr = occ && ((near_bits[i >> 5] >> (i & 31)) & 1u), fr = occ && !nr;
        const unsigned long long nm = __ballot(nr), fm = __ballot(fr);
        int nb = 0, fb = 0;
        if ((tid & 63) == 0) {
            if (nm) nb = atomicAdd(&sh.cell_count, (int)__popcll(nm));
            if (fm) fb = atomicAdd(&sh.list_entries, (int)__popcll(fm));
        }
        nb = __shfl(nb, 0, 64);
        fb = __shfl(fb, 0, 64);
        const int nrank = nb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
        const int frank = fb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
        if (nr && nrank < Lcap) cells[nrank] = (unsigned short)i;
        if (fr && frank < Lcap) cells[Lcap - 1 - frank] = (unsigned short)i;
    }
    __syncthreads();
    const int n_near = sh.cell_count, n_far = sh.list_entries, n_cells = n_near + n_far;
    const bool cells_lost = n_cells > Lcap;  // (more distinct cells than the list holds)
    __syncthreads();
    if (tid == 0) sh.cell_count = sh.list_entries = 0;  // (the caller's counters again)
    if (cells_lost) {  // nothing has been entered: the queries are as they were, the caller comes again with fewer of them
        if (mine) {
            metas[tid].valid = 0;
            metas[tid].list_state = 0;
        }
        __syncthreads();
        return 2;
    }
    stamp(0);
    // ---- 2b: one map lookup per distinct cell, three in flight per thread; occupied voxels enter the table and file a fetch job
    constexpr int kBatch = 2;
    // (two segments with a barrier between them: every near cell has its room in the store before the first of the others asks)
    for (int seg = 0; seg < 2; ++seg) {
    const int j_end = seg == 0 ? n_near : n_cells;
    for (int jb = seg == 0 ? 0 : n_near; jb < j_end; jb += kIcpThreads * kBatch) {  // (every thread takes every trip: the counters below are kept wave by wave)
        unsigned long long key[kBatch];
        unsigned rkey[kBatch];
        Slot a[kBatch][kProbeAhead];
        uint32_t hs[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int j = jb + tid + u * kIcpThreads;  // near cells first: they take their room in the store before the others ask
            rkey[u] = j < j_end ? set[cells[j < n_near ? j : Lcap - 1 - (j - n_near)]] : kTileEmpty;
            key[u] = 0;
            hs[u] = 0;
            if (rkey[u] != kTileEmpty) {
                key[u] = tile_unrel(tile, rkey[u]);
                hs[u] = hash_key(key[u], m.mask);
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
#pragma unroll
            for (int i = 0; i < kProbeAhead; ++i) {
                a[u][i].key = kKeyEmpty;
                a[u][i].block = -1;
                a[u][i].count = 0;
                if (rkey[u] != kTileEmpty) a[u][i] = load_slot(m.slots + ((hs[u] + i) & m.mask));
            }
        int blks[kBatch], cnts[kBatch];
        bool open_any = false;
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const bool done = probe_resolve(a[u], key[u], blks[u], cnts[u]);  // (an unused entry resolves at once: its slots read "empty")
            if (!done) {
                hs[u] = (hs[u] + kProbeAhead) & m.mask;
                open_any = true;
            } else {
                key[u] = kKeyEmpty;  // closed
            }
        }
        for (uint32_t probes = kProbeAhead; open_any && probes <= m.mask; probes += kProbeAhead) {  // long chains: all of a thread's together
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
#pragma unroll
                for (int i = 0; i < kProbeAhead; ++i)
                    if (key[u] != kKeyEmpty) a[u][i] = load_slot(m.slots + ((hs[u] + i) & m.mask));
            open_any = false;
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (key[u] == kKeyEmpty) continue;
                if (probe_resolve(a[u], key[u], blks[u], cnts[u])) {
                    key[u] = kKeyEmpty;
                } else {
                    hs[u] = (hs[u] + kProbeAhead) & m.mask;
                    open_any = true;
                }
            }
        }
        // Entering: the table slot is a CAS of the lane's own; the counters (slots occupied, points asked of the store, jobs
        // filed, the store's end) are ONE LDS atomic per wave and counter (prefix sums over ballots), as in tile_fill_bulk.
        const int lane64 = tid & 63;
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int blk = blks[u], cnt = cnts[u];
            const bool occ = rkey[u] != kTileEmpty && blk >= 0 && cnt > 0;  // (an empty voxel: the table holds occupied ones only)
            const unsigned long long om = __ballot(occ);
            if (om == 0ull) continue;  // (the whole wave)
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(om >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)om, 0u));
            int pre = occ ? cnt : 0;  // inclusive prefix of the points asked for
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(pre, o, 64);
                if (lane64 >= o) pre += t;
            }
            const int total = __shfl(pre, 63, 64);
            int ebase = 0, obase = 0;
            if (lane64 == 0) {
                ebase = atomicAdd(tile.entries, (int)__popcll(om));  // slots are reserved before they are taken: the limit holds exactly
                obase = atomicAdd(tile.count, total);
            }
            ebase = __shfl(ebase, 0, 64);
            obase = __shfl(obase, 0, 64);
            const int off = obase + pre - (occ ? cnt : 0);
            unsigned sidx = 0;
            bool won = false;
            if (occ && ebase + rank < tile.load_limit) {
                sidx = tile_hash(tile, rkey[u]);
                for (int probes = 0; probes < kTileMaxProbes; ++probes) {
                    if (atomicCAS(&tile.keys[sidx], kTileEmpty, rkey[u]) == kTileEmpty) {  // (no other thread enters this key)
                        won = true;
                        break;
                    }
                    sidx = (sidx + 1) & (unsigned)tile.slots_mask;
                }
            }
            bool failed = occ && !won;  // table full
            // (offsets only: the points arrive in phase 3, when the set is dead.  The window phase fills 5/8 of the store at
            // most: the rest is for the voxels the searches turn out to visit -- wide_serve_items)
            const bool fits = won && (unsigned)(off + cnt) * 24u <= prefill_top && off + cnt <= 0xFFFF;
            const bool want = fits && (unsigned)blk < 0x1000000u;
            const unsigned long long jm = __ballot(want);
            int jbase = 0;
            if (jm != 0ull) {
                if (lane64 == 0) jbase = atomicAdd(&sh.job_count, (int)__popcll(jm));
                jbase = __shfl(jbase, 0, 64);
            }
            const int job = jbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(jm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)jm, 0u));
            const bool filed = want && job < kWideJobs;
            if (filed) {
                jobs[2 * job] = (unsigned)blk | ((unsigned)cnt << 24);
                jobs[2 * job + 1] = (unsigned)off | (sidx << 16);
            }
            const unsigned long long fm = __ballot(filed);
            if (fm != 0ull) {  // the store's end: offsets ascend with the lane, so the highest lane that filed holds it
                const int end = __shfl(off + cnt, 63 - (int)__clzll(fm), 64);
                if (lane64 == 0) atomicMax(tile.stored, end);
            }
            if (won && !filed) {
                // LDS store (or the job list) full: the table remembers where the voxel is in the map instead
                const unsigned val = (unsigned)blk < 0x1000000u ? ((unsigned)blk | ((unsigned)cnt << 24) | kTileGlobal | kTileReady) : kTileOverflow;
                __hip_atomic_store(&tile.vals[sidx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                failed = val == kTileOverflow;
            }
            if (failed) {  // the queries whose windows hold this cell must search the map directly (phase 4)
                const int f = atomicAdd(&sh.bulk_failed, 1);
                if (f < kBulkFailMax) sh.bulk_fail_keys[f] = rkey[u];
            }
        }
    }
    __syncthreads();
    }
    stamp(1);
    // ---- 3: the points of the voxels won: a thread per POINT of the store (owner[p]: the job whose voxel point p belongs to;
    // the owner map lies where the set was), or -- a store that reaches up there -- a 32-lane group per voxel
    const int n_jobs = min(sh.job_count, kWideJobs);
    const int s1 = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned owner_bytes = ((unsigned)(s1 - s0) * 2u + 15u) & ~15u;
    if ((unsigned)s1 * 24u + owner_bytes <= free_top) {
        unsigned short *owner = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(tile.points) + free_top - owner_bytes);
        for (int p = s0 + tid; p < s1; p += kIcpThreads) owner[p - s0] = 0xFFFFu;  // (a voxel without a job leaves a gap)
        __syncthreads();
        for (int j = tid; j < n_jobs; j += kIcpThreads) {
            const unsigned w0 = jobs[2 * j], w1 = jobs[2 * j + 1];
            const int cnt = (int)(w0 >> 24), off = (int)(w1 & 0xFFFFu) - s0;
            for (int i = 0; i < cnt; ++i) owner[off + i] = (unsigned short)j;
        }
        __syncthreads();
        constexpr int kPerThread = 4;  // points a thread keeps in flight: 2048 per trip
        for (int p0 = s0 + tid; p0 < s1; p0 += kIcpThreads * kPerThread) {
            double2 xy[kPerThread];
            double zz[kPerThread];
            bool ok[kPerThread];
#pragma unroll
            for (int u = 0; u < kPerThread; ++u) {
                const int p = p0 + u * kIcpThreads;
                ok[u] = false;
                if (p < s1) {
                    const unsigned j = owner[p - s0];
                    if (j != 0xFFFFu) {
                        const unsigned w0 = jobs[2 * j], w1 = jobs[2 * j + 1];
                        const int blk = (int)(w0 & 0xFFFFFFu), i = p - (int)(w1 & 0xFFFFu);
                        xy[u] = block_xy(m, blk)[i];
                        zz[u] = block_z(m, blk)[i];
                        ok[u] = true;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kPerThread; ++u)
                if (ok[u]) {
                    double *q = tile.points + 3 * (p0 + u * kIcpThreads);
                    q[0] = xy[u].x;
                    q[1] = xy[u].y;
                    q[2] = zz[u];
                }
        }
    } else {
        constexpr int kFly = 4;   // voxels a 32-lane group keeps in flight (lane i fetches point i: one 16-byte and one 8-byte load)
        const int lane = tid & (kIcpGroup - 1), grp = tid / kIcpGroup;
        for (int j0 = grp; j0 < n_jobs; j0 += kIcpGroupsPerBlock * kFly) {  // job j0 + 16 u: the groups share every trip evenly
            double2 xy[kFly];
            double zz[kFly];
            int dst[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                dst[u] = -1;
                const int j = j0 + u * kIcpGroupsPerBlock;
                if (j < n_jobs) {
                    const unsigned w0 = jobs[2 * j], w1 = jobs[2 * j + 1];
                    const int blk = (int)(w0 & 0xFFFFFFu), cnt = (int)(w0 >> 24), off = (int)(w1 & 0xFFFFu);
                    if (lane < cnt) {
                        dst[u] = off + lane;
                        xy[u] = block_xy(m, blk)[lane];
                        zz[u] = block_z(m, blk)[lane];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kFly; ++u)
                if (dst[u] >= 0) {
                    double *q = tile.points + 3 * dst[u];
                    q[0] = xy[u].x;
                    q[1] = xy[u].y;
                    q[2] = zz[u];
                }
        }
    }
    __syncthreads();  // the points are in the store before their table entries say so
    for (int j = tid; j < n_jobs; j += kIcpThreads) {
        const unsigned w0 = jobs[2 * j], w1 = jobs[2 * j + 1];
        __hip_atomic_store(&tile.vals[w1 >> 16], (w1 & 0xFFFFu) | (w0 & 0xFF000000u) | kTileReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    stamp(2);
    // ---- 4: the queries' verdicts ----------------------------------------------------------------------------------------
    if (sh.bulk_failed) {  // rare: which windows hold a cell that could not be entered
        const int nf = sh.bulk_failed;
        for_each_cell([&](int qt, unsigned rkey, bool) {
            bool hit = nf > kBulkFailMax;
            for (int f = 0; f < min(nf, kBulkFailMax); ++f) hit = hit || sh.bulk_fail_keys[f] == rkey;
            if (hit) metas[qt].valid = -1;
        });
        __syncthreads();
    }
    if (mine) {  // 1: the window is in the tile; -1: it cannot be; 0: not settled here (the caller establishes it on its own)
        WideMeta *meta = metas + tid;
        meta->valid = meta->valid == -1 ? -1 : (meta->list_state == 3 ? 0 : 1);
        if (meta->valid == 0) sh.bulk_ticks[7] = 1u;
        meta->list_state = 0;
    }
    if (range_err) *range_err_out = 1;
    if (tid == 0) {
        sh.job_count = sh.bulk_failed = 0;  // (the caller's queue counters again)
        // the store's allocator goes on where the points end (it has counted every voxel that asked, also those without room)
        *tile.count = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    stamp(3);
    return 1;
}

}  // namespace kicp
