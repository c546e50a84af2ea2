// kicp_icp.hip -- k_icp: Registration::AlignPointsToMap (core/Registration.cpp:138-167)
//   = TransformPoints (:55-58) + DataAssociation (:60-78) + VoxelHashMap::GetClosestNeighbor
//     (core/VoxelHashMap.cpp:46-70) + BuildLinearSystem (:80-121) + LDLT solve / SE3::exp update
//     (:156-163), the whole <=500-iteration loop in ONE persistent launch.
//
// No MFMA anywhere: the normal equations are a 16-scalar f64 reduction per point (~0.4 flop/byte),
// not a dense contraction.  The work is HBM/L2-latency bound; what matters is one aligned 16-byte
// load per hash probe, contiguous voxel blocks, 32 lanes cooperating on each query, wave-level
// (DPP) + LDS reductions, and no host round trip inside the ICP loop.
#include <mutex>

#include "kicp_search.hpp"

namespace kicp {

// ------------------------------------------------------------------------------------------
// k_icp: the whole ICP loop of AlignPointsToMap in one persistent launch
//
// grid = G workgroups (all co-resident, G <= 256 = one per CU, each owning its CU's 160 KiB of LDS), 512 threads =
// 16 groups of 32 lanes.  A workgroup serves a contiguous RUN of the spatially sorted source cloud (kicp_sort.hip)
// and keeps the map voxels that run can reach in an LDS tile (kicp_search.hpp).  Per iteration, in three phases per
// chunk of 128 points:
//   A  one thread per point: s = est * s (TransformPoints, Registration.cpp:159; est = initial_guess for the first
//      iteration, :147), its voxel, is its known window still good;
//   B  one 32-lane group per point (pulled dynamically): (nn, d) = closest neighbour among the 27 voxels, against the
//      tile (VoxelHashMap.cpp:46-70), after the queries that left their window have re-established it (B0);
//   C  one thread per point: keep iff d < max_dist (strict, :72), w = sigma^2 / (sigma + |r|^2)^2, the 16 unique
//      scalars of J^T w J and J^T w r with J = [I | -hat(s)], r = s - nn (:81-98), added in point order.
// The exchange, once per iteration (tagged 16-byte granule pairs, sc1 stores and loads: the data is its own flag):
//   1. every workgroup publishes its 18 partial sums (its 16 groups' sums of a term added by four DPP row operations, stored by the
//      lane that holds them), a block of 384 bytes -- three lines nobody else writes;
//   2. workgroup g < 16 (a "leader") gathers the partials of the workgroups b = g, g + 16, g + 32, ... (at most 16),
//      sums them in that order and publishes the group's sums, in eight copies (a workgroup reads copy b mod 8);
//   3. EVERY workgroup gathers the (at most) 16 group sums, adds them in order, and solves the same 6x6 system on its
//      first four waves: dx = LDLT(JTJ).solve(-JTr), est = exp(dx), stop when |dx| < convergence_criterion
//      (:156-163); workgroup 0 also keeps T_icp = est * T_icp and the statistics.
// Two short hops (14 x 288 B into 16 CUs, then 16 x 288 B into every CU) instead of one long one (224 x 288 B into ONE
// CU's memory queue) plus a broadcast hop for the result; the summation tree depends on G only, never on timing or
// placement, so results are reproducible bit for bit.
// ------------------------------------------------------------------------------------------
// one source point of the chunk a workgroup is working on (phases A -> B -> C of an iteration)
struct IcpPoint {
    double s[3];   // transformed source point                               (A)
    double nn[3];  // its closest map point                                   (B)
    double d2;     // squared distance, DBL_MAX when the neighbourhood is empty (B)
    int E;         // map points the reference examines for it                (B)
    int flag;      // 0 staged window valid, 1 window must be (re)staged, 2 search HBM directly (A)
    int v[3];      // voxel of s                                              (A)
    int pad;
};
static_assert(sizeof(IcpPoint) == 80, "IcpPoint layout");

constexpr int kBulkFailMax = 30;  // failed cells remembered one by one; more: every query of the chunk searches the map directly
struct alignas(16) IcpShared {  // head of the dynamic LDS; the region records and the candidate pool follow
    double part[kIcpGroupsPerBlock][kIcpSums];
    double range_sum[kIcpSumRows][kIcpSums];  // leader: its members' partials; every workgroup: the leaders' group sums
    double tot[kIcpSums];
    double est[8];  // q[4], t[3], |dx|
    double far_point[4];  // a point so far away that its squared distance to anything overflows to +inf (tile_scan_list's tail)
    // kept by ONE thread (kIcpBookThread of workgroup 0), off the critical path and out of registers:
    double T_icp[7];  // accumulated update, q[4] t[3]
    double guess[7];
    double book[8];   // pipeline mode: last_pose q[4] t[3] as the prologue read it (the frame's bookkeeping at the end of the launch
                      // takes it from here: a round trip to memory on the frame's serial chain otherwise)
    unsigned long long ncorr_last, ncorr_total, examined_total;
    double pad2;
    int fail;
    int tile_points;  // points asked of the tile's store (the demand; what does not fit stays in the map)
    int tile_stored;  // end of the points kept in the store
    int next_point;   // phase B: next unserved point of the chunk
    int origin[3];    // voxel with relative tile coordinates (0, 0, 0)
    int any_fill;     // phase A: some query of the chunk is outside its known window
    int tile_entries;  // occupied slots of the tile's table
    int list_entries; // entries handed out from the scan-list pool
    int run[2];       // first sorted position of this workgroup's run, and of the next workgroup's
    int job_count;    // tile_fill_bulk: point-fetch jobs filed so far
    int cell_count;   // tile_fill_bulk: distinct cells to look up
    int bulk_failed;  // tile_fill_bulk: cells that could not be entered (table full / block id beyond 24 bits)
    unsigned bulk_ticks[8];  // (profiling build) tile_fill_bulk's phases, 10 ns ticks (5 used)
    unsigned bulk_fail_keys[kBulkFailMax];
    int search_count;                        // group form, phase B: points of the chunk that need a search ...
    int pad3[3];
    unsigned char search_idx[kIcpChunk];     // ... and which (filed by phase A in any order: nothing depends on who serves which)
    double terms[kIcpTermChunk][kIcpTerms];  // phase C: the products of kIcpTermChunk points
    IcpPoint pts[kIcpChunk];
};
static_assert(offsetof(IcpShared, part) % 16 == 0 && offsetof(IcpShared, range_sum) % 16 == 0 && offsetof(IcpShared, terms) % 16 == 0 &&
                  (kIcpTerms * sizeof(double)) % 16 == 0,
              "rows that are read and written sixteen bytes at a time (icp_row_sum, phase C)");
static_assert(sizeof(IcpShared) % 16 == 0 && offsetof(IcpShared, pts) % 16 == 0 && sizeof(IcpPoint) % 16 == 0,
              "the query records behind the point slots must stay 16-byte aligned");

// ------------------------------------------------------------------------------------------
// ---- runs of equal WEIGHT, settled here, inside the launch ---------------------------------------------------
// A workgroup's tile must hold the map voxels its run can reach, and the map is far from uniform: next to the
// sensor voxels are full, far away they hold a point or two.  Runs of equal LENGTH would need tiles of very
// different sizes, and an iteration is as slow as its slowest workgroup; so runs are cut to equal weight, a
// point weighing  base + c (+ c^2 / quad) + max(0, E - dense_min) / dense_div,  c = the population of the map voxel it
// falls in under the initial guess, E = the population of the 27 voxels around it (what the reference examines for
// it).  The last term is what keeps the densest runs inside LDS: c saturates at max_points_per_voxel, E tells a point
// whose whole neighbourhood is full (a tile of 500 points for it alone) from one at the edge of the map.  (Until round 3 a single-workgroup kernel in front of this one computed the weights and their prefix:
// 19 us + a dispatch gap on the frame's serial chain.)  Here: workgroup b weighs the b-th slice of L points of
// the sorted cloud (one map lookup per point, all in flight together), publishes the weights and their sum as
// tagged granules; every workgroup gathers the G sums -- one hop -- and reads the two slices its own run
// boundaries fall into.  The point at sorted position q goes to workgroup floor(E[q] G / W), E the exclusive
// prefix of the weights, W their total: integer arithmetic on data, so the partition never depends on timing.
// Leaves the two ends of this workgroup's run in sh.run; false: a bounded wait gave up (the workgroups are not all
// resident).  (Inlined, like the window phase below: as functions of their own the two cost 3-6 us per call -- the
// registers live around the call go to scratch memory and back, 11 MB of writes per launch -- profiles/r03_t, r03_ac.)
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// The first iteration's window phase, workgroup-wide (tile_fill, kicp_search.hpp, does the same query by query and stays
// in use for the occasional query that leaves its window later, and for chunks whose region has no room for this
// routine's scratch).  In the first iteration EVERY query must establish its window, and query by query that is a chain
// of dependent memory round trips per query -- lookups, then the points of the voxels won, eight voxels per trip -- over
// two or three rounds of 16 groups: 21 us on average, 37 at worst (profiles/r03_q_icp_probe_steady.txt).  Here, between
// barriers: (1) a thread per query writes its window; (2a) a thread per window cell enters the cell into a set of
// DISTINCT cells in LDS (neighbouring queries share most of theirs); (2b) a thread per distinct cell looks it up in the
// map -- one wave of loads for the workgroup -- and enters the occupied ones into the tile's table, filing a fetch job
// each; (3) a thread per POINT of the store fetches it -- one more wave; (4) the entries are published and the queries
// get their verdicts.  10 us on average, 18 at worst (profiles/r03_ac_icp_probe_steady.txt).  Which voxels end up in LDS
// and which stay "global" when the store is full may differ from the query-by-query order; results never depend on
// that (tests/test_gpu_paths.py::test_first_iteration_window_phase_is_bitwise_neutral).
// (LDS is 160 KiB: the store holds fewer than 7 k points, so the 12 k entries of the owner map and the 16-bit job
// and slot indices below always suffice.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ticks32() { return (unsigned)wall_clock64(); }
constexpr int kBulkSetLog2 = 12, kBulkSet = 1 << kBulkSetLog2;  // slots of the set of distinct cells (a chunk has at most 64 x 64 cell instances)
constexpr int kBulkSetBytes = kBulkSet * 6;                 // the set (u32) and the list of its members (u16)
constexpr int kBulkJobs = (int)(sizeof(double) * kIcpTermChunk * kIcpTerms / 8);  // jobs that fit into sh.terms (8 bytes each)
__device__ __forceinline__ bool tile_fill_bulk(MapView m, Tile tile, IcpShared *shp, int cn, IcpQueryMeta *metas, int *range_err_out, bool prof) {
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
    const int s0 = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // the store's end before this chunk
    // The region holds points from the bottom and (a later chunk of a long run) the scan lists of the chunks before from
    // the top; neither moves during a fill phase.  This routine's scratch -- the set of distinct cells, the list of its
    // members, later the owner map -- goes right below the lists, and needs that much room above the points.
    const unsigned lists_bytes = tile.lists ? min(2u * (unsigned)__hip_atomic_load(tile.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), tile.region_bytes) : 0u;
    const unsigned free_top = (tile.region_bytes - lists_bytes) & ~15u;  // points end here at the latest
    if ((unsigned)s0 * 24u + (unsigned)kBulkSetBytes > free_top) return false;  // (the whole workgroup: query by query then)
    // ---- 1: the windows (one thread per query); the cell set is cleared ---------------------------------------------------
    // The set of distinct cells and the list of its members live at the top of the tile's point region, which the
    // points of phase 3 may overwrite: by then both are dead.
    unsigned *set = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(tile.points) + free_top) - kBulkSet;
    unsigned short *cells = reinterpret_cast<unsigned short *>(set) - kBulkSet;
    for (int i = tid; i < kBulkSet; i += kIcpThreads) set[i] = kTileEmpty;
    if (tid == 0) sh.job_count = sh.cell_count = sh.bulk_failed = 0;
    if (tid < cn && sh.pts[tid].flag == 1) {
        const IcpPoint &pt = sh.pts[tid];
        IcpQueryMeta *meta = metas + tid;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double f = pt.s[a] / m.voxel_size - (double)pt.v[a];  // position inside the voxel, [0, 1)
            meta->v[a] = pt.v[a];
            meta->lo[a] = (signed char)((f < kWindowMargin) ? -2 : -1);
            meta->hi[a] = (signed char)((f > 1.0 - kWindowMargin) ? 2 : 1);
        }
        meta->valid = 0;       // pending; -1 as soon as any of its cells cannot be served from the tile
        meta->list_state = 0;  // whatever list there was belongs to the old window
    }
    __syncthreads();
    // ---- 2a: the DISTINCT cells of all windows that the table does not know yet (LDS only).  Neighbouring queries share
    // most of their cells: looked up query by query the map would be asked five times for the same voxel.
    auto for_each_cell = [&](auto &&fn) {  // fn(query, relative key) for every in-range cell of every pending window
        for (int idx = tid; idx < cn * 64; idx += kIcpThreads) {
            const int qt = idx >> 6;
            // (the query's flag and the window part of its record in one round trip, in front of the test)
            const int qflag = sh.pts[qt].flag;
            struct { int v[3]; signed char lo[3], hi[3], valid, list_state; } Mq;
            static_assert(sizeof(Mq) == 20 && offsetof(IcpQueryMeta, v) == 24 && offsetof(IcpQueryMeta, lo) == 36 && offsetof(IcpQueryMeta, hi) == 39, "window part of IcpQueryMeta");
            __builtin_memcpy(&Mq, __builtin_assume_aligned(reinterpret_cast<const char *>(metas + qt) + offsetof(IcpQueryMeta, v), 4), sizeof Mq);
            if (qflag != 1) continue;
            const auto *meta = &Mq;
            const int ny = meta->hi[1] - meta->lo[1] + 1, nz = meta->hi[2] - meta->lo[2] + 1, nx = meta->hi[0] - meta->lo[0] + 1;
            const int w = idx & 63;
            if (w >= nx * ny * nz) continue;
            // (window sides are 3 or 4 cells, w < 64: a shift or a multiplication instead of four integer divisions)
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
            fn(qt, rkey);
        }
    };
    const bool fresh = __hip_atomic_load(tile.entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;  // (the run's first chunk)
    for_each_cell([&](int qt, unsigned rkey) {
        const int slot = fresh ? -1 : tile_find(tile, rkey);
        if (slot >= 0) {
            if (__hip_atomic_load(&tile.vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kTileOverflow) metas[qt].valid = -1;
            return;
        }
        unsigned s = (rkey * 0x9E3779B1u) >> (32 - kBulkSetLog2);
        bool done = false;
        for (int probes = 0; probes < 64; ++probes) {
            const unsigned old = atomicCAS(&set[s], kTileEmpty, rkey);
            // (one counter update per new cell: kept wave by wave -- ballot, leader, shuffle inside this divergent loop --
            // the phase was half as slow again, 3.7 against 2.4 us, profiles/r03_aa)
            if (old == kTileEmpty) cells[atomicAdd(&sh.cell_count, 1)] = (unsigned short)s;
            if (old == kTileEmpty || old == rkey) {
                done = true;
                break;
            }
            s = (s + 1) & (unsigned)(kBulkSet - 1);
        }
        if (!done) metas[qt].valid = -1;  // (a set this crowded: thousands of distinct cells in one chunk)
    });
    __syncthreads();
    stamp(0);
    // ---- 2b: one map lookup per distinct cell, all in flight together; occupied voxels enter the table and file a fetch job
    const int n_cells = sh.cell_count;
    constexpr int kBatch = 3;  // lookups a thread keeps in flight (four: the kernel starts to spill)
    for (int jb = 0; jb < n_cells; jb += kIcpThreads * kBatch) {  // (every thread takes every trip: the counters below are kept wave by wave)
        const int j0 = jb + tid;
        unsigned long long key[kBatch];
        unsigned rkey[kBatch];
        Slot a[kBatch][kProbeAhead];
        uint32_t hs[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int j = j0 + u * kIcpThreads;
            rkey[u] = kTileEmpty;
            key[u] = 0;
            hs[u] = 0;
            if (j < n_cells) {
                rkey[u] = set[cells[j]];
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
        if (prof) {  // (profiling build, and only while the loop has a single trip: the lookups apart from the entering)
            if (n_cells <= kIcpThreads * kBatch) {
                __syncthreads();
                stamp(4);
            } else if (tid == 0) {
                sh.bulk_ticks[4] = 0;
            }
        }
        // Entering: the table slot is a CAS of the lane's own; the counters (slots occupied, points asked of the store, jobs
        // filed, the store's end) are ONE LDS atomic per wave and counter -- 64 lanes adding to the same address one by
        // one were the longest part of this phase (8 of 11 us in a workgroup with 1500 cells, profiles/r03_z).
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
            const bool fits = won && (unsigned)(off + cnt) * 24u <= free_top && off + cnt <= 0xFFFF;
            const bool want = fits && (unsigned)blk < 0x1000000u;
            const unsigned long long jm = __ballot(want);
            int jbase = 0;
            if (jm != 0ull) {
                if (lane64 == 0) jbase = atomicAdd(&sh.job_count, (int)__popcll(jm));
                jbase = __shfl(jbase, 0, 64);
            }
            const int job = jbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(jm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)jm, 0u));
            const bool filed = want && job < kBulkJobs;
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
    stamp(1);
    // ---- 3: the points of the voxels won, in one or two memory round trips for the whole workgroup.  Far from the sensor a
    // voxel holds a point or two: fetched voxel by voxel (a 32-lane group per voxel, below) a workgroup with 400 such
    // voxels needed three trips of 4 us.  So: a thread per POINT of the store -- owner[p] names the job whose voxel point p
    // belongs to (written by a thread per job), consecutive lanes read consecutive points of a block -- which makes every
    // load instruction 64 points whatever the voxels' populations.  The owner map lies at the top of the region, where
    // the cell set was; a store that reaches up there (dense voxels next to the sensor) is filled voxel by voxel instead.
    const int n_jobs = min(sh.job_count, kBulkJobs);
    const int s1 = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((unsigned)s1 * 24u + (unsigned)kBulkSetBytes <= free_top) {
        unsigned short *owner = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(tile.points) + free_top - kBulkSetBytes);
        for (int p = s0 + tid; p < s1; p += kIcpThreads) owner[p - s0] = 0xFFFFu;  // (a voxel without a job leaves a gap)
        __syncthreads();
        for (int j = tid; j < n_jobs; j += kIcpThreads) {
            const unsigned w0 = jobs[2 * j], w1 = jobs[2 * j + 1];
            const int cnt = (int)(w0 >> 24), off = (int)(w1 & 0xFFFFu) - s0;
            for (int i = 0; i < cnt; ++i) owner[off + i] = (unsigned short)j;
        }
        __syncthreads();
        constexpr int kPerThread = 6;  // points a thread keeps in flight: 3072 per trip
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
                        const int blk = (int)(w0 & 0xFFFFFFu), i = KICP_IDX(m.dbg, m.ctr + C_ERR, p - (int)(w1 & 0xFFFFu), m.max_points, 5);
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
        constexpr int kFly = 12;  // voxels a 32-lane group keeps in flight (lane i fetches point i: one 16-byte and one 8-byte load)
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
        for_each_cell([&](int qt, unsigned rkey) {
            bool hit = nf > kBulkFailMax;
            for (int f = 0; f < min(nf, kBulkFailMax); ++f) hit = hit || sh.bulk_fail_keys[f] == rkey;
            if (hit) metas[qt].valid = -1;
        });
        __syncthreads();
    }
    if (tid < cn && sh.pts[tid].flag == 1) {
        IcpQueryMeta *meta = metas + tid;
        const bool ok = meta->valid != -1;
        meta->valid = ok ? 1 : -1;
        sh.pts[tid].flag = ok ? 0 : 2;
    }
    if (range_err) *range_err_out = 1;
    __syncthreads();
    stamp(3);
    return true;
}

struct IcpRunArgs {  // (by value: a reference would pin the kernel's parameter block and the guess in scratch memory)
    const double *frame;
    const unsigned long long *order;
    unsigned long long *wts, *granules;
    const Slot *slots;
    uint32_t mask;
    double voxel_size;
    PipeState *state;
    int weight_base, weight_long_base, weight_long_emul, weight_quad, dense_min, dense_div;
    unsigned spin_limit;
    BoundsRec *dbg;
};
// THREAD_PER_POINT: long runs are weighed by a thread per point (the thread-per-query form of the kernel; the group form
// keeps its registers as they were)
template <bool THREAD_PER_POINT>
__device__ __forceinline__ bool icp_weighted_run(IcpRunArgs P, IcpShared *shp, SE3 guess, unsigned epoch_base, int n, int G) {
    IcpShared &sh = *shp;
    const int tid = threadIdx.x;
    MapView m;  // (only what a lookup reads)
    m.slots = const_cast<Slot *>(P.slots);
    m.mask = P.mask;
    m.voxel_size = P.voxel_size;
    PipeState *st = P.state;
    const int L = (n + G - 1) / G;
    const int s0 = min(n, (int)blockIdx.x * L), s1 = min(n, s0 + L);
    if (tid == 0) sh.run[0] = sh.run[1] = 0;  // (a barrier follows before anybody reads or sets them)
    // Two regimes, both measured: with a few dozen points per run (full-size voxels: at most 64 points per workgroup) a
    // workgroup's time is whether its tile fits and the number of 16-point rounds -> base + c + c^2 / 10, plus the
    // population of the whole neighbourhood where that is large (profiles/r03_n).  The base (128, with the last term
    // undivided) is what bounds the LENGTH of the sparse runs: a search costs ~1.3 + 0.85 us per round of 16 points
    // whatever the tile holds, and with base 32 the sparsest runs were 41-51 points -- four rounds against one for the
    // dense ones (profiles/r03_ae: 16.4 -> 15.2 us per iteration).  With hundreds of points per run (the
    // 1M-point / 0.1 m configuration) the per-point work dominates, and that is the neighbourhood's population E -> a
    // larger base + c + E (462 -> 547 scans/s against the short-run rule, profiles/r03_p).
    const bool long_runs = n > kIcpListRunMax * G;
    const int quad = P.weight_quad >= 0 ? P.weight_quad : (long_runs ? 0 : 10);
    const int w_base = long_runs ? P.weight_long_base : P.weight_base;
    const int dense_min = long_runs ? 0 : P.dense_min, dense_div = long_runs ? 1 : P.dense_div;
    long long *x_pref = reinterpret_cast<long long *>(sh.range_sum);  // [G + 2]: exclusive prefix of the slice sums, then the delta
    static_assert((kIcpMaxBlocks + 2) * sizeof(long long) <= sizeof(IcpShared::range_sum), "x_pref lives in range_sum");
    long long my_sum = 0;
    const int lane = tid & (kIcpGroup - 1);
    // Long runs (hundreds of points per slice): a THREAD per point, its 27 lookups nine at a time (the registers are there: the kernel's allocation is the iterations') -- served 16 points at a
    // time by the groups below, 420 points were 26 dependent rounds of lookups, ~100 us of every launch of the 1M-point
    // configuration (profiles/r04_r_icp_probe_livox100.txt).  Same c, same E, same weight.
    if (THREAD_PER_POINT && long_runs) {
        for (int q = s0 + tid; q < s1; q += kIcpThreads) {
            const int p = KICP_IDX(P.dbg, &st->err, key_index(P.order[q]), n, 1);
            const double pin[3] = {P.frame[3 * p], P.frame[3 * p + 1], P.frame[3 * p + 2]};
            double sp[3];
            se3_act(guess, pin, sp);
            const int vx = voxel_coord(sp[0], m.voxel_size), vy = voxel_coord(sp[1], m.voxel_size), vz = voxel_coord(sp[2], m.voxel_size);
            int c = 0, E = 0;
            // (seven lookups in flight -- 4 slots of 16 bytes each: 112 registers; nine at a time were the last four registers
            // this form of the kernel could not hold)
            constexpr int kAhead = 7;
#pragma unroll
            for (int jb = 0; jb < 27; jb += kAhead) {
                unsigned long long key[kAhead];
                uint32_t hs[kAhead];
                bool ok[kAhead];
                Slot a[kAhead][kProbeAhead];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const int j = jb + u < 27 ? jb + u : 26;
                    const int qx = vx + (int)((kShift.x >> (2 * j)) & 3) - 1, qy = vy + (int)((kShift.y >> (2 * j)) & 3) - 1, qz = vz + (int)((kShift.z >> (2 * j)) & 3) - 1;
                    ok[u] = jb + u < 27 && voxel_in_range(qx, qy, qz);
                    key[u] = pack_voxel(qx, qy, qz);
                    hs[u] = hash_key(key[u], m.mask);
#pragma unroll
                    for (int i = 0; i < kProbeAhead; ++i) {
                        a[u][i].key = kKeyEmpty;
                        a[u][i].block = -1;
                        a[u][i].count = 0;
                        if (ok[u]) a[u][i] = load_slot(m.slots + ((hs[u] + i) & m.mask));
                    }
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    int blk, cnt;
                    if (!probe_resolve(a[u], key[u], blk, cnt)) probe_tail(m, (hs[u] + kProbeAhead) & m.mask, key[u], blk, cnt);
                    if (blk < 0 || jb + u >= 27) cnt = 0;
                    E += cnt;
                    if (jb + u == 0) c = cnt;  // the point's own voxel (shift 0 of the table)
                }
            }
            const int dense = (dense_div > 0 ? max(0, E - dense_min) / dense_div : 0) * P.weight_long_emul;
            const int w = w_base + c + (quad > 0 ? (c * c) / quad : 0) + dense;
            granule_store(P.wts + q, epoch_base, (unsigned)w);
            my_sum += w;
        }
    }
    // (short runs) one 32-lane group per point: lane j looks up the j-th voxel of the point's 27-neighbourhood (all in flight together)
    for (int q = s0 + tid / kIcpGroup; !(THREAD_PER_POINT && long_runs) && q < s1; q += kIcpGroupsPerBlock) {
        const int p = KICP_IDX(P.dbg, &st->err, key_index(P.order[q]), n, 2);
        const double pin[3] = {P.frame[3 * p], P.frame[3 * p + 1], P.frame[3 * p + 2]};
        double sp[3];
        se3_act(guess, pin, sp);
        int rerr = 0;
        const Probe pr = probe27(m, sp[0], sp[1], sp[2], lane, rerr);
        const int c = __shfl(pr.cnt, 0, kIcpGroup);  // the point's own voxel (shift 0 of the table)
        const int dense = (dense_div > 0 ? max(0, pr.E - dense_min) / dense_div : 0) * (long_runs ? P.weight_long_emul : 1);
        const int w = w_base + c + (quad > 0 ? (c * c) / quad : 0) + dense;
        if (lane == 0) {
            granule_store(P.wts + q, epoch_base, (unsigned)w);
            my_sum += w;
        }
    }
    // workgroup sum (integers: any order)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_sum += __shfl_xor(my_sum, o, 64);
    if ((tid & 63) == 0) reinterpret_cast<long long *>(sh.part)[tid >> 6] = my_sum;
    __syncthreads();
    unsigned long long *sum_gran = P.granules + (size_t)G * (2 * kIcpGranStride);  // the partials' buffer of odd iterations; first used by iteration 1
    const __amdgpu_buffer_rsrc_t sum_rsrc = granule_rsrc(sum_gran, (unsigned)(G * 2 * kIcpGranStride * sizeof(unsigned long long)));
    if (tid == 0) {
        long long t = 0;
#pragma unroll
        for (int w = 0; w < kIcpThreads / 64; ++w) t += reinterpret_cast<long long *>(sh.part)[w];
        granule_store_pair(sum_rsrc, (unsigned)(((size_t)blockIdx.x * kIcpGranStride) * 16), epoch_base, (unsigned)(unsigned long long)t,
                           (unsigned)((unsigned long long)t >> 32));
    }
    bool pfail = false;
    if (tid < G) {  // thread t fetches the sum of slice t
        unsigned long long lo, hi;
        const unsigned off = (unsigned)(((size_t)tid * kIcpGranStride) * 16);
        granule_load_pair(sum_rsrc, off, lo, hi);
        unsigned spins = 0;
        while ((unsigned)(lo >> 32) != epoch_base || (unsigned)(hi >> 32) != epoch_base) {
            if (++spins > P.spin_limit || ((spins & 255u) == 0 && (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & E_TIMEOUT))) {
                pfail = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            granule_load_pair(sum_rsrc, off, lo, hi);
        }
        x_pref[tid + 1] = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
    if (tid == 0) x_pref[0] = 0;
    if (pfail) sh.run[0] = -1;  // (no __syncthreads_or: it brings static LDS, and this kernel's 160 KiB are all dynamic)
    __syncthreads();
    if (sh.run[0] < 0) {  // the workgroups are not all resident: give up like a failed exchange
        return false;
    }
    if (tid == 0) {
        for (int g = 0; g < G; ++g) x_pref[g + 1] += x_pref[g];  // (G <= 256 additions)
        // Long runs: no run longer than kRunCap points (the thread-per-query form carries one chunk of 512 queries through
        // the iterations in registers; a run is at most total weight / (G x smallest weight) points long).  Every weight
        // is raised by the same delta -- E'[q] = E[q] + delta q -- just enough for that bound; integer arithmetic on
        // data again.  (x_pref[G + 1]: the delta, for the scan of the boundary slices below.)
        long long delta = 0;
        constexpr long long kRunCap = 500;
        if (long_runs && kRunCap * G > (long long)n) {
            const long long W = x_pref[G], need = W - kRunCap * G * (long long)w_base, room = kRunCap * G - (long long)n;
            if (need > 0) delta = (need + room - 1) / room;
        }
        if (delta > 0)
            for (int g = 0; g <= G; ++g) x_pref[g] += delta * (long long)min(n, g * L);
        x_pref[G + 1] = delta;
    }
    __syncthreads();
    const long long w_delta = x_pref[G + 1];
    if (tid < 128) {  // wave 0: where this run starts; wave 1: where the next one does
        const int which = tid >> 6, l = tid & 63;
        const long long W = x_pref[G];
        const long long target = ((long long)blockIdx.x + which) * W;  // position q lies in front of the boundary iff E[q] * G < target
        int result;
        if (which == 1 && (int)blockIdx.x == G - 1) {
            result = n;
        } else {
            int below = 0;  // slices whose first point lies in front of the boundary
            for (int g = l; g < G; g += 64) below += (x_pref[g] * (long long)G < target) ? 1 : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) below += __shfl_xor(below, o, 64);
            if (below == 0) {
                result = 0;
            } else {
                const int j = below - 1;  // the boundary falls into slice j: count its points in front of it
                const int j0 = min(n, j * L), j1 = min(n, j0 + L);
                long long run = x_pref[j];
                int count = 0;
                bool wfail = false;
                for (int qb = j0; qb < j1 && !wfail; qb += 64) {
                    const int q = qb + l;
                    long long w = 0;
                    if (q < j1) {
                        unsigned long long gq = granule_load(P.wts + q);
                        unsigned spins = 0;
                        while ((unsigned)(gq >> 32) != epoch_base) {
                            if (++spins > P.spin_limit) {
                                wfail = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            gq = granule_load(P.wts + q);
                        }
                        w = (long long)(unsigned)gq + w_delta;
                    }
                    wfail = __ballot(wfail) != 0ull;
                    long long incl = w;  // inclusive scan of the 64 weights
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const long long up = __shfl_up(incl, o, 64);
                        if (l >= o) incl += up;
                    }
                    const long long excl = run + incl - w;
                    count += __popcll(__ballot(q < j1 && excl * (long long)G < target));
                    run += __shfl(incl, 63, 64);
                }
                result = wfail ? -1 : j0 + count;
            }
        }
        if (l == 0) sh.run[which] = result;
    }
    __syncthreads();
    if (sh.run[0] < 0 || sh.run[1] < 0) {
        return false;
    }
    return true;
}


// ------------------------------------------------------------------------------------------
// The same weights, computed by a kernel of their own IN FRONT of the registration (option "icp_weights_kernel").
// Inside the launch a workgroup weighs its ~21 points sixteen at a time (two rounds of three dependent round trips to the
// map), waits for the slowest of 224 to publish its sum, adds 224 sums on one thread and scans a neighbour's slice: 16 us of
// every launch before the first association starts (profiles/r06_l_*: first iteration 46.6 us, 29.5 of them the iteration).
// k_icp_weights weighs all points at once -- a 32-lane group per point, every group its own lookups, nothing to wait for --
// and leaves plain 32-bit weights; every workgroup of k_icp then reads them all (19 KB), forms the prefix sums itself and finds
// its two boundaries: no exchange.  The weights, the rule that cuts the runs (position q lies in front of boundary b iff
// E[q] * G < b * W, E the exclusive prefix) and so the runs are the in-launch path's, integer for integer: same poses bit for bit
// (tests/test_gpu_paths.py::test_run_weights_from_their_own_kernel_give_the_same_runs).  Short runs only (the group form's
// regime and the thread-per-query form on small clouds); long runs keep the in-launch path with its cap on the run length.
// ------------------------------------------------------------------------------------------
struct IcpWeightArgs {
    const double *frame;
    const unsigned long long *order;
    unsigned *wts32;
    const int *n_ptr;
    int n_imm;
    MapView map;
    PipeState *state;
    int pipeline_mode;
    int icp_grid, force_blocks, points_per_group;  // what k_icp will derive its G from
    int weight_base, weight_quad, dense_min, dense_div;
};
__device__ __forceinline__ int icp_grid_of(int n, int force_blocks, int points_per_group, int grid) {
    int G = force_blocks > 0 ? force_blocks : (n + kIcpGroupsPerBlock * points_per_group - 1) / (kIcpGroupsPerBlock * points_per_group);
    return max(1, min(G, grid));
}
__global__ __launch_bounds__(256) void k_icp_weights(IcpWeightArgs A) {
    // (what the tests below need from memory is asked for in front of them, in one round trip: k_icp's prologue, and why)
    const int n = count_of(A.n_ptr, A.n_imm);
    const int live0 = A.map.ctr[C_LIVE];
    const int err0 = __hip_atomic_load(&A.state->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SE3 pose_a, pose_b;
    if (A.pipeline_mode) {
        pose_a = A.state->last_pose;
        pose_b = A.state->last_delta;
    } else {
        pose_a = pose_b = A.state->guess;
    }
    const int G = icp_grid_of(n, A.force_blocks, A.points_per_group, A.icp_grid);
    // (the cases in which k_icp does not cut weighted runs, or cuts long ones by its own prologue)
    if (n < kIcpWeightedMin || A.force_blocks > 0 || n > kIcpListRunMax * G || live0 == 0) return;
    if (err0 & E_TIMEOUT) return;
    const SE3 guess = A.pipeline_mode ? se3_mul(pose_a, pose_b) : pose_a;  // (k_icp's own expression)
    const int quad = A.weight_quad >= 0 ? A.weight_quad : 10;
    const int lane = threadIdx.x & (kIcpGroup - 1);
    const int groups = (int)(gridDim.x * blockDim.x) / kIcpGroup;
    for (int q = (int)(blockIdx.x * blockDim.x + threadIdx.x) / kIcpGroup; q < n; q += groups) {
        const int p = KICP_IDX(A.map.dbg, &A.state->err, key_index(A.order[q]), n, 3);
        const double pin[3] = {A.frame[3 * p], A.frame[3 * p + 1], A.frame[3 * p + 2]};
        double sp[3];
        se3_act(guess, pin, sp);
        int rerr = 0;
        const Probe pr = probe27(A.map, sp[0], sp[1], sp[2], lane, rerr);
        const int c = __shfl(pr.cnt, 0, kIcpGroup);  // the point's own voxel (shift 0 of the table)
        const int dense = A.dense_div > 0 ? max(0, pr.E - A.dense_min) / A.dense_div : 0;
        const int w = A.weight_base + c + (quad > 0 ? (c * c) / quad : 0) + dense;
        if (lane == 0) A.wts32[q] = (unsigned)w;
    }
}
void launch_icp_weights(const IcpParams &P, int icp_grid, size_t n_hint, hipStream_t s) {
    IcpWeightArgs A;
    A.frame = P.frame;
    A.order = P.order;
    A.wts32 = P.wts32;
    A.n_ptr = P.n_ptr;
    A.n_imm = P.n_imm;
    A.map = P.map;
    A.state = P.state;
    A.pipeline_mode = P.pipeline_mode;
    A.icp_grid = icp_grid;
    A.force_blocks = P.force_blocks;
    A.points_per_group = P.points_per_group;
    A.weight_base = P.weight_base;
    A.weight_quad = P.weight_quad;
    A.dense_min = P.weight_dense_min;
    A.dense_div = P.weight_dense_div;
    // a group per point of the cloud there will probably be (the loop covers any other count)
    size_t groups = n_hint + n_hint / 4 + 64;
    if (groups > (size_t)kIcpListRunMax * kIcpMaxBlocks) groups = (size_t)kIcpListRunMax * kIcpMaxBlocks;
    const int grid = (int)((groups * kIcpGroup + 255) / 256);
    hipLaunchKernelGGL(k_icp_weights, dim3(grid), dim3(256), 0, s, A);
}

// k_icp's side: this workgroup's run [sh.run[0], sh.run[1]) from the weights k_icp_weights left.  scratch: LDS for n 32-bit words
// (the tile's region: nothing lives there yet), or nullptr -- the weights are then read from memory twice.
__device__ __forceinline__ void icp_runs_from_weights(const unsigned *wts32, IcpShared *shp, unsigned *scratch, int n, int G) {
    IcpShared &sh = *shp;
    const int tid = threadIdx.x;
    const unsigned *w = wts32;
    if (scratch) {
        for (int i = tid; i < n; i += kIcpThreads) scratch[i] = wts32[i];  // (coalesced; every workgroup reads all of them: 19 KB)
        __syncthreads();
        w = scratch;
    }
    // thread t takes the contiguous piece [t per, (t + 1) per): its sum, then the exclusive prefix of the 512 sums
    const int per = (n + kIcpThreads - 1) / kIcpThreads;
    const int a = min(n, tid * per), b = min(n, a + per);
    // (a thread's piece four words per round trip, here and below: ten dependent ones per loop otherwise)
    long long mine = 0;
    for (int i = a; i < b; i += 4) {
        unsigned x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = w[i + u < b ? i + u : a];
#pragma unroll
        for (int u = 0; u < 4; ++u) mine += i + u < b ? (long long)x[u] : 0ll;
    }
    long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += up;
    }
    long long *wave_tot = reinterpret_cast<long long *>(sh.part);  // [8]
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    if (tid < 2) sh.run[tid] = 0;
    __syncthreads();
    long long before = 0, W = 0;
#pragma unroll
    for (int v = 0; v < kIcpThreads / 64; ++v) {
        const long long t = wave_tot[v];
        before += v < (tid >> 6) ? t : 0;
        W += t;
    }
    // position q lies in front of boundary `which` iff E[q] * G < (blockIdx.x + which) * W, E the exclusive prefix (icp_weighted_run)
    const long long t0 = (long long)blockIdx.x * W, t1 = ((long long)blockIdx.x + 1) * W;
    long long E = before + incl - mine;
    int c0 = 0, c1 = 0;
    for (int i = a; i < b; i += 4) {
        unsigned x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = w[i + u < b ? i + u : a];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = i + u < b;
            c0 += (in && E * (long long)G < t0) ? 1 : 0;
            c1 += (in && E * (long long)G < t1) ? 1 : 0;
            E += in ? (long long)x[u] : 0ll;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        c0 += __shfl_xor(c0, o, 64);
        c1 += __shfl_xor(c1, o, 64);
    }
    if ((tid & 63) == 0) {
        atomicAdd(&sh.run[0], c0);
        atomicAdd(&sh.run[1], c1);
    }
    __syncthreads();
    if (tid == 0 && (int)blockIdx.x == G - 1) sh.run[1] = n;  // (the last run ends with the cloud)
    __syncthreads();
}

}  // namespace kicp
#include "kicp_icp_wide.hpp"
namespace kicp {

// One point's row of phase C: the sixteen products of J^T w J and J^T w r (Registration.cpp:80-121, Geman-McClure weight, strict
// d < max_dist), the correspondence count and the examined count, written 16 bytes at a time.  One expression for whoever forms the
// row (phase C's thread in both forms of the kernel; session y's experiment had the searching groups and an idle wave form
// rows too), so the bits do not depend on who does.
__device__ __forceinline__ void icp_terms_row(const double s[3], const double nn[3], double d2, int E, double max_dist, double ks, double *row_out) {
    double T[kIcpTerms];
#pragma unroll
    for (int k = 0; k < 17; ++k) T[k] = 0.0;
    T[17] = (double)E;
    // Registration.cpp:72: sqrt(d2) < max_dist, strict.  The root is taken only where it can decide: a squared distance 2^-40 below
    // max_dist^2 or above it is on its side of the threshold whatever the two roundings do (a root and its chain of ~25 dependent
    // instructions less on phase C's path, same verdicts).
    const double m2 = max_dist * max_dist;
    bool corr = d2 < m2 * (1.0 - 0x1p-40);
    if (__builtin_expect(!corr && d2 < m2 * (1.0 + 0x1p-40), 0)) corr = sqrt(d2) < max_dist;
    if (d2 < DBL_MAX && corr) {
        const double rx = s[0] - nn[0], ry = s[1] - nn[1], rz = s[2] - nn[2];
        const double r2 = (rx * rx + ry * ry) + rz * rz;
        const double w = (ks * ks) / ((ks + r2) * (ks + r2));
        T[0] = w;
        T[1] = w * s[0];
        T[2] = w * s[1];
        T[3] = w * s[2];
        // w * hat(s)^T hat(s) = w * (|s|^2 I - s s^T), upper triangle
        T[4] = w * (s[1] * s[1] + s[2] * s[2]);
        T[5] = w * (-(s[0] * s[1]));
        T[6] = w * (-(s[0] * s[2]));
        T[7] = w * (s[0] * s[0] + s[2] * s[2]);
        T[8] = w * (-(s[1] * s[2]));
        T[9] = w * (s[0] * s[0] + s[1] * s[1]);
        T[10] = w * rx;
        T[11] = w * ry;
        T[12] = w * rz;
        // w * (s x r)
        T[13] = w * (s[1] * rz - s[2] * ry);
        T[14] = w * (s[2] * rx - s[0] * rz);
        T[15] = w * (s[0] * ry - s[1] * rx);
        T[16] = 1.0;
    }
    double2 *row = reinterpret_cast<double2 *>(row_out);  // (144 bytes per row: 16-byte aligned)
#pragma unroll
    for (int k = 0; k < kIcpTerms / 2; ++k) row[k] = make_double2(T[2 * k], T[2 * k + 1]);
}

// One scalar's fixed-order sum over the (at most 16) rows of a reduction stage.  The stage's values lie TRANSPOSED in LDS -- row k =
// the sixteen contributions to scalar k, 128 contiguous bytes -- so that the thread of scalar k has all of them in flight at once
// (eight 16-byte loads, one round trip) and adds them from registers: v = ((0 + a0) + a1) + ... over the first `count`, the same
// additions in the same order as a loop over the contributors, only without a dependent LDS round trip per addend (8 - 16 of
// them per stage and three stages per iteration: ~2 us of every Gauss-Newton step, profiles/r06_n_*).  Entries beyond `count` are
// stale; their sums are formed and dropped.  The profiling build's tick slot is max-reduced instead.
template <bool PROF>
__device__ __forceinline__ double icp_row_sum(const double *row, int count, bool tick) {
    const double2 *r2 = reinterpret_cast<const double2 *>(row);
    double a[kIcpSumRows];
#pragma unroll
    for (int i = 0; i < kIcpSumRows / 2; ++i) {
        const double2 t = r2[i];
        a[2 * i] = t.x;
        a[2 * i + 1] = t.y;
    }
    double v = 0.0;
    // (the count is uniform and goes through an empty asm statement: left alone, the compiler forms the sixteen masks "j < count"
    // ahead of k_icp's iteration loop and keeps them in scalar registers it does not have)
    count = __builtin_amdgcn_readfirstlane(count);
    asm volatile("" : "+s"(count));
    if (count == kIcpSumRows) {
#pragma unroll
        for (int j = 0; j < kIcpSumRows; ++j) v = v + a[j];
    } else if (count == kIcpSumRows - 2) {  // (a leader's 14 members at 224 workgroups: the additions without their selects)
#pragma unroll
        for (int j = 0; j < kIcpSumRows - 2; ++j) v = v + a[j];
    } else {
#pragma unroll
        for (int j = 0; j < kIcpSumRows; ++j) {
            const double w = v + a[j];
            v = j < count ? w : v;
        }
    }
    if constexpr (PROF) {
        if (tick) {
            v = 0.0;
#pragma unroll
            for (int j = 0; j < kIcpSumRows; ++j) v = j < count ? fmax(v, a[j]) : v;
        }
    } else {
        v = tick ? 0.0 : v;  // (nobody fills that row in the release build)
    }
    return v;
}
static_assert(kIcpSumRows == 16 && kIcpGroupsPerBlock == 16 && kIcpExchangeGroups <= 16 && kIcpMaxMembers <= 16, "icp_row_sum's rows hold sixteen contributions");

// The solve of one Gauss-Newton step (Registration.cpp:156-157) on the sixteen sums: dx = LDLT(JTJ).solve(-JTr) -- through the
// 3 x 3 Schur complement when that is well conditioned (kicp_math.hpp), else the reference's pivoted LDLT --, est = SE3::exp(dx);
// nrm2 = |dx|^2 (all six components, :163).
__device__ __forceinline__ SE3 icp_solve_sums(const double *tot, int schur, double &nrm2) {
    double S[kIcpSums];
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) S[k] = tot[k];
    double dx[6];
    // well-conditioned systems (the rule) through their 3 x 3 Schur complement; anything else the reference's way
    if (!(schur && schur3_solve(S, dx))) {
        double JTJ[36], nb[6];
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
    }
    nrm2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) nrm2 += dx[i] * dx[i];
    return se3_exp(dx);
}
// ... out of line, result {q[4], t[3], |dx|^2} into LDS: the thread-per-query form's (one wave calls it)
__device__ __noinline__ void icp_solve_to_lds(const double *tot, double *est_out, int schur) {
    double nrm2;
    const SE3 est = icp_solve_sums(tot, schur, nrm2);
    if (threadIdx.x == 0) {
        est_out[0] = est.q[0];
        est_out[1] = est.q[1];
        est_out[2] = est.q[2];
        est_out[3] = est.q[3];
        est_out[4] = est.t[0];
        est_out[5] = est.t[1];
        est_out[6] = est.t[2];
        est_out[7] = nrm2;
    }
}

// WIDE: the association phases of kicp_icp_wide.hpp (a thread per source point) instead of the 32-lane groups below;
// everything else -- runs, the order of additions, exchange, solve -- is shared, so both forms give the same pose bit for bit.
template <bool PROF, bool WIDE>
__global__ __launch_bounds__(kIcpThreads) void k_icp(IcpParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (all LDS is carved from the dynamic region: a static __shared__ in front of it would
    // shift its base off 8/16-byte alignment)
    IcpShared &sh = *reinterpret_cast<IcpShared *>(smem);

    const int tid = threadIdx.x;  // (the iteration loop has its own, with lane and group: see there)
    const MapView &m = P.map;
    PipeState *st = P.state;

    // Everything the prologue needs from memory -- the error word, the cloud's size, the map's population, the state the initial
    // guess and the threshold come from -- is asked for HERE, before the first test: one round trip.  Read where it was used,
    // behind one early exit after the other, it was six dependent ones (~5 us of the first iteration: profiles/r06_q_*).
    const int err0 = __hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = count_of(P.n_ptr, P.n_imm);
    const int live0 = m.ctr[C_LIVE];
    const unsigned epoch_base = st->epoch_base;
    const int n_pre0 = P.prep ? P.prep->n_pre : 0, n_fd0 = P.prep ? P.prep->n_fd : 0;  // (for the frame's record, written at the very end)
    SE3 pose_a, pose_b;  // pipeline: last_pose, last_delta; else: the caller's guess (twice)
    double sse0 = 0.0;
    int samples0 = 1;
    if (P.pipeline_mode) {
        pose_a = st->last_pose;
        pose_b = st->last_delta;
        sse0 = st->model_sse;
        samples0 = st->num_samples;
    } else {
        pose_a = pose_b = st->guess;
    }
    // a frame whose registration timed out (workgroups not co-resident) poisons the frames queued behind
    // it: they leave the state untouched so that the host can replay from the failed frame
    if (err0 & E_TIMEOUT) {
        // (workgroup 0 advances the tag base on EVERY path it can leave by -- also when it only started after the others
        // had given up: the replay must never meet granules that carry this launch's tags)
        if (blockIdx.x == 0 && threadIdx.x == 0) st->epoch_base = epoch_base + (unsigned)P.max_iters + 2u;
        return;
    }
    if (P.inject_timeout) {  // test hook: what a launch that never became co-resident leaves behind
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicOr(&st->err, E_TIMEOUT);
            st->epoch_base = epoch_base + (unsigned)P.max_iters + 2u;
        }
        return;
    }
    const unsigned long long launch_cyc = clock64(), launch_tick = wall_clock64();
    // How many of the launched workgroups take part is decided here, from the actual N_src, so the
    // summation order (hence the result, bit for bit) never depends on host-side hints.
    int G = P.force_blocks > 0 ? P.force_blocks
                               : (n + kIcpGroupsPerBlock * P.points_per_group - 1) / (kIcpGroupsPerBlock * P.points_per_group);
    G = max(1, min(G, (int)gridDim.x));
    if ((int)blockIdx.x >= G) return;
    // The source cloud is taken in its spatial order (P.order: sorted {Morton code of the point's cell, index}
    // keys); workgroup b serves the contiguous run [b * n_run, (b + 1) * n_run) of it -- a compact patch of
    // the scene, so that the map voxels its points can reach fit in the workgroup's LDS tile.  The first
    // n_meta points of a run use the tile; any beyond that search HBM directly.
    SE3 guess;
    double max_dist, ks;
    if (P.pipeline_mode) {
        // KissICP.cpp:44-47: sigma = ComputeThreshold(); initial_guess = last_pose * last_delta
        const double sigma = sqrt(sse0 / (double)samples0);
        guess = se3_mul(pose_a, pose_b);
        max_dist = 3.0 * sigma;
        ks = sigma;
    } else {
        guess = pose_a;
        max_dist = P.max_dist;
        ks = P.kernel_scale;
    }
    const bool map_empty = (live0 == 0);  // Registration.cpp:143 (nothing to align to: no iteration, no exchange, no runs)
    int q0, n_local;
    if (P.wts32 && P.wts && P.order && n >= kIcpWeightedMin && P.force_blocks <= 0 && !map_empty && n <= kIcpListRunMax * G) {
        // (k_icp_weights has weighed the points: exactly the condition under which it does)
        const long room = (long)P.lds_bytes - (long)sizeof(IcpShared);
        unsigned *scratch = (long)n * 4 <= room ? reinterpret_cast<unsigned *>(smem + sizeof(IcpShared)) : nullptr;
        icp_runs_from_weights(P.wts32, &sh, scratch, n, G);
        q0 = sh.run[0];
        n_local = max(0, sh.run[1] - q0);
        __syncthreads();  // sh.part and the scratch are reused
    } else if (P.wts && P.order && n >= kIcpWeightedMin && P.force_blocks <= 0 && !map_empty) {
        IcpRunArgs R;
        R.frame = P.frame;
        R.order = P.order;
        R.wts = P.wts;
        R.granules = P.granules;
        R.slots = m.slots;
        R.mask = m.mask;
        R.voxel_size = m.voxel_size;
        R.state = st;
        R.weight_base = P.weight_base;
        R.weight_quad = P.weight_quad;
        R.weight_long_base = P.weight_long_base;
        R.weight_long_emul = P.weight_long_emul;
        R.dense_min = P.weight_dense_min;
        R.dense_div = P.weight_dense_div;
        R.spin_limit = P.spin_limit;
        R.dbg = m.dbg;
        if (!icp_weighted_run<WIDE>(R, &sh, guess, epoch_base, n, G)) {
            if (tid == 0) {
                atomicOr(&st->err, E_TIMEOUT);
                if (blockIdx.x == 0) st->epoch_base = epoch_base + (unsigned)P.max_iters + 2u;
            }
            return;
        }
        q0 = sh.run[0];
        n_local = max(0, sh.run[1] - q0);
        __syncthreads();  // sh.range_sum and sh.part are reused by the iterations
    } else {
        const int n_run = (n + G - 1) / G;
        q0 = (int)blockIdx.x * n_run;
        n_local = max(0, min(n_run, n - q0));
    }
    const int n_meta = (P.use_lds && m.max_points <= 32) ? min(n_local, WIDE ? kWideChunk : kIcpMaxMeta) : 0;
    const bool use_lists = !WIDE && n_meta > 0 && n_local <= kIcpListRunMax;
    // (group form) queries whose neighbour provably stays skip the search: runs that keep scan lists -- a single chunk, every
    // query with its record, its point slot its own through the launch
    const bool use_stable = use_lists && P.group_stable != 0 && n_local <= kIcpChunk && n_meta == n_local;
    // LDS behind the fixed part: only as many point slots of a chunk as the run can fill (a run of 16 points leaves
    // 9 KiB of the 128 to the tile), then the query records, the table, and the region of points and lists
    // (WIDE: all of sh.pts stays -- the slow-path queue and, with sh.terms, phase C's rows; 20-byte query records)
    const size_t head_bytes = WIDE ? sizeof(IcpShared) : offsetof(IcpShared, pts) + (size_t)min(kIcpChunk, max(n_local, 1)) * sizeof(IcpPoint);
    IcpQueryMeta *metas = reinterpret_cast<IcpQueryMeta *>(smem + head_bytes);
    WideMeta *wmetas = reinterpret_cast<WideMeta *>(smem + head_bytes);
    Tile tile;
    {
        char *q = smem + head_bytes + (WIDE ? (((size_t)n_meta * sizeof(WideMeta) + 15) & ~(size_t)15) : (size_t)n_meta * sizeof(IcpQueryMeta));
        // (WIDE with a few hundred queries: their windows hold 3 .. 5 k occupied voxels at the 1M-point configuration's steady
        // state -- a 4096-slot table was full there, misses walked 32 slots (58 us per search in the slowest waves) and
        // whole workgroups fell back to the map: profiles/r04_l_icp_probe_livox.txt -- so 8192 slots, at most 5/8 full)
        const int slots = WIDE ? (n_local > 192 ? 2 * kIcpTileSlots : kIcpTileSlots) : (n_local <= kIcpListRunMax ? kIcpTileSlots / 2 : kIcpTileSlots);
        tile.slots_mask = slots - 1;
        tile.hash_shift = slots == 2 * kIcpTileSlots ? 19 : (slots == kIcpTileSlots ? 20 : 21);
        tile.load_limit = WIDE ? (slots * P.wide_load_eighths) / 8 : (slots * 3) / 4;
        tile.keys = reinterpret_cast<unsigned *>(q);
        tile.vals = tile.keys + slots;
        q += (size_t)2 * slots * sizeof(unsigned);
        tile.points = reinterpret_cast<double *>(q);
        const long room = (long)P.lds_bytes - (long)(q - smem);
        tile.region_bytes = n_meta > 0 && room > 0 ? (unsigned)min(room, (long)0xFFFF * 24) & ~15u : 0u;
        tile.cap_points = (int)(tile.region_bytes / 24u);
        tile.lists = use_lists ? reinterpret_cast<unsigned short *>(q) : nullptr;
        tile.list_top = use_lists ? (int)(tile.region_bytes / 2u) : 0;
        tile.list_count = &sh.list_entries;
        tile.count = &sh.tile_points;
        tile.stored = &sh.tile_stored;
        tile.entries = &sh.tile_entries;
        tile.ox = tile.oy = tile.oz = 0;  // set once the first point's voxel is known
        tile.far = sh.far_point;
    }
    double(*terms)[kIcpTerms] = sh.terms;
    unsigned t_assoc = 0, t_publish = 0, t_gather = 0, t_solve = 0;
    unsigned gather_passes = 0;

    const double inv_voxel = 1.0 / m.voxel_size;

    if (tid == 0) {
        sh.fail = 0;
        sh.far_point[0] = sh.far_point[1] = sh.far_point[2] = 1e200;
        sh.tile_points = 0;
        sh.tile_stored = 0;
        sh.tile_entries = 0;
        sh.list_entries = 0;
        sh.any_fill = 0;
        sh.search_count = 0;
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
#pragma unroll
        for (int i = 0; i < 4; ++i) sh.book[i] = pose_a.q[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) sh.book[4 + i] = pose_a.t[i];
        sh.book[7] = 0.0;
    }
    if constexpr (WIDE) {
        for (int i = tid; i < n_meta; i += kIcpThreads) {
            wmetas[i].valid = 0;
            wmetas[i].list_state = 0;
        }
    } else {
        for (int i = tid; i < n_meta; i += kIcpThreads) {
            metas[i].valid = 0;
            metas[i].list_state = 0;
            metas[i].list_base = 0;
            metas[i].list_n = metas[i].list_cap = 0;
            metas[i].lr_state = 0;
        }
    }
    if (n_meta > 0)
        for (int i = tid; i <= tile.slots_mask; i += kIcpThreads) {
            tile.keys[i] = kTileEmpty;
            tile.vals[i] = 0u;
        }
    __syncthreads();

    SE3 est = guess;
    int iterations = 0, converged = 0;
    int range_err = 0;
    bool failed = false;
    WideQuery wq;  // (WIDE) this thread's query: kept in registers from iteration to iteration when the run is a single chunk
    wq.have_nn = false;
    wq.occ_valid = false;
    wq.lr_valid = false;
    wq.Lr = 0.0;
    wq.occ = 0u;
    wq.flag = 2;
    wq.E = 0;
    wq.d2 = DBL_MAX;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        wq.s[a] = wq.nn[a] = 0.0;
        wq.v[a] = 0;
    }

    const int max_iters = map_empty ? 0 : P.max_iters;
    for (int it = 0; it < max_iters; ++it) {
        const unsigned c0 = PROF ? ticks32() : 0u;
        // ---- association + accumulation, in three phases per chunk of kIcpChunk local points ---------------
        //   A  one THREAD per point: s = est * s (TransformPoints), its voxel, is its staged window still good
        //   B  one 32-lane GROUP per point: (re)stage the window if needed, closest neighbour (a2 / a3)
        //   C  one THREAD per point: correspondence test, weight, the 16 products of J^T w J, J^T w r (a5);
        //      then thread (g, k) adds term k of the points of group g in ascending local index -- the order
        //      in which a group's lane 0 used to accumulate them, so the sums are bit for bit what they were
        // A and C cost one instruction stream per WORKGROUP instead of one per point.
        // (The thread index goes through an empty asm statement at the top of every iteration and shadows the kernel's: whatever
        // is derived from it -- lane and group, the predicates "tid < 64", "lane >= o" of every prefix scan, LDS addresses, term and
        // group of phase C, scalar and member of the exchange -- is then recomputed where it is used.  Left alone, the compiler
        // hoists all of it out of the iteration loop: dozens of 64-bit predicate masks, which do not fit the scalar registers and
        // are parked lane by lane in vector registers (a dozen of them), and as many addresses -- in the thread-per-query form a
        // good part of the registers that did not fit.)
        const int tidv = kicp_tid();
        const int tid = tidv, lane = tidv & (kIcpGroup - 1), grp = tidv / kIcpGroup;
        // phase C: term and group of this thread.  kIcpPublishDpp: a term's 16 groups on the 16 lanes of one DPP row (thread 16 k + g),
        // so that the workgroup's sum of term k is four row operations away and goes to memory from the lane that holds it -- no
        // LDS round trip, no barrier and no chain of sixteen additions between phase C and the first hop (the exchange alone:
        // publish 0.36 -> 0.16 us, scripts/probes/xchg_bench.hip).  The groups' sums are then added as a tree, not one after the
        // other: the same tree in every workgroup, launch and form.
        const int ck = kIcpPublishDpp ? tidv / kIcpGroupsPerBlock : tidv % kIcpTerms, cg = kIcpPublishDpp ? tidv % kIcpGroupsPerBlock : tidv / kIcpTerms;
        const bool c_on = kIcpPublishDpp ? ck < kIcpTerms : cg < kIcpGroupsPerBlock;
        static_assert(kIcpGroupsPerBlock == 16 && kIcpTerms * kIcpGroupsPerBlock <= kIcpThreads, "a DPP row per term");
        double acc = 0.0;
        unsigned t_group = 0;
        unsigned prof_path = 0;
        if constexpr (WIDE) {
            // ---- the thread-per-query form (kicp_icp_wide.hpp) ---------------------------------------------------
            double(*rows)[kIcpTerms] = sh.terms;  // phase C: kWideTermRows rows, continued into sh.pts
            const bool single = n_local <= kWideChunk;  // the whole run is one chunk: s / nn stay in registers between iterations
            const double limit_corr = (max_dist * max_dist) * (1.0 + 0x1p-40);  // sqrt(d) < max_dist (Registration.cpp:72) implies d below this
            for (int base = 0; base < n_local; base += kWideChunk) {
                const int cn = min(kWideChunk, n_local - base);
                const bool active = tid < cn;
                const int j = base + tid;
                const bool has_meta = active && j < n_meta;  // (n_meta <= kWideChunk: queries of the first chunk)
                WideMeta *meta = wmetas + (has_meta ? j : 0);
                const unsigned ta = PROF ? ticks32() : 0u;
                double moved = 0.0;
                // ---- A: s = est * s, its voxel, is the known window still good ------------------------------------
                if (active) {
                    double pin[3];
                    if (it == 0) {
                        const int p = KICP_IDX(m.dbg, &st->err, P.order ? key_index(P.order[q0 + j]) : q0 + j, n, 3);
                        pin[0] = P.frame[3 * p];
                        pin[1] = P.frame[3 * p + 1];
                        pin[2] = P.frame[3 * p + 2];
                    } else if (single) {
                        pin[0] = wq.s[0];
                        pin[1] = wq.s[1];
                        pin[2] = wq.s[2];
                    } else {  // (a run of more than kWideChunk points: the running points live in HBM, by sorted position)
                        pin[0] = P.work[3 * (size_t)(q0 + j)];
                        pin[1] = P.work[3 * (size_t)(q0 + j) + 1];
                        pin[2] = P.work[3 * (size_t)(q0 + j) + 2];
                    }
                    // (the transform comes from LDS -- the initial guess, then every solve's result -- instead of staying in
                    // fourteen registers per thread through the whole iteration: this form's registers are all taken)
                    SE3 step;
                    {
                        const double *e = it == 0 ? sh.guess : sh.est;
                        step.q[0] = e[0];
                        step.q[1] = e[1];
                        step.q[2] = e[2];
                        step.q[3] = e[3];
                        step.t[0] = e[4];
                        step.t[1] = e[5];
                        step.t[2] = e[6];
                    }
                    se3_act(step, pin, wq.s);
                    if (single && it > 0) {  // how far the query has moved since the last iteration (for the stability test), rounded up
                        const double mx = wq.s[0] - pin[0], my = wq.s[1] - pin[1], mz = wq.s[2] - pin[2];
                        moved = sqrt((mx * mx + my * my) + mz * mz) * (1.0 + 0x1p-30) + DBL_MIN;
                    }
                    if (!single) {
                        P.work[3 * (size_t)(q0 + j)] = wq.s[0];
                        P.work[3 * (size_t)(q0 + j) + 1] = wq.s[1];
                        P.work[3 * (size_t)(q0 + j) + 2] = wq.s[2];
                        wq.have_nn = false;
                        wq.occ_valid = false;
                        wq.lr_valid = false;
                    }
                    const int vx = voxel_coord_fast(wq.s[0], m.voxel_size, inv_voxel), vy = voxel_coord_fast(wq.s[1], m.voxel_size, inv_voxel),
                              vz = voxel_coord_fast(wq.s[2], m.voxel_size, inv_voxel);
                    if (vx != wq.v[0] || vy != wq.v[1] || vz != wq.v[2]) {  // another voxel: another 27 cells -- nothing that was found out carries over
                        wq.occ_valid = false;
                        wq.lr_valid = false;
                        wq.have_nn = false;
                    }
                    wq.v[0] = vx;
                    wq.v[1] = vy;
                    wq.v[2] = vz;
                    // is the 27-neighbourhood of (vx, vy, vz) inside the known window?  (the record -- five words -- in one round
                    // trip and the test without short-circuits: ten dependent LDS round trips otherwise, like the group form's)
                    WideMeta Mw;
                    __builtin_memcpy(&Mw, __builtin_assume_aligned(meta, 4), sizeof Mw);
                    const int wrx = vx - Mw.v[0], wry = vy - Mw.v[1], wrz = vz - Mw.v[2];
                    const bool cached = has_meta & (Mw.valid > 0) & ((int)Mw.lo[0] <= wrx - 1) & (wrx + 1 <= (int)Mw.hi[0]) & ((int)Mw.lo[1] <= wry - 1) &
                                        (wry + 1 <= (int)Mw.hi[1]) & ((int)Mw.lo[2] <= wrz - 1) & (wrz + 1 <= (int)Mw.hi[2]);
                    wq.flag = cached ? 0 : ((has_meta & (Mw.valid >= 0)) ? 1 : 2);
                    if (wq.flag == 1) sh.any_fill = 1;
                    if (it == 0 && j == 0) {
                        // The tile's relative voxel coordinates: centred on the SENSOR when the map's reach fits the span (every
                        // query is a scan point: within max_distance of it) -- a far-field run of a few hundred sparse points
                        // may lie all around the sensor, farther apart than any span centred on one of them covers --, else on
                        // the run's first point.
                        const bool reach_fits = m.max_distance * inv_voxel < (double)(kTileSpanXY / 2 - 4);
                        const int cx = reach_fits ? voxel_coord(sh.guess[4], m.voxel_size) : vx, cy = reach_fits ? voxel_coord(sh.guess[5], m.voxel_size) : vy;
                        sh.origin[0] = cx - kTileSpanXY / 2;
                        sh.origin[1] = cy - kTileSpanXY / 2;
                        sh.origin[2] = vz - kTileSpanZ / 2;
                    }
                }
                if (tid == 0) {
                    sh.next_point = 0;    // queue of the window phase
                    sh.list_entries = 0;  // queue of the map-direct searches (this form keeps no scan lists)
                    sh.cell_count = 0;    // some query of the chunk needs the map-direct search
                    sh.job_count = 0;     // queue of the voxels in the LDS store that are left to the groups
                    sh.bulk_failed = 0;   // queue of the voxels in the map
                    if (PROF) sh.bulk_ticks[5] = sh.bulk_ticks[6] = 0u;  // (slowest table lookups of a search, searches that had to look up)
                }
                __syncthreads();
                tile.ox = sh.origin[0];
                tile.oy = sh.origin[1];
                tile.oz = sh.origin[2];
                // The slow paths: the queries that need one file themselves in a queue (sh.pts), the 32-lane groups serve it
                // with the routines of the first form, the owners read the verdicts back.  mode 1: establish the window
                // (tile_fill); mode 2: search the map directly (closest_neighbor_any).
                bool tie_redo = false;  // this iteration's search may have met a tie in NORM (kicp_search.hpp): the map-direct search settles it
                auto serve = [&](int mode, int *counter) -> int {
                    bool pending = active && (wq.flag == mode || (mode == 2 && tie_redo));
                    int served = 0;
                    // (every round serves up to kWideQueue of the at most kWideChunk pending queries: the trip count is bounded)
                    for (int round = 0; round < kWideChunk / kWideQueue + 2; ++round) {
                        int slot = -1;
                        if (pending) {
                            slot = atomicAdd(counter, 1);
                            if (slot < kWideQueue) {
                                IcpPoint &r = sh.pts[slot];
                                r.s[0] = wq.s[0];
                                r.s[1] = wq.s[1];
                                r.s[2] = wq.s[2];
                                r.v[0] = wq.v[0];
                                r.v[1] = wq.v[1];
                                r.v[2] = wq.v[2];
                                r.flag = mode;
                                r.pad = tid;
                            }
                        }
                        __syncthreads();
                        const int filed = min(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), kWideQueue);
                        if (filed == 0) break;  // (the whole workgroup)
                        served += filed;
                        for (int e = grp; e < filed; e += kIcpGroupsPerBlock) {
                            IcpPoint &r = sh.pts[e];
                            const double s[3] = {r.s[0], r.s[1], r.s[2]};
                            const int v[3] = {r.v[0], r.v[1], r.v[2]};
                            if (mode == 1) {
                                const bool ok = tile_fill(m, tile, s, v, lane, wmetas + base + r.pad, range_err);
                                if (lane == 0) r.flag = ok ? 0 : 2;
                            } else {
                                double nn[3];
                                int E = 0;
                                const double d2 = closest_neighbor_any(m, s[0], s[1], s[2], lane, nn, E, range_err);
                                if (lane == 0) {
                                    r.nn[0] = nn[0];
                                    r.nn[1] = nn[1];
                                    r.nn[2] = nn[2];
                                    r.d2 = d2;
                                    r.E = E;
                                }
                            }
                        }
                        __syncthreads();
                        if (pending && slot < kWideQueue) {
                            const IcpPoint &r = sh.pts[slot];
                            if (mode == 1) {
                                wq.flag = r.flag;
                            } else {
                                wq.nn[0] = r.nn[0];
                                wq.nn[1] = r.nn[1];
                                wq.nn[2] = r.nn[2];
                                wq.d2 = r.d2;
                                wq.E = r.E;
                                wq.have_nn = false;
                                wq.lr_valid = false;
                                wq.occ_valid = false;
                            }
                            pending = false;
                        }
                        if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __syncthreads();
                    }
                    return served;
                };
                // ---- B0: windows.  All of them in the first iteration (workgroup-wide), a few now and then later ----
                const unsigned tb00 = PROF ? ticks32() : 0u;
                if (sh.any_fill) {
                    bool bulk_done = false;
                    if (it == 0 && base == 0 && P.bulk_fill) {
                        int rerr = 0;
                        const bool mine = active && wq.flag == 1;
                        if (tid == 0) sh.bulk_ticks[7] = 0u;  // (set by a query the workgroup-wide phase leaves unsettled)
                        // 256 queries at a time (their windows have 16384 cell instances at most, a few thousand distinct cells:
                        // what the scratch beside an 8192-slot table holds); a batch with more distinct cells than the member
                        // list holds (sparse surroundings: nobody shares a cell) comes again in halves
                        bulk_done = true;
                        for (int lo = 0; lo < cn; lo += 256) {
                            const int hi = min(cn, lo + 256);
                            const bool part = mine && tid >= lo && tid < hi;
                            int r = wide_fill_bulk(m, tile, &sh, lo, hi, wmetas, part, wq.s, wq.v, &rerr, PROF && lo == 0, P.wide_prefill);
                            if (r == 1) {
                                if (part) wq.flag = meta->valid > 0 ? 0 : (meta->valid == 0 ? 1 : 2);
                            } else if (r == 2) {
                                for (int l2 = lo; l2 < hi; l2 += 128) {
                                    const int h2 = min(hi, l2 + 128);
                                    const bool part2 = mine && tid >= l2 && tid < h2;
                                    r = wide_fill_bulk(m, tile, &sh, l2, h2, wmetas, part2, wq.s, wq.v, &rerr, false, P.wide_prefill);
                                    if (r == 1) {
                                        if (part2) wq.flag = meta->valid > 0 ? 0 : (meta->valid == 0 ? 1 : 2);
                                    } else {
                                        bulk_done = false;  // (the rest one by one, below)
                                    }
                                }
                            } else {
                                bulk_done = false;
                            }
                        }
                        if (rerr) range_err = 1;
                        if (sh.bulk_ticks[7]) bulk_done = false;  // (a barrier has passed since it was written; some windows are left: one by one, below)
                    }
                    if (!bulk_done) serve(1, &sh.next_point);
                    if (tid == 0) sh.any_fill = 0;  // (barriers inside both routes: everybody has read it)
                }
                const unsigned t_fill = PROF ? ticks32() - tb00 : 0u;
                // ---- B: this thread's query against the tile ------------------------------------------------------
                const unsigned tb0 = PROF ? ticks32() : 0u;
                WideCounters ctr;
                ctr.visited_lds = ctr.visited_map = 0u;
                ctr.t_lookup = ctr.t_chains = ctr.t_walk = 0u;
                // (1) who needs a full search.  A query that has stayed in its voxel, whose last neighbour is still closer than
                // anything else can have come (WideQuery::Lr), needs nothing: its answer is nn again, from registers.
                bool need_full = active && wq.flag == 0;
                double limit0 = limit_corr;
                if (need_full) {
                    double dp = DBL_MAX;
                    if (wq.have_nn) {  // (found from this very voxel: the point is among the 27 cells)
                        const double ex = wq.nn[0] - wq.s[0], ey = wq.nn[1] - wq.s[1], ez = wq.nn[2] - wq.s[2];
                        dp = (ex * ex + ey * ey) + ez * ez;  // (as the search computes it)
                    }
                    if (P.wide_stable && single && it > 0 && wq.lr_valid && wq.occ_valid) {
                        wq.Lr -= moved;
                        bool ok;
                        if (wq.have_nn)
                            ok = sqrt(dp) * (1.0 + 0x1p-30) < wq.Lr;
                        else  // (no neighbour last time: nothing within the correspondence threshold -- or nothing at all)
                            ok = wq.occ == 0u || sqrt(limit_corr) * (1.0 + 0x1p-30) < wq.Lr;
                        if (ok) {
                            wq.d2 = dp;  // (E stays)
                            need_full = false;
                        }
                    }
                    // what can still matter in a full search: the correspondence threshold and the last neighbour
                    if (need_full && P.wide_prune > 1 && wq.have_nn) limit0 = dp < limit0 ? dp : limit0;
                    if (need_full) wq.lr_valid = false;
                }
                // (2) the full searches run on the first lanes when there are few of them (a wave's search costs the same with
                // one lane busy as with 64), in place when most queries need one (the first iterations)
                int my_rank = -1;
                if (need_full) my_rank = atomicAdd(&sh.next_point, 1);
                __syncthreads();
                const int n_full = sh.next_point;  // (the whole workgroup)
                // (few of them: a 32-lane group per query, no queues -- wide_group_scan; its records live in the queues' memory)
                const bool by_groups = P.wide_stable && n_full > 0 && n_full <= min(P.wide_group_max, kWideGroupRecs);
                const bool compact = !by_groups && P.wide_stable && n_full > 0 && n_full <= kWideRecs;
                WideRec *recs = reinterpret_cast<WideRec *>(sh.part);
                static_assert(kWideFlatScratchBytes <= sizeof(double) * (kIcpGroupsPerBlock + kIcpSumRows) * kIcpSums, "scratch of the flat service");
                if (compact) {
                    if (need_full) {
                        WideRec &r = recs[my_rank];
                        r.s[0] = wq.s[0];
                        r.s[1] = wq.s[1];
                        r.s[2] = wq.s[2];
                        r.limit = limit0;
                        r.v[0] = wq.v[0];
                        r.v[1] = wq.v[1];
                        r.v[2] = wq.v[2];
                        r.occ = wq.occ;
                        r.E = wq.E;
                        r.cached = wq.occ_valid ? 1 : 0;
                    }
                    __syncthreads();
                }
                // (compacted: record r goes to lane r % 16 of wave r / 16 -- the first four waves sit on different SIMDs, and a
                // search's data-dependent loops end with the slowest of 16 lanes instead of 64)
                const int my_rec = (tid >> 6) * 16 + (tid & 63);
                const bool rec_worker = compact && (tid & 63) < 16 && my_rec < n_full;
                bool searching = compact ? rec_worker : need_full;  // this lane runs a full search, for the query `job`
                WideJob job;
                job.s[0] = wq.s[0];
                job.s[1] = wq.s[1];
                job.s[2] = wq.s[2];
                job.v[0] = wq.v[0];
                job.v[1] = wq.v[1];
                job.v[2] = wq.v[2];
                job.occ = wq.occ;
                job.E = wq.E;
                job.cached = wq.occ_valid;
                job.d2 = DBL_MAX;
                job.Lr = 0.0;
                job.bkey = 0x7FFFFFFF;
                if (rec_worker) {
                    const WideRec &r = recs[my_rec];
                    job.s[0] = r.s[0];
                    job.s[1] = r.s[1];
                    job.s[2] = r.s[2];
                    job.v[0] = r.v[0];
                    job.v[1] = r.v[1];
                    job.v[2] = r.v[2];
                    limit0 = r.limit;
                    job.occ = r.occ;
                    job.E = r.E;
                    job.cached = r.cached != 0;
                }
                WideBest wb;
                wb.best = DBL_MAX;
                wb.bx = wb.by = wb.bz = 0.0;
                wb.bkey = 0x7FFFFFFF;
                wb.limit = limit0;
                wb.m_map = wb.m_lds = 0u;
                wb.seen = 0u;
                wb.sec = DBL_MAX;
                int job_bad = 0;
                bool got = false;  // this thread's query has had its full search
                if (by_groups) {
                    WideRec *grecs = reinterpret_cast<WideRec *>(sh.terms);
                    if (need_full) {
                        WideRec &r = grecs[my_rank];
                        r.s[0] = wq.s[0];
                        r.s[1] = wq.s[1];
                        r.s[2] = wq.s[2];
                        r.v[0] = wq.v[0];
                        r.v[1] = wq.v[1];
                        r.v[2] = wq.v[2];
                        r.limit = P.wide_prune > 0 ? limit0 : DBL_MAX;
                    }
                    __syncthreads();
                    for (int r0 = grp; r0 < n_full; r0 += kIcpGroupsPerBlock) {
                        WideRec &r = grecs[r0];
                        double gnn[3], gsec;
                        int gE, gbad;
                        unsigned gocc;
                        const double gs[3] = {r.s[0], r.s[1], r.s[2]};
                        const int gv[3] = {r.v[0], r.v[1], r.v[2]};
                        const double gd = wide_group_scan(m, tile, gs, gv, r.limit, lane, gnn, gE, gocc, gsec, gbad);
                        if (lane == 0) {
                            r.s[0] = gnn[0];
                            r.s[1] = gnn[1];
                            r.s[2] = gnn[2];
                            r.limit = gd;
                            r.v[2] = gbad;
                            r.occ = gocc;
                            r.E = gE;
                            r.Lr = sqrt(gsec) * (1.0 - 0x1p-30);
                        }
                    }
                    __syncthreads();
                    if (need_full) {
                        const WideRec &r = grecs[my_rank];
                        job_bad = r.v[2];
                        wb.bx = r.s[0];
                        wb.by = r.s[1];
                        wb.bz = r.s[2];
                        job.d2 = r.limit;
                        job.occ = r.occ;
                        job.E = r.E;
                        job.Lr = r.Lr;
                        got = true;
                    }
                    searching = false;
                    __syncthreads();  // (the records' memory is the queues')
                }
                if (searching) {
                    const bool looked_up = !job.cached;
                    wide_search_lds<PROF>(m, tile, job, limit0, P.wide_prune > 0, job_bad, ctr, wb);
                    if (PROF) {
                        if (looked_up) atomicAdd(&sh.bulk_ticks[6], 1u);
                        atomicMax(&sh.bulk_ticks[5], ctr.t_lookup + ctr.t_chains);
                    }
                    if (job_bad) searching = false;  // (the tile cannot answer: a voxel outside the key span, an entry that did not fit)
                }
                // Whatever this query still has to look at -- voxels in the LDS store beyond the first, voxels in the map -- goes
                // into the two queues the groups serve; what does not fit waits for the next round.
                WideItem *items = reinterpret_cast<WideItem *>(sh.terms);
                unsigned pend_lds = searching ? wb.m_lds : 0u, pend_map = searching ? wb.m_map : 0u;
                if (active && wq.flag == 2) sh.cell_count = 1;  // (queries without a tile; those a search has just found unanswerable join below)
                const unsigned t_scan = PROF ? ticks32() - tb0 : 0u;
                int prof_items = 0, prof_map_items = 0, prof_rounds = 0, prof_direct = 0;
                unsigned prof_file = 0, prof_serve = 0, prof_merge = 0, prof_c = 0;
                // items a thread files per round and queue: the nearest first, the rest is held against what they bring back -- all at
                // once when only the few compacted searches are filing (everything fits one round)
                const int kPerRound = compact ? 27 : P.wide_per_round;
                auto file_items = [&](unsigned &pend, int *counter, int cap, WideItem *dst, int &base, int &n_filed) {
                    const int n_want = min(__popc(pend), kPerRound);
                    base = 0;
                    n_filed = 0;
                    if (n_want == 0) return;
                    base = atomicAdd(counter, n_want);
                    n_filed = max(0, min(n_want, cap - base));
                    for (int r = 0; r < n_filed; ++r) {
                        const int jj = __ffs(pend) - 1;
                        pend &= pend - 1u;
                        WideItem &it = dst[base + r];
                        it.s[0] = job.s[0];
                        it.s[1] = job.s[1];
                        it.s[2] = job.s[2];
                        it.d2 = DBL_MAX;
                        unsigned slot = 0u;
                        it.blk_cnt = wide_entry(tile, job.v[0], job.v[1], job.v[2], jj, &slot) & ~(kTileReady | kTileGlobal);
                        it.slot = (unsigned short)slot;
                        it.j = (unsigned char)jj;
                        it.k = 0;
                    }
                    if (PROF) ctr.visited_map += (unsigned)n_filed;  // (items filed, of either kind)
                };
                auto merge_items = [&](const WideItem *src, int base, int n_filed) {
                    for (int r = 0; r < n_filed; ++r) {
                        const WideItem &it = src[base + r];
                        wide_take(wb, job.s[0], job.s[1], job.s[2], it.s[0], it.s[1], it.s[2], ((int)it.j << 5) | (int)it.k, it.d2 < DBL_MAX);
                        const double runner_up = (double)__uint_as_float(it.blk_cnt);  // (FLT_MAX: the voxel holds one point)
                        if (it.d2 < DBL_MAX && runner_up < (double)FLT_MAX && runner_up < wb.sec) wb.sec = runner_up;
                        wb.seen |= 1u << it.j;
                    }
                };
                for (int round = 0;; ++round) {
                    if (round >= kWideRoundLimit) {  // (every round files at least one item while any is pending: far beyond any real count)
                        if (tid == 0) sh.fail = 1;   // ... the launch gives up like a failed exchange instead of spinning
                        break;
                    }
                    const unsigned tr0 = PROF ? ticks32() : 0u;
                    // (while the store is empty -- the first iteration, as a rule -- the whole queue is the map's.  The store only
                    // changes between barriers of this loop and in the window phases: every thread reads the same value here.)
                    const int cap_l = __hip_atomic_load(tile.stored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 ? 0 : kWideItemsLds, cap_m = kWideItems - cap_l;
                    if (round > 0 && pend_map && it >= P.wide_promote_from) {  // a voxel may have been promoted into the store since this query classified it
                        unsigned todo = pend_map;
                        while (todo) {
                            const int jj = __ffs(todo) - 1;
                            todo &= todo - 1u;
                            if (!(wide_entry(tile, job.v[0], job.v[1], job.v[2], jj) & kTileGlobal)) {
                                pend_map &= ~(1u << jj);
                                pend_lds |= 1u << jj;
                            }
                        }
                    }
                    int base_l, nf_l, base_m, nf_m;
                    file_items(pend_lds, &sh.job_count, cap_l, items, base_l, nf_l);
                    file_items(pend_map, &sh.bulk_failed, cap_m, items + cap_l, base_m, nf_m);
                    __syncthreads();
                    const unsigned tr1 = PROF ? ticks32() : 0u;
                    if (PROF) prof_file += tr1 - tr0;
                    const int n_l = min(sh.job_count, cap_l), n_m = min(sh.bulk_failed, cap_m);  // (the whole workgroup)
                    if (n_l + n_m == 0) break;
                    if (PROF) {
                        prof_items += n_l + n_m;
                        prof_map_items += n_m;
                        ++prof_rounds;
                    }
                    // (the runner-up also in the first iteration: with its margin a good half of the queries need no search in the second)
                    // (icp_wide_flat: a thread per point instead of a group per item; the compacted searches' records are in the
                    // owners' registers by now and go back after the last round: their memory is the flat service's scratch)
                    if (n_m) {
                        if (P.wide_flat & 1)
                            wide_serve_flat<false>(m, tile, items + cap_l, n_m, recs, it >= P.wide_promote_from);
                        else
                            wide_serve_items<false>(m, tile, items + cap_l, n_m, grp, lane, it >= P.wide_promote_from, true);
                    }
                    if (n_l) {
                        if (P.wide_flat & 2)
                            wide_serve_flat<true>(m, tile, items, n_l, recs, false);
                        else
                            wide_serve_items<true>(m, tile, items, n_l, grp, lane, false, true);
                    }
                    __syncthreads();
                    const unsigned tr2 = PROF ? ticks32() : 0u;
                    merge_items(items, base_l, nf_l);
                    merge_items(items + cap_l, base_m, nf_m);
                    if ((pend_lds | pend_map) != 0u && P.wide_prune > 0) {  // what is left, against what the answers have brought
                        wb.limit = wb.best < wb.limit ? wb.best : wb.limit;
                        const unsigned keep = wide_keep_mask(wide_gaps(job.s, job.v, m.voxel_size), wb.limit);
                        pend_lds &= keep;
                        pend_map &= keep;
                    }
                    if (tid == 0) sh.job_count = sh.bulk_failed = 0;
                    __syncthreads();  // (the queues' memory is the next round's, the slow paths' and phase C's)
                    if (PROF) {
                        prof_serve += tr2 - tr1;
                        prof_merge += ticks32() - tr2;
                    }
                }
                // (3) the answers go home
                if (searching) wide_finish(m, job, wb);
                if (compact) {
                    if (rec_worker) {
                        WideRec &r = recs[my_rec];
                        r.s[0] = wb.bx;
                        r.s[1] = wb.by;
                        r.s[2] = wb.bz;
                        r.limit = job.d2;
                        r.v[0] = job.bkey;
                        r.v[2] = job_bad;
                        r.occ = job.occ;
                        r.E = job.E;
                        r.Lr = job.Lr;
                    }
                    __syncthreads();
                    if (need_full) {
                        const WideRec &r = recs[my_rank];
                        job_bad = r.v[2];
                        wb.bx = r.s[0];
                        wb.by = r.s[1];
                        wb.bz = r.s[2];
                        job.d2 = r.limit;
                        job.occ = r.occ;
                        job.E = r.E;
                        job.Lr = r.Lr;
                        got = true;
                    }
                } else if (!by_groups) {
                    got = need_full;
                }
                if (got) {
                    if (job_bad) {  // the map from now on
                        meta->valid = -1;
                        wq.flag = 2;
                        wq.have_nn = false;
                        wq.occ_valid = wq.lr_valid = false;
                        sh.cell_count = 1;
                    } else {
                        wq.d2 = job.d2;
                        wq.E = job.E;
                        wq.occ = job.occ;
                        wq.occ_valid = true;
                        wq.Lr = job.Lr;
                        wq.lr_valid = true;
                        wq.have_nn = job.d2 < DBL_MAX;
                        if (wq.have_nn) {
                            wq.nn[0] = wb.bx;
                            wq.nn[1] = wb.by;
                            wq.nn[2] = wb.bz;
                        }
                        // The reference compares NORMS (kicp_search.hpp): if anything else can be as close as the neighbour to
                        // within a few units in the last place -- Lr bounds every other point from below, shaved by 2^-30 -- the
                        // squared distances may have picked another candidate than the rounded roots would: the map-direct
                        // search, which settles such ties the reference's way, answers this query in this iteration.
                        if (wq.have_nn && job.Lr <= sqrt(job.d2) * (1.0 + 0x1p-28)) {
                            tie_redo = true;
                            sh.cell_count = 1;
                        }
                    }
                }
                if (compact || n_full > 0) __syncthreads();  // (sh.cell_count; the records' memory is the exchange's)
                if (sh.cell_count) prof_direct = serve(2, &sh.list_entries);
                if (PROF) t_group += ticks32() - tb0;
                // ---- C: products of kWideTermRows points at a time, added in the first form's order ----------------
                const unsigned tc0 = PROF ? ticks32() : 0u;
                for (int sub = 0; sub < cn; sub += kWideTermRows) {
                    const int sn = min(kWideTermRows, cn - sub);
                    if (tid >= sub && tid < sub + sn) {
                        const double s[3] = {wq.s[0], wq.s[1], wq.s[2]};
                        const double d2 = wq.d2;
                        double *T = rows[tid - sub];
#pragma unroll
                        for (int k = 0; k < 17; ++k) T[k] = 0.0;
                        T[17] = (double)wq.E;
                        if (d2 < DBL_MAX && sqrt(d2) < max_dist) {  // Registration.cpp:72 (strict)
                            const double rx = s[0] - wq.nn[0], ry = s[1] - wq.nn[1], rz = s[2] - wq.nn[2];
                            const double r2 = (rx * rx + ry * ry) + rz * rz;
                            const double w = (ks * ks) / ((ks + r2) * (ks + r2));
                            T[0] = w;
                            T[1] = w * s[0];
                            T[2] = w * s[1];
                            T[3] = w * s[2];
                            T[4] = w * (s[1] * s[1] + s[2] * s[2]);
                            T[5] = w * (-(s[0] * s[1]));
                            T[6] = w * (-(s[0] * s[2]));
                            T[7] = w * (s[0] * s[0] + s[2] * s[2]);
                            T[8] = w * (-(s[1] * s[2]));
                            T[9] = w * (s[0] * s[0] + s[1] * s[1]);
                            T[10] = w * rx;
                            T[11] = w * ry;
                            T[12] = w * rz;
                            T[13] = w * (s[1] * rz - s[2] * ry);
                            T[14] = w * (s[2] * rx - s[0] * rz);
                            T[15] = w * (s[0] * ry - s[1] * rx);
                            T[16] = 1.0;
                        }
                    }
                    __syncthreads();
                    if (c_on) {
                        // term ck of the points cg, cg + 16, ... in that order, four asked for together (eight: this form's registers
                        // are all taken, the kernel spills)
                        constexpr int kPer = 4;
                        static_assert((kWideTermRows / kIcpGroupsPerBlock) % kPer == 0, "phase C's batches");
                        for (int i0 = cg; i0 < sn; i0 += kPer * kIcpGroupsPerBlock) {
                            double a[kPer];
#pragma unroll
                            for (int u = 0; u < kPer; ++u) {
                                const int i = i0 + u * kIcpGroupsPerBlock;
                                a[u] = rows[i < sn ? i : cg][ck];
                            }
#pragma unroll
                            for (int u = 0; u < kPer; ++u) {
                                const double w = acc + a[u];
                                acc = i0 + u * kIcpGroupsPerBlock < sn ? w : acc;
                            }
                        }
                    }
                    __syncthreads();
                }
                if (PROF) prof_c = ticks32() - tc0;
                if (PROF && P.prof_groups && lane == 0 && it < kIcpProfIters && base == 0) {
                    // this thread's record of the iteration (10 ns ticks); same layout as the first form's group records
                    unsigned *r = P.prof_groups + ((size_t)it * (kIcpMaxBlocks * kIcpGroupsPerBlock) + (size_t)blockIdx.x * kIcpGroupsPerBlock + grp) * 4;
                    r[0] = (unsigned)(tb00 - ta) | (min(ctr.visited_lds, 255u) << 16) | (min(ctr.visited_map, 255u) << 24);  // phase A; voxels visited (LDS, map)
                    if (it == 0 && P.bulk_fill && grp < 5) r[0] = (r[0] & 0xFFFFu) | (min(sh.bulk_ticks[grp], 0xFFFFu) << 16);  // groups 0..4: the window phase's parts instead
                    r[1] = (unsigned)min(t_fill, 0xFFFFu) | ((unsigned)min(t_scan, 0xFFFFu) << 16);
                    r[2] = (unsigned)min(sh.tile_points, 0xFFFF) | ((unsigned)min(wq.E, 0xFFFF) << 16);
                    if (grp & 1) {  // the odd groups (same wave as the even one in front): where the search's time went instead
                        r[0] = (r[0] & 0xFFFFu) | (min(ctr.t_walk, 0xFFFFu) << 16);
                        r[2] = min(ctr.t_lookup, 0xFFFFu) | (min(ctr.t_chains, 0xFFFFu) << 16);
                    }
                    if (grp == 2) r[2] = (unsigned)min(prof_direct, 0xFFFF) | ((unsigned)min(prof_rounds, 0xFFFF) << 16);  // workgroup: map-direct queries, queue rounds
                    if (grp == 6) r[2] = min(prof_file, 0xFFFFu) | (min(prof_serve, 0xFFFFu) << 16);  // workgroup: filing (+ wait), serving
                    if (grp == 8) r[2] = min(prof_merge, 0xFFFFu) | (min(prof_c, 0xFFFFu) << 16);     // workgroup: merging, phase C
                    if (grp == 4) r[2] = (unsigned)min(prof_items, 0xFFFF) | ((unsigned)min(prof_map_items, 0xFFFF) << 16);  // workgroup: items served, of them in the map
                    if (grp == 12) r[2] = (unsigned)min(n_full, 0xFFFF) | (min(sh.bulk_ticks[6], 0xFFFFu) << 16);  // workgroup: full searches, of them with table lookups
                    if (grp == 14) r[2] = min(sh.bulk_ticks[5], 0xFFFFu) | (compact ? 0x10000u : 0u);  // workgroup: the slowest search's lookups + chains; searches compacted
                    r[3] = 4u;
                    prof_path = 4u;
                }
            }
        } else
        for (int base = 0; base < n_local; base += kIcpChunk) {
            const int cn = min(kIcpChunk, n_local - base);
            // ---- A -------------------------------------------------------------------------------------
            // (One wave's instruction stream on every workgroup's critical path: everything it needs from LDS -- the query's record,
            // what its last search left in the point slot -- is asked for at the top, in one round trip, and the tests are formed
            // without short-circuits; read field by field under its conditions the phase was ~25 dependent LDS round trips,
            // ~1 us of its 1.5: profiles/r06_n_*.  Values that are read before anything has written them -- the first
            // iteration's -- only ever reach selects that drop them.)
            if (tid < cn) {
                const int j = base + tid;
                const bool has_meta = j < n_meta;
                IcpQueryMeta *meta = metas + (has_meta ? j : 0);
                IcpPoint &pt = sh.pts[tid];
                IcpQueryMeta M;
                __builtin_memcpy(&M, __builtin_assume_aligned(meta, 16), sizeof M);
                const double last_d2 = pt.d2, last_nn0 = pt.nn[0], last_nn1 = pt.nn[1], last_nn2 = pt.nn[2];
                // (the point's index in the cloud: only where the cloud or the work array is touched -- a global load on the path
                // of every iteration otherwise)
                int p = 0;
                double pin[3];
                if (it > 0 && has_meta) {  // running source point lives in LDS
                    pin[0] = M.s[0];
                    pin[1] = M.s[1];
                    pin[2] = M.s[2];
                } else {
                    p = KICP_IDX(m.dbg, &st->err, P.order ? key_index(P.order[q0 + j]) : q0 + j, n, 4);
                    const double *src = (it == 0) ? P.frame : P.work;
                    pin[0] = src[3 * p];
                    pin[1] = src[3 * p + 1];
                    pin[2] = src[3 * p + 2];
                }
                double s[3];
                se3_act(est, pin, s);
                const int vx = voxel_coord_fast(s[0], m.voxel_size, inv_voxel), vy = voxel_coord_fast(s[1], m.voxel_size, inv_voxel),
                          vz = voxel_coord_fast(s[2], m.voxel_size, inv_voxel);
                // is the 27-neighbourhood of (vx, vy, vz) inside the known window?
                const int rx = vx - M.v[0], ry = vy - M.v[1], rz = vz - M.v[2];
                const bool cached = has_meta & (M.valid > 0) & ((int)M.lo[0] <= rx - 1) & (rx + 1 <= (int)M.hi[0]) & ((int)M.lo[1] <= ry - 1) &
                                    (ry + 1 <= (int)M.hi[1]) & ((int)M.lo[2] <= rz - 1) & (rz + 1 <= (int)M.hi[2]);
                // STABILITY (IcpQueryMeta::L2).  The query's last search, made from this very voxel, left its neighbour and count in
                // this point slot (a run with lists is a single chunk: the slot is this query's through the launch), the position it
                // was made from, and the second smallest squared distance over the 27 cells.  While the neighbour's new distance
                // plus the way from there to here stays strictly below that runner-up's distance, the neighbour is still the unique
                // minimum the reference's strict '<' loops would find (VoxelHashMap.cpp:55-63), with the same points examined: no
                // search, and the distance is computed here -- by the expression the search uses, so the bits are the search's.
                const bool may_stay = use_stable & (it > 0) & cached & (M.lr_state == 1) & (M.lv[0] == vx) & (M.lv[1] == vy) & (M.lv[2] == vz);
                const bool has_nn = last_d2 < DBL_MAX;  // (DBL_MAX: the 27 cells hold no candidate at all -- and never will)
                const double mx = s[0] - M.ss[0], my = s[1] - M.ss[1], mz = s[2] - M.ss[2];
                const double b2 = (mx * mx + my * my) + mz * mz;
                const double ex = last_nn0 - s[0], ey = last_nn1 - s[1], ez = last_nn2 - s[2];
                const double dp = (ex * ex + ey * ey) + ez * ez;  // (as the search computes it)
                const double R = (M.L2 - dp) - b2;
                const bool holds = (R > 0.0) & ((4.0 * (1.0 + 0x1p-20)) * (dp * b2) < R * R);
                const bool stable = may_stay & (holds | !has_nn);
                if (stable & has_nn) pt.d2 = dp;
                if (has_meta) {
                    meta->s[0] = s[0];
                    meta->s[1] = s[1];
                    meta->s[2] = s[2];
                    if (!stable) meta->lr_state = 0;  // (renewed by the search that follows, if it is a list scan)
                } else {
                    P.work[3 * p] = s[0];
                    P.work[3 * p + 1] = s[1];
                    P.work[3 * p + 2] = s[2];
                }
                pt.s[0] = s[0];
                pt.s[1] = s[1];
                pt.s[2] = s[2];
                pt.v[0] = vx;
                pt.v[1] = vy;
                pt.v[2] = vz;
                const int flag = stable ? 3 : (cached ? 0 : ((has_meta & (M.valid >= 0)) ? 1 : 2));
                pt.flag = flag;
                if (flag == 1) sh.any_fill = 1;
                // the points that need a search file themselves: a chunk of one wave (the rule) by its ballot, without the round trip
                // of an atomic per point
                if (cn <= 64) {
                    const unsigned long long need = __ballot(!stable);
                    if (!stable) sh.search_idx[__popcll(need & ((1ull << tid) - 1ull))] = (unsigned char)tid;
                    if (tid == 0) sh.search_count = __popcll(need);
                } else if (!stable) {
                    sh.search_idx[atomicAdd(&sh.search_count, 1)] = (unsigned char)tid;
                }
                if (it == 0 && j == 0) {  // the tile's relative voxel coordinates are centred on the run's first point
                    sh.origin[0] = vx - kTileSpanXY / 2;
                    sh.origin[1] = vy - kTileSpanXY / 2;
                    sh.origin[2] = vz - kTileSpanZ / 2;
                }
            }
            if (tid == 0) sh.next_point = kIcpGroupsPerBlock;  // the first 16 points that need a search go to the groups directly
            __syncthreads();
            if (it == 0) {  // (set by the first iteration's phase A and never again: the group form has the registers to keep it)
                tile.ox = sh.origin[0];
                tile.oy = sh.origin[1];
                tile.oz = sh.origin[2];
            }
            // ---- B0: queries that are outside their known window (all of them in the first iteration, a few
            // now and then later) establish a new one.  Kept apart from the searches: a voxel one group is
            // still fetching may be the neighbour another group is about to look for.
            const unsigned tb00 = PROF ? ticks32() : 0u;
            bool bulk_done = false;
            if (sh.any_fill && it == 0 && P.bulk_fill) {
                int rerr = 0;
                bulk_done = tile_fill_bulk(m, tile, &sh, cn, metas + base, &rerr, PROF);
                if (rerr) range_err = 1;
            }
            if (bulk_done) {
                if (tid == 0) sh.any_fill = 0;
            } else if (sh.any_fill) {
                for (int t = grp; t < cn; t += kIcpGroupsPerBlock) {
                    IcpPoint &pt = sh.pts[t];
                    if (pt.flag != 1) continue;
                    const double s[3] = {pt.s[0], pt.s[1], pt.s[2]};
                    const int v[3] = {pt.v[0], pt.v[1], pt.v[2]};
                    const bool ok = tile_fill(m, tile, s, v, lane, metas + base + t, range_err);
                    if (lane == 0) pt.flag = ok ? 0 : 2;
                }
                __syncthreads();
                if (tid == 0) sh.any_fill = 0;
            }
            const unsigned t_fill = PROF ? ticks32() - tb00 : 0u;
            // ---- B -------------------------------------------------------------------------------------
            // Neighbourhoods differ by an order of magnitude in size (15 .. 540 points examined), so the groups
            // do not take a fixed share: each takes the next unserved point when it is done.  Which group
            // serves a point has no influence on the result (phase C adds in point order).
            const unsigned tb0 = PROF ? ticks32() : 0u;
            const int n_search = sh.search_count;  // (filed before the barrier behind phase A; reset behind the one that ends this phase)
            if (PROF && P.prof_groups && lane == 0 && it < kIcpProfIters && base == 0) {
                // every group's record of this iteration starts as "no search" (path 5; the workgroup's searches in the examined
                // field): a group that serves a point overwrites it below
                unsigned *r = P.prof_groups + ((size_t)it * (kIcpMaxBlocks * kIcpGroupsPerBlock) + (size_t)blockIdx.x * kIcpGroupsPerBlock + grp) * 4;
                r[0] = (unsigned)(tb0 - c0);
                r[1] = (unsigned)min(t_fill, 0xFFFFu);
                r[2] = (unsigned)min(sh.tile_points, 0xFFFF) | ((unsigned)min(n_search, 0xFFFF) << 16);
                prof_path = 5u;
            }
            // (The first eight searches go to eight different WAVES -- group 2 w takes search w, group 2 w + 1 search 8 + w --: the two
            // groups of a wave run in lock step, a scan as long as the longer list, a list build of one half with the other
            // masked off, and a later iteration has a handful of searches per workgroup.  kIcpSpreadSearches, profiles/r06_t_*.)
            const int e_first = kIcpSpreadSearches ? (grp >> 1) + (kIcpGroupsPerBlock / 2) * (grp & 1) : grp;
            // (a group's first search is asked for beside the count, not behind the test on it: one LDS round trip less in front of
            // every workgroup's first search)
            const int t_first = (int)sh.search_idx[e_first];
            for (int e = e_first; e < n_search;) {
                const int t = e == e_first ? t_first : (int)sh.search_idx[e];
                IcpPoint &pt = sh.pts[t];
                IcpQueryMeta *meta = metas + ((base + t < n_meta) ? base + t : 0);
                // (the point slot and the query's record in ONE round trip, as register copies: what decides about the list -- its
                // state, the voxel it belongs to, where it lies -- is not read field by field behind the conditions)
                IcpPoint ptc;
                __builtin_memcpy(&ptc, __builtin_assume_aligned(&pt, 16), sizeof ptc);
                IcpQueryMeta Mc;
                __builtin_memcpy(&Mc, __builtin_assume_aligned(meta, 16), sizeof Mc);
                const double s[3] = {ptc.s[0], ptc.s[1], ptc.s[2]};
                const int vx = ptc.v[0], vy = ptc.v[1], vz = ptc.v[2];
                int flag = ptc.flag;
                int list_state = Mc.list_state, list_base = Mc.list_base, list_n = Mc.list_n;
                int path = flag == 0 ? 0 : 3;  // profiling: 0 tile (lane per voxel), 1 tile (scan list), 3 HBM search
                const unsigned tb = PROF ? ticks32() : 0u;
                const unsigned tc = tb;
                double nn[3];
                double d2 = DBL_MAX;
                int E = 0;
                bool listed = false;
                bool tie = false;  // the fast search's answer may not be the reference's: a tie in NORM (kicp_search.hpp) -- settled below
                unsigned t_build = 0u;  // (profiling: the list build's share of the search, later iterations)
                if (flag == 0 && !listed && use_lists && list_state >= 0) {
                    // the scan list belongs to the voxel the query was in when it was built
                    // (Round 6 tried NOT rebuilding the list of a query that has entered another voxel -- with the stability shortcut it
                    // is searched once there, as a rule -- and sending it through the lane-per-voxel search instead: that search
                    // is 5 us where build + list scan are 3.4, on the critical path of the iteration: profiles/r06_d_ab_*.txt.)
                    if (list_state == 0 || Mc.lv[0] != vx || Mc.lv[1] != vy || Mc.lv[2] != vz) {
                        const unsigned t0 = PROF ? ticks32() : 0u;
                        // (where the list lies and how long it is come back in registers: the scan does not wait for the record)
                        list_state = tile_list_build(tile, vx, vy, vz, lane, meta, (int)Mc.list_base, (int)Mc.list_cap, list_base, list_n) ? 1 : -1;
                        if (PROF) t_build = ticks32() - t0;
                    }
                    if (list_state == 1) {
                        E = list_n;
                        double sec2 = DBL_MAX;
                        d2 = tile_scan_list(tile, tile.lists + list_base, E, s[0], s[1], s[2], lane, nn, &tie, use_stable, &sec2);
                        listed = true;
                        path = 1;
                        // (a tie in norm is settled by the exact search below: its neighbour has no margin worth keeping)
                        if (use_stable && lane == 0 && !tie) {
                            meta->L2 = sec2 * (1.0 - 0x1p-20);
                            meta->ss[0] = s[0];
                            meta->ss[1] = s[1];
                            meta->ss[2] = s[2];
                            meta->lr_state = 1;
                        }
                    }
                }
                if (flag == 0 && !listed) {
                    int bad;
                    double sec2 = DBL_MAX;
                    d2 = tile_scan(m, tile, s[0], s[1], s[2], vx, vy, vz, lane, nn, E, bad, &tie, use_stable, &sec2);
                    if (bad) {  // 2: a voxel of this query did not fit into the tile -> HBM from now on; 1: one is
                                // being fetched by another group this very moment -> HBM this once
                        if (bad == 2 && lane == 0) meta->valid = -1;
                        flag = 2;
                        path = 3;
                    } else if (use_stable && lane == 0 && !tie) {
                        // (a query without a scan list -- the pool was full, a voxel of it lives in the map -- walks all 27 cells
                        // here: its runner-up bounds everything but the neighbour just as a list's does.  lv: the voxel the bound
                        // belongs to; a query that cannot have a list does not use it otherwise, one that can has it set already.)
                        meta->L2 = sec2 * (1.0 - 0x1p-20);
                        meta->ss[0] = s[0];
                        meta->ss[1] = s[1];
                        meta->ss[2] = s[2];
                        meta->lv[0] = vx;
                        meta->lv[1] = vy;
                        meta->lv[2] = vz;
                        meta->lr_state = 1;
                    }
                }
                if (flag != 0) {
                    d2 = closest_neighbor_any(m, s[0], s[1], s[2], lane, nn, E, range_err);  // (settles ties in norm itself)
                } else if (__builtin_expect(tie, 0)) {
                    int e2;  // (the tile's count stands: the same 27 cells)
                    d2 = closest_neighbor_exact(m, s[0], s[1], s[2], lane, nn, e2, range_err);
                }
                if (lane == 0) {
                    pt.nn[0] = nn[0];
                    pt.nn[1] = nn[1];
                    pt.nn[2] = nn[2];
                    pt.d2 = d2;
                    pt.E = E;
                }
                const unsigned td = PROF ? ticks32() : 0u;
                if (PROF && P.prof_groups && lane == 0 && it < kIcpProfIters && base == 0 && e == e_first) {
                    // per-group record of this iteration (10 ns ticks): where the group's time went
                    unsigned *r = P.prof_groups + ((size_t)it * (kIcpMaxBlocks * kIcpGroupsPerBlock) +
                                                   (size_t)blockIdx.x * kIcpGroupsPerBlock + grp) * 4;
                    r[0] = (unsigned)(tb0 - c0) | ((it > 0 ? t_build : (unsigned)(tb - tb0)) << 16);  // phase A + barrier; wait inside B (first iteration) / list build (later ones)
                    if (it == 0 && P.bulk_fill && grp < 5) r[0] = (r[0] & 0xFFFFu) | (min(sh.bulk_ticks[grp], 0xFFFFu) << 16);  // groups 0..4: tile_fill_bulk's phases instead
                    r[1] = (unsigned)min(t_fill, 0xFFFFu) | ((unsigned)(td - tc) << 16);  // window phase of the chunk, search
                    r[2] = (unsigned)min(sh.tile_points, 0xFFFF) | ((unsigned)E << 16);  // points in the tile so far, examined
                    r[3] = (unsigned)path;
                    prof_path = (unsigned)path;
                }
                if (n_search <= kIcpGroupsPerBlock) break;  // (every search had its group: nothing to come back for)
                int nt = 0;
                if (lane == 0) nt = atomicAdd(&sh.next_point, 1);
                e = __shfl(nt, 0, 32);
            }
            if (PROF) t_group += ticks32() - tb0;
            __syncthreads();
            if (tid == 0) sh.search_count = 0;  // (the next chunk's phase A files behind phase C's barriers)
            // ---- C -------------------------------------------------------------------------------------
            for (int sub = 0; sub < cn; sub += kIcpTermChunk) {
                const int sn = min(kIcpTermChunk, cn - sub);
                // (Forming the rows during phase B instead -- the last wave for the stable points while the others search, a searching
                // group for its own point, this stage and its barrier gone -- measured 0.5 - 1 % SLOWER on both bench commands:
                // profiles/r06_y_*, the patch beside the numbers.)
                if (tid < sn) {
                    // (the point slot in one round trip, the row of products written once, from registers)
                    IcpPoint pt;
                    __builtin_memcpy(&pt, __builtin_assume_aligned(&sh.pts[sub + tid], 16), sizeof pt);
                    icp_terms_row(pt.s, pt.nn, pt.d2, pt.E, max_dist, ks, terms[tid]);
                }
                __syncthreads();
                if (c_on) {
                    // term ck of the points cg, cg + 16, ... in that order: all (at most four) asked for together
                    double a[kIcpTermChunk / kIcpGroupsPerBlock];
#pragma unroll
                    for (int u = 0; u < kIcpTermChunk / kIcpGroupsPerBlock; ++u) {
                        const int i = cg + u * kIcpGroupsPerBlock;
                        a[u] = terms[i < sn ? i : cg][ck];
                    }
#pragma unroll
                    for (int u = 0; u < kIcpTermChunk / kIcpGroupsPerBlock; ++u) {
                        const double w = acc + a[u];
                        acc = cg + u * kIcpGroupsPerBlock < sn ? w : acc;
                    }
                }
                if (sub + kIcpTermChunk < cn) __syncthreads();  // (the rows are the next sub-chunk's; behind the last one the reduction's barrier follows)
            }
        }
        if (PROF && P.prof_groups && lane == 0 && it < kIcpProfIters) {
            // the group's record of this iteration, completed: its first point's path, the points of the workgroup's
            // run, the time the group spent searching in all chunks
            unsigned *r = P.prof_groups + ((size_t)it * (kIcpMaxBlocks * kIcpGroupsPerBlock) + (size_t)blockIdx.x * kIcpGroupsPerBlock + grp) * 4;
            r[3] = (prof_path & 15u) | ((unsigned)min(n_local, 4095) << 4) | (min(t_group, 0xFFFFu) << 16);
        }
        // ---- workgroup reduction (fixed order) ----------------------------------------------
        const unsigned c1 = PROF ? ticks32() : 0u;
        // (the stages' values lie transposed -- [scalar][contributor] --: icp_row_sum)
        double *const part_t = &sh.part[0][0], *const sums_t = &sh.range_sum[0][0];
        if (!kIcpPublishDpp && c_on) part_t[ck * kIcpSumRows + cg] = acc;
        if (PROF && lane == 0) part_t[kIcpTickSlot * kIcpSumRows + grp] = (double)t_group;  // this group's search time (profiling, max-reduced)
        if (!kIcpPublishDpp || PROF) __syncthreads();
        const unsigned epoch = epoch_base + (unsigned)it + 1u;
        unsigned long long *gran = P.granules + (size_t)(it & 1) * G * (2 * kIcpGranStride);
        const __amdgpu_buffer_rsrc_t gran_rsrc = granule_rsrc(gran, (unsigned)(G * 2 * kIcpGranStride * sizeof(unsigned long long)));
        // (the release build leaves the profiling slot -- the last one -- out of the exchange: one pair less per block, ~0.06 us
        // of the exchange alone, scripts/probes/xchg_bench.hip)
        constexpr int KX = PROF ? kIcpSums : kIcpSums - 1;
        static_assert(kIcpTickSlot == kIcpSums - 1, "the profiling slot is the last");
        if (kIcpPublishDpp) {
            if (c_on) {
                const double v = row16_sum(acc);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                if (cg == 0) granule_store_pair(gran_rsrc, (unsigned)(((size_t)blockIdx.x * kIcpGranStride + ck) * 16), epoch, (unsigned)bits, (unsigned)(bits >> 32));
            }
            if (PROF && tid == kIcpTickSlot) {
                const double v = icp_row_sum<PROF>(part_t + tid * kIcpSumRows, kIcpGroupsPerBlock, true);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                granule_store_pair(gran_rsrc, (unsigned)(((size_t)blockIdx.x * kIcpGranStride + tid) * 16), epoch, (unsigned)bits, (unsigned)(bits >> 32));
            }
        } else if (tid < KX) {
            const int k = tid;
            const double v = icp_row_sum<PROF>(part_t + k * kIcpSumRows, kIcpGroupsPerBlock, k == kIcpTickSlot);
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            granule_store_pair(gran_rsrc, (unsigned)(((size_t)blockIdx.x * kIcpGranStride + k) * 16), epoch, (unsigned)bits, (unsigned)(bits >> 32));
        }
        // ---- exchange: leaders sum their groups, every workgroup sums the groups and solves ----------------------
        // (One root gathering all G partials -- round 2 -- had 224 x 19 granule pairs queue up in ONE CU's memory
        // path, ~4.3 us, and a second hop of ~1.5 us to hand the update back; every workgroup gathering every
        // partial -- round 1 -- moved 17.5 MB per iteration.)
        const unsigned c2 = PROF ? ticks32() : 0u;
        const int ng = min(kIcpExchangeGroups, G);  // groups = leaders; group g holds the workgroups g, g + ng, g + 2 ng, ...
        // (the group sums lie there kIcpGroupCopies times, a workgroup reads copy b mod 8 -- its XCD's, as workgroups are placed --: a
        // line is polled by 28 workgroups instead of 224, the second hop 1.08 -> 0.92 us in the exchange alone: profiles/r06_ae_*)
        unsigned long long *grp_gran = P.granules + (size_t)2 * kIcpMaxBlocks * (2 * kIcpGranStride) +
                                       (size_t)(it & 1) * kIcpGroupCopies * kIcpExchangeGroups * (2 * kIcpGranStride);
        const __amdgpu_buffer_rsrc_t grp_rsrc =
            granule_rsrc(grp_gran, (unsigned)(kIcpGroupCopies * kIcpExchangeGroups * 2 * kIcpGranStride * sizeof(unsigned long long)));
        const int my_copy = (int)blockIdx.x % kIcpGroupCopies;
        // poll one granule pair until both halves carry this iteration's tag; false: gave up (bounded spin, or another
        // workgroup has already raised the timeout)
        auto poll_pair = [&](const __amdgpu_buffer_rsrc_t &r, unsigned off, double &out) -> bool {
            unsigned long long lo, hi;
            granule_load_pair(r, off, lo, hi);
            unsigned spins = 0;
            while ((unsigned)(lo >> 32) != epoch || (unsigned)(hi >> 32) != epoch) {
                if (PROF) ++gather_passes;
                if (++spins > P.spin_limit ||
                    ((spins & 255u) == 0 && (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & E_TIMEOUT)))
                    return false;
                __builtin_amdgcn_s_sleep(2);
                granule_load_pair(r, off, lo, hi);
            }
            out = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
            return true;
        };
        if ((int)blockIdx.x < ng) {
            // leader: thread (k, j) fetches scalar k of the group's j-th member; kParts (28) members per pass, all of a pass in flight
            const int members = (G - (int)blockIdx.x + ng - 1) / ng;  // <= kIcpMaxMembers
            constexpr int kParts = kIcpThreads / KX;
            if (tid < kParts * KX) {
                const int k = tidv % KX;
                for (int j = tidv / KX; j < members; j += kParts) {
                    const int b = (int)blockIdx.x + ng * j;
                    double v = 0.0;
                    if (!poll_pair(gran_rsrc, (unsigned)(((size_t)b * kIcpGranStride + k) * 16), v)) sh.fail = 1;
                    sums_t[k * kIcpSumRows + j] = v;
                }
            }
            __syncthreads();
            if (tid < KX) {
                // (the failure flag is read beside the row, not in front of it: one LDS round trip on the path from the first hop to the second)
                const double v = icp_row_sum<PROF>(sums_t + tid * kIcpSumRows, members, tid == kIcpTickSlot);  // member order: fixed by G alone
                const int leader_failed = sh.fail;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                if (!leader_failed) {
#pragma unroll
                    for (int c = 0; c < kIcpGroupCopies; ++c)
                        granule_store_pair(grp_rsrc, (unsigned)((((size_t)c * kIcpExchangeGroups + blockIdx.x) * kIcpGranStride + tid) * 16), epoch, (unsigned)bits,
                                           (unsigned)(bits >> 32));
                }
            }
            __syncthreads();  // range_sum is reused below
        }
        // Round 3's form (kIcpPollAll = false) had FEW pollers: ONE lane per group watched the group's first granule pair -- its 19
        // pairs leave the leader in one store instruction --, then a sweep fetched the rest: 224 workgroups x 152 lanes re-reading
        // the group sums while the slowest workgroups still searched had slowed exactly those (their overflow voxels came from
        // L2 / HBM then: profiles/r03_g).  Since the stability shortcut the later iterations' few searches run out of LDS, and the
        // watchers cost a memory round trip and a barrier per iteration: every lane of the sweep polls its own pair from the
        // start -- a later iteration 9.9 -> 9.5 us (steady map), 9.0 -> 8.5 (young), +3 % on both bench commands, same box
        // (profiles/r06_o_*).
        if (!kIcpPollAll) {
            if (tid < ng && !sh.fail) {
                double dummy;
                if (!poll_pair(grp_rsrc, (unsigned)((((size_t)my_copy * kIcpExchangeGroups + tid) * kIcpGranStride) * 16), dummy)) sh.fail = 1;
            }
            __syncthreads();
        }
        for (int e = tidv; e < ng * KX; e += kIcpThreads) {  // (a poll that follows a failure gives up by the error word: poll_pair)
            const int k = e % KX, g = e / KX;
            double v = 0.0;
            if (!poll_pair(grp_rsrc, (unsigned)((((size_t)my_copy * kIcpExchangeGroups + g) * kIcpGranStride + k) * 16), v)) sh.fail = 1;
            sums_t[k * kIcpSumRows + g] = v;
        }
        __syncthreads();
        const int exchange_failed = sh.fail;  // (asked for in front of the rows below, looked at behind them: one round trip, not two)
        if (tid < kIcpSums) sh.tot[tid] = icp_row_sum<PROF>(sums_t + tid * kIcpSumRows, ng, tid == kIcpTickSlot);
        if (exchange_failed) {
            // tell the waiting workgroups at once (they check the error word while they spin)
            if (tid == 0) atomicOr(&st->err, E_TIMEOUT);
            failed = true;
            break;
        }
        __syncthreads();
        // ---- waves 0..3 (one per SIMD) of EVERY workgroup solve the same system; the result goes through LDS -----
        const unsigned c3 = PROF ? ticks32() : 0u;
        double nrm2 = 0.0;
        if constexpr (WIDE) {
            // (thread-per-query form: ONE wave solves, out of line, and the update is read from LDS where it is used -- phase A, the
            // bookkeeping thread.  Inlined on four waves like below, the solve's ~150 registers on top of every thread's query
            // state were where this form's spills came from: 193 registers, 380 bytes of scratch per lane, 335 MB of writes per
            // launch in round 5.)
            if (tid < 64) icp_solve_to_lds(sh.tot, sh.est, P.schur_solve);
            __syncthreads();
            nrm2 = sh.est[7];
        } else {
            if (tid < kIcpSolveThreads) {
                est = icp_solve_sums(sh.tot, P.schur_solve, nrm2);
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
        }
        if (blockIdx.x == 0 && tid == kIcpBookThread) {
            // T_icp = est * T_icp (Registration.cpp:161) and the statistics, by a thread whose wave is
            // not on the critical path; sh.tot stays valid until the next gather
            SE3 T, step = est;
            if constexpr (WIDE) {  // (this form keeps the update in LDS only)
#pragma unroll
                for (int i = 0; i < 4; ++i) step.q[i] = sh.est[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) step.t[i] = sh.est[4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) T.q[i] = sh.T_icp[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) T.t[i] = sh.T_icp[4 + i];
            T = se3_mul(step, T);
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
        // ||dx|| < convergence_criterion (Registration.cpp:163), the root taken only where it can decide (see icp_terms_row)
        // (the group form: the thread-per-query form has no register to spare for the two bounds)
        bool done;
        if constexpr (WIDE) {
            done = sqrt(nrm2) < P.conv;
        } else {
            const double conv2 = P.conv * P.conv;
            done = nrm2 < conv2 * (1.0 - 0x1p-40);
            if (__builtin_expect(!done && nrm2 < conv2 * (1.0 + 0x1p-40), 0)) done = sqrt(nrm2) < P.conv;
        }
        if (done) {
            converged = 1;
            break;
        }
    }

    if (range_err) atomicOr(&st->err, E_RANGE);
    if (failed && tid == 0) atomicOr(&st->err, E_TIMEOUT);

    __syncthreads();  // the bookkeeping thread's last update is in LDS
    if (failed) {  // nothing is committed except the tag base of the exchange
        if (blockIdx.x == 0 && tid == 0) st->epoch_base = epoch_base + (unsigned)P.max_iters + 2u;
        return;
    }
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
            st->n_pre = n_pre0;
            st->n_fd = n_fd0;
        }
        st->icp_blocks_used = G;
#pragma unroll
        for (int k = 0; k < 18; ++k) st->icp_last_sums[k] = iterations > 0 ? sh.tot[k] : 0.0;
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
            if (model_error > P.min_motion_th) {  // (the sums as the prologue read them: nobody else writes them)
                st->model_sse = sse0 + model_error * model_error;
                st->num_samples = samples0 + 1;
            }
            SE3 last_pose;
#pragma unroll
            for (int i = 0; i < 4; ++i) last_pose.q[i] = sh.book[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) last_pose.t[i] = sh.book[4 + i];
            st->last_delta = se3_mul(se3_inverse(last_pose), new_pose);
            st->last_pose = new_pose;
        }
    }
}

size_t icp_granule_words(int G) {  // the workgroups' partials, then the leaders' group sums; both twice (iteration parity)
    return (size_t)2 * G * 2 * kIcpGranStride + (size_t)2 * kIcpGroupCopies * kIcpExchangeGroups * 2 * kIcpGranStride;
}

int icp_prepare(int device_id) {
    // opt in to the full 160 KiB of LDS (dynamic regions above 64 KiB need the attribute), once per
    // device: the attribute belongs to the device's copy of the code object
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device_id < 0 || device_id >= 64) return (int)hipErrorInvalidDevice;
    if (done[device_id]) return 0;
    const void *kernels[4] = {reinterpret_cast<const void *>(k_icp<false, false>), reinterpret_cast<const void *>(k_icp<true, false>),
                              reinterpret_cast<const void *>(k_icp<false, true>), reinterpret_cast<const void *>(k_icp<true, true>)};
    for (const void *k : kernels) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kIcpLdsBytesMax);
        if (e != hipSuccess) return (int)e;
    }
    done[device_id] = true;
    return 0;
}
// device self-test of the 6x6 solve (kicp_selftest_solve): the kernel's ldlt6_solve on caller-supplied systems
__global__ __launch_bounds__(64) void k_selftest_solve(const double *A, const double *b, int n, double *x) {
    for (int c = threadIdx.x; c < n; c += 64) {
        double M[36], rhs[6], xs[6];
#pragma unroll
        for (int e = 0; e < 36; ++e) M[e] = A[36 * c + e];
#pragma unroll
        for (int e = 0; e < 6; ++e) rhs[e] = b[6 * c + e];
        ldlt6_solve(M, rhs, xs);
#pragma unroll
        for (int e = 0; e < 6; ++e) x[6 * c + e] = xs[e];
    }
}
void launch_selftest_solve(const double *A, const double *b, int n, double *x, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_solve, dim3(1), dim3(64), 0, s, A, b, n, x);
}

int icp_blocks_per_cu(int lds_bytes) {
    int a[4] = {0, 0, 0, 0};
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a[0], k_icp<false, false>, kIcpThreads, (size_t)lds_bytes) != hipSuccess) a[0] = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a[1], k_icp<true, false>, kIcpThreads, (size_t)lds_bytes) != hipSuccess) a[1] = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a[2], k_icp<false, true>, kIcpThreads, (size_t)lds_bytes) != hipSuccess) a[2] = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a[3], k_icp<true, true>, kIcpThreads, (size_t)lds_bytes) != hipSuccess) a[3] = 0;
    int r = a[0];
    for (int i = 1; i < 4; ++i) r = a[i] < r ? a[i] : r;
    return r;
}
// wide: the thread-per-query form of the association (kicp_icp_wide.hpp) -- same result, for clouds of many points per workgroup
template <bool PROF, bool WIDE>
static void launch_icp_as(const IcpParams &P, int G, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    if (start || stop)
        hipExtLaunchKernelGGL((k_icp<PROF, WIDE>), dim3(G), dim3(kIcpThreads), (size_t)P.lds_bytes, s, start, stop, 0, P);
    else
        hipLaunchKernelGGL((k_icp<PROF, WIDE>), dim3(G), dim3(kIcpThreads), (size_t)P.lds_bytes, s, P);
}
void launch_icp(IcpParams P, int G, bool profile, bool wide, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    if (wide) {
        if (profile)
            launch_icp_as<true, true>(P, G, s, start, stop);
        else
            launch_icp_as<false, true>(P, G, s, start, stop);
    } else {
        if (profile)
            launch_icp_as<true, false>(P, G, s, start, stop);
        else
            launch_icp_as<false, false>(P, G, s, start, stop);
    }
}

}  // namespace kicp
