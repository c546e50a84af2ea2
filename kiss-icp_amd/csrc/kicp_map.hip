// kicp_map.hip -- the voxel hash map in HBM:
//   k_closest_neighbor     VoxelHashMap::GetClosestNeighbor, batched     core/VoxelHashMap.cpp:46-70
//   k_map_link/apply       VoxelHashMap::AddPoints                       core/VoxelHashMap.cpp:97-119
//   k_map_prune            VoxelHashMap::RemovePointsFarFromLocation     core/VoxelHashMap.cpp:121-132
//   (fused update: the verdicts of RemovePointsFarFromLocation are taken beside k_map_link, carried out by k_map_apply)
//   k_map_rehash, k_map_count_points                                     (table upkeep, Pointcloud sizing)
#include "kicp_search.hpp"

namespace kicp {

static inline int grid_for(long n, int threads, int cap) {
    long g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
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
// VoxelHashMap::RemovePointsFarFromLocation (VoxelHashMap.cpp:121-132), the parts every form of it shares: a voxel dies
// iff its FIRST point is >= max_distance from the origin.
// ------------------------------------------------------------------------------------------
// the reference's test on a point given in registers (VoxelHashMap.cpp:126-128)
__device__ __forceinline__ bool prune_point_dies(double x, double y, double z, double ox, double oy, double oz, double md2) {
    const double dx = x - ox, dy = y - oy;
    const double a = dx * dx + dy * dy;
    const double dz = z - oz;
    return a + dz * dz >= md2;
}
// ... and on a live block.  The header and xy[0] share the block's first cache line, z[0] lies in another.  The key says which
// layer of voxels z[0] is in: floor(fl(z / v)) == vz puts z into [vz v, (vz + 1) v] up to a few units in the last place (the
// slack below is 2^-48 of the magnitudes, as in kicp_icp_wide.hpp: wide_gaps), subtraction and the sum below are monotone
// under rounding, so dz = fl(z - oz) lies in [lo, hi] and the test's left-hand side between the two sums formed from the
// smallest and the largest |dz| -- when both fall on the same side of max_distance^2, as they do for all but a thin shell of
// voxels, the verdict is the reference's without z ever being read: half the traffic (the prune of a 2.4 M-voxel map read
// 283 MB, 0.2 - 0.3 ms, profiles/r04_final2_*).
__device__ __forceinline__ bool prune_block_dies(const MapView &m, const BlockHdr *hdr, int b, double ox, double oy, double oz, double md2) {
    const double2 p0 = block_xy(m, b)[0];
    const double dx = p0.x - ox, dy = p0.y - oy;
    const double a = dx * dx + dy * dy;
    int vx, vy, vz;
    unpack_voxel(hdr->key, vx, vy, vz);
    const double f0 = (double)vz * m.voxel_size, f1 = (double)(vz + 1) * m.voxel_size;
    const double slack = (fabs(f0) + fabs(f1)) * 0x1p-48 + DBL_MIN;
    const double lo = (f0 - slack) - oz, hi = (f1 + slack) - oz;
    const double d_far = fmax(fabs(lo), fabs(hi)), d_near = (lo <= 0.0 && hi >= 0.0) ? 0.0 : fmin(fabs(lo), fabs(hi));
    if (a + d_far * d_far < md2) return false;
    if (a + d_near * d_near >= md2) return true;
    const double dz = block_z(m, b)[0] - oz;
    return a + dz * dz >= md2;  // VoxelHashMap.cpp:126-128
}
// tombstone the slot, recycle the block (the caller owns the voxel: nobody else writes its slot or header)
__device__ __forceinline__ void voxel_remove(const MapView &m, BlockHdr *hdr, int b) {
    Slot *sl = m.slots + KICP_IDX(m.dbg, m.ctr + C_ERR, hdr->slot, (long long)m.mask + 1, 30);
    sl->key = kKeyTomb;
    sl->block = -1;
    sl->count = 0;
    hdr->count = 0;
    hdr->doom = 2;
    const unsigned k = (unsigned)atomicAdd(&m.ctr[C_FPEND], 1);
    m.free_ids[k % (unsigned)m.free_cap] = b;
    atomicSub(&m.ctr[C_LIVE], 1);
    atomicAdd(&m.ctr[C_TOMB], 1);
}
// the last workgroup of a frame's last kernel copies the map counters and the PipeState behind them -- rec_words 32-bit
// words, contiguous in HBM -- straight into the frame's slot of the host-pinned ring: no blit kernel.  `expected`
// workgroups sign off (all of them call this, or none).
// What the record holds are words that were written by device-scope atomics (performed where all XCDs see them) or by
// earlier kernels; a workgroup's atomics have been performed when its barrier lets thread 0 through (the barrier waits for
// every wave's outstanding memory operations), so NO device-scope release fence stands in front of the sign-off: on this
// device that fence writes the XCD's L2 back, and 500 workgroups of k_map_apply signing off behind one each took 17 us of a
// 35 us kernel (profiles/r06_g_timeline_*.txt).  And the sign-off is counted on two levels -- workgroup b on word b mod 16,
// whoever completes a word on C_DONE: 500 returning atomics on ONE word are 6 us (12 ns each, one after the other).
__device__ __forceinline__ void frame_record_handoff(const MapView &m, unsigned *host_rec, int rec_words, int expected) {
    __shared__ int sh_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 0;
        if (expected <= 2 * kDoneSub) {
            last = atomicAdd(&m.ctr[C_DONE], 1) == expected - 1;
        } else {
            const int j = (int)blockIdx.x % kDoneSub;
            const int mine = (expected - j + kDoneSub - 1) / kDoneSub;  // workgroups b < expected with b mod kDoneSub == j
            int *w = m.done_sub + j * kCtrStride;
            if (atomicAdd(w, 1) == mine - 1) {
                __hip_atomic_store(w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (idle again: nobody signs here before the next frame)
                last = atomicAdd(&m.ctr[C_DONE], 1) == kDoneSub - 1;
            }
        }
        sh_last = last;
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
        if (threadIdx.x == 0) __hip_atomic_store(&m.ctr[C_DONE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------
// VoxelHashMap::AddPoints (VoxelHashMap.cpp:97-119), made deterministic on the device:
//   k_map_link   every new point finds or claims its voxel's slot (CAS on the packed key), opens or
//                joins the voxel's record of this insert and files its index there (plus a chain
//                for voxels that receive more than kRecList points);
//   k_map_apply  one 32-lane group per record applies the reference's sequential acceptance rule
//                (voxel full? closer than map_resolution to a stored point? else append) to the
//                record's points in ascending point index -- the same result as the serial loop.
// The FUSED update (the pipeline's frame; kicp_map_update_*; option "map_fused_update") folds RemovePointsFarFromLocation into
// the two kernels -- one launch and ~10 us less on the serial chain between two registrations (KissICP.cpp:61):
//   beside k_map_link, `scan_blocks` extra workgroups take the verdicts of all voxels that exist (read-only: the first point of
//   an existing voxel does not change by appending to it; VoxelHashMap.cpp:126-128) and list the sentenced blocks
//   (BlockHdr::doom = 1, MapView::doomed); removing them there would be wrong: a point of this frame that falls into such a
//   voxel must find it (the reference appends, then removes the whole voxel -- a tombstone in front of k_map_link's probe would
//   make it create the voxel anew);
//   k_map_apply then (a) carries the list out, (b) does not append to a sentenced voxel but removes it -- whoever changes
//   doom 1 -> 2 first, the list's executor or the record's group, owns the removal, the other keeps its hands off --, and
//   (c) judges the voxels it CREATES itself by their first point, which it holds in registers.
// Same map as link -> apply -> prune, voxel for voxel and point for point (tests/test_gpu_paths.py: test_fused_map_update_*).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_link(MapView m, InsertScratch sc, const double *in, const int *n_ptr,
                                                  int n_imm, const PipeState *state, int use_pose, int scan_blocks,
                                                  int use_state_origin, double ox, double oy, double oz) {
    const int link_blocks = (int)gridDim.x - scan_blocks;
    if ((int)blockIdx.x >= link_blocks) {
        // ---- fused update: the verdict pass of RemovePointsFarFromLocation (read-only but for the sentences) ----------
        const int bump0 = m.ctr[C_BUMP];
        const int verr0 = state ? __hip_atomic_load(&state->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        if (use_state_origin) {
            ox = state->new_pose.t[0];
            oy = state->new_pose.t[1];
            oz = state->new_pose.t[2];
        }
        const double md2 = m.max_distance * m.max_distance;
        int nb = min(bump0, m.blocks_cap);
        if (verr0 & E_TIMEOUT) nb = 0;  // no pose
        int *n_doomed = &m.ctr[C_DOOMED0 + sc.parity * kCtrStride];
        for (int b = ((int)blockIdx.x - link_blocks) * (int)blockDim.x + (int)threadIdx.x; b < nb; b += scan_blocks * (int)blockDim.x) {
            BlockHdr *hdr = block_hdr(m, b);
            if (hdr->count <= 0) continue;
            if (prune_block_dies(m, hdr, b, ox, oy, oz, md2)) {
                hdr->doom = 1;
                m.doomed[atomicAdd(n_doomed, 1)] = b;
            }
        }
        return;
    }
    // (count, error word and pose are asked for together, in front of the tests on them: one round trip, not three)
    const int n = count_of(n_ptr, n_imm);
    const int err0 = use_pose ? __hip_atomic_load(&state->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    SE3 pose = se3_identity();
    if (use_pose) pose = state->new_pose;
    int *touched = &m.ctr[C_TOUCHED0 + sc.parity * kCtrStride];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        m.ctr[C_TOUCHED0 + (sc.parity ^ 1) * kCtrStride] = 0;  // re-arm for the next insert
        m.ctr[C_DOOMED0 + (sc.parity ^ 1) * kCtrStride] = 0;
        // free-block queue: undo the pop cursor's overshoot of the previous insert, then admit the
        // blocks recycled since (nothing else touches these words while k_map_link runs)
        const unsigned head = (unsigned)m.ctr[C_FHEAD], tail = (unsigned)m.ctr[C_FTAIL];
        if ((int)(tail - head) < 0) m.ctr[C_FHEAD] = (int)tail;
        m.ctr[C_FTAIL] = m.ctr[C_FPEND];
    }
    // a registration that gave up (E_TIMEOUT) left no pose: the frame inserts nothing (the host replays it)
    if (use_pose && (err0 & E_TIMEOUT)) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += link_blocks * blockDim.x) {
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
                sc.rec_block[nt] = m.slots[slot].block;  // (-1: claimed a moment ago, by this insert; blocks are attached by k_map_apply only)
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
// independent of the others: record -> {slot, list} -> {block, points} is three round trips (fused: the record carries
// the block, so header and points are read beside the slot: two).
// FUSED (see above): sentenced voxels are removed instead of appended to, created voxels judged by their first point, the
// doomed list carried out, and -- with host_rec -- the frame record handed to the host by the last workgroup.
struct ApplyPrune {
    int fused;             // 0: AddPoints only
    int use_state_origin;  // origin = state->new_pose.t (else ox, oy, oz)
    const PipeState *state;
    double ox, oy, oz;
    unsigned *host_rec;  // pipeline: the frame's slot of the host-pinned ring (null: no hand-off)
    int rec_words;
};
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_map_apply(MapView m, InsertScratch sc, ApplyPrune pr) {
    constexpr int kGroups = THREADS / 32;
    __shared__ int sh_need[kGroups];
    __shared__ int sh_alloc[4];  // queue position, entries available there, bump base, blocks in the pool
    const int lane = threadIdx.x & 31;
    const int g = threadIdx.x >> 5;
    const int half_shift = threadIdx.x & 32;  // this group's half of the 64-bit wave ballot
    // (what the kernel needs before its records -- their count, the origin, the doomed list's length and this thread's first entry
    // of it: the list is k_map_link's, complete before this kernel starts -- is asked for together, in front of the first test)
    const int touched = m.ctr[C_TOUCHED0 + sc.parity * kCtrStride];
    double ox = pr.ox, oy = pr.oy, oz = pr.oz;
    if (pr.fused && pr.use_state_origin) {
        ox = pr.state->new_pose.t[0];
        oy = pr.state->new_pose.t[1];
        oz = pr.state->new_pose.t[2];
    }
    const int k_first = blockIdx.x * THREADS + threadIdx.x;
    int nd = 0, b_first = -1;
    if (pr.fused) {
        nd = m.ctr[C_DOOMED0 + sc.parity * kCtrStride];
        if (k_first < m.blocks_cap) b_first = m.doomed[k_first];  // (whatever it holds beyond nd is not used)
    }
    // the workgroups that have records to serve (at least one): the others leave at once -- in the fused form they would
    // have to sign off one by one on a single word for the frame record's sake
    const int busy = min((int)gridDim.x, max(1, (touched + kGroups - 1) / kGroups));
    if ((int)blockIdx.x >= busy) return;
    const double md2 = m.max_distance * m.max_distance;
    // workgroup-uniform trip count: the groups of a workgroup allocate their blocks together
    for (int t0 = blockIdx.x * kGroups; t0 < touched; t0 += busy * kGroups) {
        const int t = t0 + g;
        int slot = -1, L = 0, head = -1, my_idx = 0x7FFFFFFF, rb = -1;
        if (t < touched) {
            slot = sc.rec_slot[t];
            if (slot >= 0) slot = KICP_IDX(m.dbg, m.ctr + C_ERR, slot, (long long)m.mask + 1, 31);
            L = sc.rec_count[t];
            head = sc.rec_head[t];
            if (pr.fused) rb = sc.rec_block[t];
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
        // (fused: the slot of a sentenced voxel may be tombstoned under this group's eyes by the list's executor -- block and
        // count come from the record and the block's header, which the executor leaves alone until it owns the voxel)
        int b = pr.fused ? (slot >= 0 ? rb : -1) : cur.block;
        int cnt = (b >= 0) ? cur.count : 0;
        bool gone = false;  // fused: a sentenced voxel -- removed here or by the executor, nothing is appended
        if (pr.fused && b >= 0) {
            BlockHdr *hdr = block_hdr(m, b);
            // (a plain read: 0 is what k_map_link's verdict pass left -- nobody sentences a voxel in this kernel --, and anything
            // else is settled by the exchange below.  Read beside the count: no round trip of its own.)
            int d = hdr->doom;
            cnt = hdr->count;
            if (__builtin_expect(d != 0, 0)) {
                if (lane == 0) {
                    d = (atomicCAS(&hdr->doom, 1, 2) == 1) ? 1 : 2;
                    if (d == 1) voxel_remove(m, hdr, b);  // VoxelHashMap.cpp:126-129, the appended points with it
                }
                gone = true;
                cnt = 0;
            }
        }
        const bool need = slot >= 0 && b < 0;  // new voxel (VoxelHashMap.cpp:112-116)
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
                hdr->doom = 0;
                sl->block = b;
            }
        }
        __syncthreads();  // sh_need / sh_alloc are reused by the next trip
        if (slot < 0 || b < 0 || gone) continue;  // idle group, a record that lost its race, pool exhausted, or a sentenced voxel
        if (L > kRecList || m.max_points > 32) {  // long list or wide voxel: serial fallback over the chain
            if (lane == 0) {
                map_apply_voxel_serial(m, slot, head, b, sc.world, sc.next);
                if (pr.fused && need) {  // a voxel created here: judged by its first point
                    const double2 p0 = block_xy(m, b)[0];
                    if (prune_point_dies(p0.x, p0.y, block_z(m, b)[0], ox, oy, oz, md2)) voxel_remove(m, block_hdr(m, b), b);
                }
            }
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
        double nx = 0.0, ny = 0.0, nz = 0.0;  // incoming point held by this lane
        if (lane < L) {
            nx = sc.world[3 * my_idx];
            ny = sc.world[3 * my_idx + 1];
            nz = sc.world[3 * my_idx + 2];
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
            // fused: a voxel created here is judged by its first point (lane 0's); its points above land in a block that is
            // on the free ring by then -- nobody reads it before the next insert hands it out (VoxelHashMap.cpp:126-129)
            if (pr.fused && need && cnt > 0 && prune_point_dies(ex, ey, ez, ox, oy, oz, md2)) voxel_remove(m, block_hdr(m, b), b);
        }
    }
    if (pr.fused) {
        // ---- the doomed list: voxels that no record of this frame touches, and the touched ones whose group has not got there yet
        for (int k = k_first; k < nd; k += busy * THREADS) {
            const int b = KICP_IDX(m.dbg, m.ctr + C_ERR, k == k_first ? b_first : m.doomed[k], m.blocks_cap, 32);
            BlockHdr *hdr = block_hdr(m, b);
            if (atomicCAS(&hdr->doom, 1, 2) == 1) voxel_remove(m, hdr, b);
        }
        if (pr.host_rec) frame_record_handoff(m, pr.host_rec, pr.rec_words, busy);
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
    int nb = min(m.ctr[C_BUMP], m.blocks_cap);
    if (use_state_origin && (__hip_atomic_load(&state->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & E_TIMEOUT))
        nb = 0;  // no pose, nothing to prune around (the frame record is still handed over)
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        BlockHdr *hdr = block_hdr(m, b);
        if (hdr->count <= 0) continue;
        if (prune_block_dies(m, hdr, b, ox, oy, oz, md2)) voxel_remove(m, hdr, b);
    }
    if (host_rec) frame_record_handoff(m, host_rec, rec_words, (int)gridDim.x);
}

// rebuild the slot array from the live blocks (after growth, or to drop tombstones)
__global__ __launch_bounds__(256) void k_map_rehash(MapView m) {
    const int nb = min(m.ctr[C_BUMP], m.blocks_cap);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        BlockHdr *hdr = block_hdr(m, b);
        if (hdr->count <= 0) continue;
        const unsigned long long key = hdr->key;
        uint32_t s = hash_key(key, m.mask);
        bool placed = false;
        for (uint32_t probes = 0; probes <= m.mask; ++probes) {  // (bounded by the table: a slot array without room is an error, not a spin)
            const unsigned long long old = atomicCAS(&m.slots[s].key, kKeyEmpty, key);
            if (old == kKeyEmpty) {
                placed = true;
                break;
            }
            s = (s + 1) & m.mask;
        }
        if (!placed) {
            atomicOr(&m.ctr[C_ERR], E_TABLE_FULL);
            continue;
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

void launch_closest_neighbor(const MapView &m, const double *q, int nq, double *nn, double *dist,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_closest_neighbor, dim3(grid_for((long)nq * 32, 256, 2048)), dim3(256), 0, s, m, q,
                       nq, nn, dist);
}
void launch_map_link(const MapView &m, const InsertScratch &sc, const double *in, const int *n_ptr, int n_imm,
                     int n_max, const PipeState *state, int use_pose, hipStream_t s, const MapPrune *pr) {
    // fused update: the verdict pass's workgroups ride behind the link's (one block per thread on the bench's 46 k-voxel
    // map; grid-stride beyond 256 workgroups)
    const int link_blocks = grid_for(n_max, 256, 2048);
    const int scan_blocks = pr ? grid_for(pr->bump_ub, 256, 256) : 0;
    hipLaunchKernelGGL(k_map_link, dim3(link_blocks + scan_blocks), dim3(256), 0, s, m, sc, in, n_ptr, n_imm, state, use_pose,
                       scan_blocks, pr ? pr->use_state_origin : 0, pr ? pr->origin[0] : 0.0, pr ? pr->origin[1] : 0.0,
                       pr ? pr->origin[2] : 0.0);
}
void launch_map_apply(const MapView &m, const InsertScratch &sc, int n_max, hipStream_t s, const MapPrune *pr, hipEvent_t done) {
    // one 32-lane group per voxel record (at most one record per incoming point).  A workgroup makes
    // ONE allocation (three returning atomics on shared words) per trip for all its groups, so larger
    // workgroups mean fewer serialised atomics.
    ApplyPrune ap = {};
    if (pr) {
        ap.fused = 1;
        ap.use_state_origin = pr->use_state_origin;
        ap.state = pr->state;
        ap.ox = pr->origin[0];
        ap.oy = pr->origin[1];
        ap.oz = pr->origin[2];
        ap.host_rec = pr->host_rec;
        ap.rec_words = pr->rec_words;
    }
    const int threads = (int)options().map_apply_threads;
    // (a dispatch that carries a completion signal ends with a system-scope release: only when an event is asked for)
#define KICP_LAUNCH_APPLY(T, CAP)                                                                                                  \
    do {                                                                                                                           \
        if (done)                                                                                                                  \
            hipExtLaunchKernelGGL(k_map_apply<T>, dim3(grid_for((long)n_max * 32, T, CAP)), dim3(T), 0, s, nullptr, done, 0, m, sc, ap); \
        else                                                                                                                       \
            hipLaunchKernelGGL(k_map_apply<T>, dim3(grid_for((long)n_max * 32, T, CAP)), dim3(T), 0, s, m, sc, ap);               \
    } while (0)
    if (threads >= 1024)
        KICP_LAUNCH_APPLY(1024, 1024);
    else if (threads >= 512)
        KICP_LAUNCH_APPLY(512, 2048);
    else
        KICP_LAUNCH_APPLY(256, 2048);
#undef KICP_LAUNCH_APPLY
}
void launch_map_prune(const MapView &m, long bump_ub, const PipeState *state, int use_state_origin,
                      const double origin[3], unsigned *host_rec, int rec_words, hipStream_t s, hipEvent_t done) {
    // grid-stride over the blocks; at most 256 workgroups: every one of them signs off with a fenced
    // atomic (frame-record hand-off), which would serialise over thousands of workgroups
    if (done)
        hipExtLaunchKernelGGL(k_map_prune, dim3(grid_for(bump_ub, 256, 256)), dim3(256), 0, s, nullptr, done, 0, m, state,
                              use_state_origin, origin ? origin[0] : 0.0, origin ? origin[1] : 0.0,
                              origin ? origin[2] : 0.0, host_rec, rec_words);
    else
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
