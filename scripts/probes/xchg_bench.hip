// Micro-benchmark of k_icp's in-launch exchange, alone: G workgroups of 512 threads (one per CU: they ask for most of the LDS)
// run `iters` rounds of  [busy-wait "work"] -> workgroup sums -> publish (tagged 16-byte granule pairs, sc1) -> leaders gather
// their members and publish group sums -> everybody gathers the leaders' sums -> row sums -> [busy-wait "solve"],  the same
// data structures, instructions and reduction order as kicp_icp.hip, with the association taken out.  Workgroup 0 (a leader)
// timestamps the stages of every round; the sums are checked against their closed form.  Variants of the exchange are
// template parameters, all timed in one process on one box.
//   hipcc -O3 --offload-arch=gfx950 -o xchg_bench xchg_bench.hip && ./xchg_bench [G=224] [iters=400]
// (A diagnostic, not product code: nothing links it.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#ifndef XCHG_ROWS
#define XCHG_ROWS 16
#endif
constexpr int kThreads = 512, kSumsMax = 19, kRows = XCHG_ROWS, kMaxBlocks = 256;  // (-DXCHG_ROWS=32: room for 8 leaders of 28 members / 32 leaders)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void load_pair(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned long long &a, unsigned long long &b) {
    const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
    a = (unsigned long long)(unsigned)v.x | ((unsigned long long)(unsigned)v.y << 32);
    b = (unsigned long long)(unsigned)v.z | ((unsigned long long)(unsigned)v.w << 32);
}
__device__ __forceinline__ void store_pair(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, double val) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(val);
    v4i v;
    v.x = (int)(unsigned)bits;
    v.y = (int)tag;
    v.z = (int)(unsigned)(bits >> 32);
    v.w = (int)tag;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 16);
}
__device__ __forceinline__ int tid_now() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ double row_sum(const double *row, int count) {
    const double2 *r2 = reinterpret_cast<const double2 *>(row);
    double a[kRows];
#pragma unroll
    for (int i = 0; i < kRows / 2; ++i) {
        const double2 t = r2[i];
        a[2 * i] = t.x;
        a[2 * i + 1] = t.y;
    }
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const double w = v + a[j];
        v = j < count ? w : v;
    }
    return v;
}
// the sum of the 16 lanes of a DPP row, in every lane of it: four steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), the
// same tree whatever the lane -- a + b on one side is b + a on the other
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double row16_sum(double v) {
    v = v + dpp_f64<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v = v + dpp_f64<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v = v + dpp_f64<0x141>(v);  // row_half_mirror
    v = v + dpp_f64<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ void busy(unsigned ticks) {  // 10 ns ticks
    const unsigned long long t0 = wall_clock64();
    while ((unsigned)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}

struct Args {
    unsigned long long *gran;      // [2][G][kSums] pairs
    unsigned long long *grp_gran;  // [2][NG][kSums] pairs
    unsigned long long *mailbox;   // [2][G][NG][kSums] pairs (MODE 2)
    int G, iters;
    unsigned epoch_base, spin_limit;
    unsigned work_ticks, jitter_ticks, solve_ticks;
    unsigned tail_permille, tail_ticks;  // with this probability a workgroup's round is longer by tail_ticks (a search with a list build)
    unsigned *stamps;  // [iters][6]: workgroup 0's stage ends, 10 ns ticks from the round's start
    int *err;
};

// MODE 0: the product's form (every lane of the sweep polls its own pair, both hops)
// MODE 1: second hop with two pairs per lane (half the polling lanes, two loads in flight each)
// MODE 2: second hop PUSHED: a leader stores its group sums into every workgroup's own mailbox; a workgroup polls lines nobody else reads
// MODE 3: second hop with four pairs per lane
// MODE 4: no LDS staging in either hop: the lane that sums scalar k fetches all its addends itself (16 loads in flight) and adds from registers
// KS: scalars a workgroup exchanges (the product: 19 -- 16 of the normal equations, two counts, the profiling build's slot)
// ST: 16-byte slots between one workgroup's block of pairs and the next (the product: 19 -- blocks of 304 bytes, so that most
//     128-byte lines are written by TWO workgroups, which as a rule sit on different XCDs)
template <int NG, int MODE, int SLEEP, bool STAMPS = true, int KS = kSumsMax, int ST = KS, bool GM = false>
__global__ __launch_bounds__(kThreads) void k_xchg(Args P) {
    constexpr int kSums = KS;
    // GM: the blocks of a leader's members lie side by side (block of workgroup w at (w mod NG) * 16 + w / NG) instead of NG blocks apart
    auto place = [](int w) { return GM ? (w % NG) * kRows + w / NG : w; };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *part_t = reinterpret_cast<double *>(smem);  // [kSums][kRows]
    double *sums_t = part_t + kSumsMax * kRows;          // [kSums][kRows]
    double *tot = sums_t + kSumsMax * kRows;             // [kSums]
    int *fail = reinterpret_cast<int *>(tot + kSumsMax + 1);
    const int G = P.G;
    const int b = (int)blockIdx.x;
    if (threadIdx.x == 0) *fail = 0;
    __syncthreads();
    unsigned rng = 12345u + 977u * (unsigned)b;
    for (int it = 0; it < P.iters; ++it) {
        const int tid = tid_now();
        const int ck = tid % 18, cg = tid / 18;
        const unsigned long long t_start = STAMPS ? wall_clock64() : 0ull;
        // ---- "association": a fixed wait plus this workgroup's share of the jitter
        rng = rng * 1664525u + 1013904223u;
        unsigned extra = P.jitter_ticks ? (rng >> 8) % P.jitter_ticks : 0u;
        rng = rng * 1664525u + 1013904223u;
        if (P.tail_permille && (rng >> 10) % 1000u < P.tail_permille) extra += P.tail_ticks;
        busy(P.work_ticks + extra);
        const unsigned s0 = STAMPS ? (unsigned)(wall_clock64() - t_start) : 0u;
        // ---- workgroup reduction
        constexpr bool DPUB = MODE == 5 || MODE == 8 || MODE == 9 || MODE == 12;  // the workgroup's own reduction by DPP rows
        if (!DPUB) {
            if (cg < kRows) part_t[ck * kRows + cg] = (cg == 0) ? (double)(b + 1) * (double)(ck + 1) + (double)it : 0.0;
            if (tid < kRows) part_t[18 * kRows + tid] = 0.0;
            __syncthreads();
        }
        const unsigned epoch = P.epoch_base + (unsigned)it + 1u;
        unsigned long long *gran = P.gran + (size_t)(it & 1) * kMaxBlocks * (2 * ST);
        const __amdgpu_buffer_rsrc_t gran_r = rsrc_of(gran, (unsigned)(kMaxBlocks * 2 * ST * 8));
        const int rk = tid >> 4, rl = tid & 15;  // MODE 5: scalar and addend of this lane (a DPP row per scalar)
        if (DPUB) {
            if (rk < kSums) {
                const double mine = (rl == 0 && rk < 18) ? (double)(b + 1) * (double)(rk + 1) + (double)it : 0.0;
                const double v = row16_sum(mine);
                if (rl == 0) store_pair(gran_r, (unsigned)((b * ST + rk) * 16), epoch, v);
            }
        } else if (tid < kSums) store_pair(gran_r, (unsigned)((place(b) * ST + tid) * 16), epoch, row_sum(part_t + tid * kRows, kRows));
        const unsigned s1 = STAMPS ? (unsigned)(wall_clock64() - t_start) : 0u;
        auto poll = [&](const __amdgpu_buffer_rsrc_t &r, unsigned off, double &out) -> bool {
            unsigned long long lo, hi;
            load_pair(r, off, lo, hi);
            unsigned spins = 0;
            while ((unsigned)(lo >> 32) != epoch || (unsigned)(hi >> 32) != epoch) {
                if (++spins > P.spin_limit) return false;
                __builtin_amdgcn_s_sleep(SLEEP);
                load_pair(r, off, lo, hi);
            }
            out = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
            return true;
        };
        const int ng = NG < G ? NG : G;
        unsigned long long *grp_gran = P.grp_gran + (size_t)(it & 1) * 64 * (2 * ST);
        const __amdgpu_buffer_rsrc_t grp_r = rsrc_of(grp_gran, (unsigned)(64 * 2 * ST * 8));
        unsigned long long *mbox = P.mailbox + (size_t)(it & 1) * kMaxBlocks * 64 * (2 * kSumsMax);
        const __amdgpu_buffer_rsrc_t mbox_r = rsrc_of(mbox, (unsigned)((size_t)kMaxBlocks * 64 * 2 * kSumsMax * 8));
        unsigned s2 = s1, s3 = s1;
        // fetch `count` pairs {base + i * stride} until all carry the tag, add them in order (MODE 4)
        auto gather_sum = [&](const __amdgpu_buffer_rsrc_t &r, unsigned base_off, unsigned stride, int count, double &out) -> bool {
            unsigned long long lo[kRows], hi[kRows];
            unsigned spins = 0;
            for (;;) {
#pragma unroll
                for (int u = 0; u < kRows; ++u) load_pair(r, base_off + (unsigned)(u < count ? u : 0) * stride, lo[u], hi[u]);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < kRows; ++u) ok &= (unsigned)(lo[u] >> 32) == epoch && (unsigned)(hi[u] >> 32) == epoch;
                if (ok) break;
                if (++spins > P.spin_limit) return false;
                __builtin_amdgcn_s_sleep(SLEEP);
            }
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const double w = v + __longlong_as_double((long long)(((unsigned long long)(unsigned)hi[u] << 32) | (unsigned)lo[u]));
                v = u < count ? w : v;
            }
            out = v;
            return true;
        };
        if (MODE == 12) {
            // the leader's gather inside waves (a wave takes four scalars of all members: four DPP row operations, no LDS staging, no
            // barrier in front of the group sums); the second hop staged as in the product
            const int wv = tid >> 6, kk = (tid >> 4) & 3, jj = tid & 15;
            const int k8 = wv * 4 + kk;
            if (b < ng && k8 < kSums) {
                const int members = (G - b + ng - 1) / ng;
                double v = 0.0;
                if (jj < members && !poll(gran_r, (unsigned)((place(b + ng * jj) * ST + k8) * 16), v)) *fail = 1;
                v = row16_sum(v);
                if (jj < 8) store_pair(mbox_r, (unsigned)(((jj * 64 + b) * ST + k8) * 16), epoch, v);
            }
        } else if (MODE == 8) {
            // a wave takes four scalars of all addends (lane 16 kk + j): the reduction stays inside the wave -- four DPP row operations,
            // no LDS staging, no workgroup barrier between the hops; a wave's poll asks for one 64-byte piece of every block
            const int wv = tid >> 6, kk = (tid >> 4) & 3, jj = tid & 15;
            const int k8 = wv * 4 + kk;
            const int c = b % 8;
            if (b < ng && k8 < kSums) {
                const int members = (G - b + ng - 1) / ng;
                double v = 0.0;
                if (jj < members && !poll(gran_r, (unsigned)((place(b + ng * jj) * ST + k8) * 16), v)) *fail = 1;
                v = row16_sum(v);
                if (jj < 8) store_pair(mbox_r, (unsigned)(((jj * 64 + b) * ST + k8) * 16), epoch, v);  // (lane c of the row stores copy c)
            }
            if (k8 < kSums) {
                double v = 0.0;
                if (jj < ng && !poll(mbox_r, (unsigned)(((c * 64 + jj) * ST + k8) * 16), v)) *fail = 1;
                v = row16_sum(v);
                if (jj == 0) tot[k8] = v;
            }
            __syncthreads();
        } else if (MODE == 5) {
            if (b < ng && rk < kSums) {
                const int members = (G - b + ng - 1) / ng;
                double v = 0.0;
                if (rl < members && !poll(gran_r, (unsigned)(((b + ng * rl) * ST + rk) * 16), v)) *fail = 1;
                v = row16_sum(v);
                if (rl == 0) store_pair(grp_r, (unsigned)((b * ST + rk) * 16), epoch, v);
            }
            if (rk < kSums) {
                double v = 0.0;
                if (rl < ng && !poll(grp_r, (unsigned)((rl * ST + rk) * 16), v)) *fail = 1;
                v = row16_sum(v);
                if (rl == 0) tot[rk] = v;
            }
            __syncthreads();
        } else if (MODE == 4) {
            if (b < ng) {
                const int members = (G - b + ng - 1) / ng;
                if (tid < kSums) {
                    double v = 0.0;
                    if (!gather_sum(gran_r, (unsigned)((b * ST + tid) * 16), (unsigned)(ng * ST * 16), members, v)) *fail = 1;
                    else store_pair(grp_r, (unsigned)((b * ST + tid) * 16), epoch, v);
                }
            }
            if (tid < kSums) {
                double v = 0.0;
                if (!gather_sum(grp_r, (unsigned)(tid * 16), (unsigned)(ST * 16), ng, v)) *fail = 1;
                tot[tid] = v;
            }
            __syncthreads();
        } else if (b < ng) {
            const int members = (G - b + ng - 1) / ng;
            constexpr int kParts = kThreads / kSums;
            if (tid < kParts * kSums) {
                const int k = tid % kSums;
                for (int j = tid / kSums; j < members; j += kParts) {
                    double v = 0.0;
                    if (!poll(gran_r, (unsigned)((place(b + ng * j) * ST + k) * 16), v)) *fail = 1;
                    sums_t[k * kRows + j] = v;
                }
            }
            __syncthreads();
            s2 = STAMPS ? (unsigned)(wall_clock64() - t_start) : 0u;
            if (MODE == 2) {
                // every lane takes (workgroup, scalar) pairs: the group's sums into every mailbox
                if (tid < kSums) tot[tid] = row_sum(sums_t + tid * kRows, members);
                __syncthreads();
                if (!*fail)
                    for (int e = tid; e < G * kSums; e += kThreads) {
                        const int w = e / kSums, k = e % kSums;
                        store_pair(mbox_r, (unsigned)(((w * NG + b) * kSums + k) * 16), epoch, tot[k]);
                    }
            } else if (MODE == 6 || MODE == 7 || MODE == 9 || MODE == 10 || MODE == 11) {
                // a copy of the group's sums per XCD (MODE 6: 8 copies, MODE 7: 2): a line is polled by 28 (112) workgroups, not 224
                constexpr int COPIES = MODE == 7 ? 2 : (MODE == 10 ? 4 : (MODE == 11 ? 16 : 8));
                if (tid < kSums) {
                    const double v = row_sum(sums_t + tid * kRows, members);
                    if (!*fail) {
#pragma unroll
                        for (int c = 0; c < COPIES; ++c) store_pair(mbox_r, (unsigned)(((c * 64 + b) * ST + tid) * 16), epoch, v);
                    }
                }
            } else if (tid < kSums) {
                const double v = row_sum(sums_t + tid * kRows, members);
                if (!*fail) store_pair(grp_r, (unsigned)((b * ST + tid) * 16), epoch, v);
            }
            __syncthreads();
            s3 = STAMPS ? (unsigned)(wall_clock64() - t_start) : 0u;
        }
        // ---- second hop
        if (MODE == 4 || MODE == 5 || MODE == 8) {
        } else if (MODE == 0) {
            for (int e = tid; e < ng * kSums; e += kThreads) {
                const int k = e % kSums, g = e / kSums;
                double v = 0.0;
                if (!poll(grp_r, (unsigned)((g * ST + k) * 16), v)) *fail = 1;
                sums_t[k * kRows + g] = v;
            }
        } else if (MODE == 6 || MODE == 7 || MODE == 9 || MODE == 10 || MODE == 11 || MODE == 12) {
            constexpr int COPIES = MODE == 7 ? 2 : (MODE == 10 ? 4 : (MODE == 11 ? 16 : 8));
            const int c = b % COPIES;
            for (int e = tid; e < ng * kSums; e += kThreads) {
                const int k = e % kSums, g = e / kSums;
                double v = 0.0;
                if (!poll(mbox_r, (unsigned)(((c * 64 + g) * ST + k) * 16), v)) *fail = 1;
                sums_t[k * kRows + g] = v;
            }
        } else if (MODE == 2) {
            for (int e = tid; e < ng * kSums; e += kThreads) {
                const int k = e % kSums, g = e / kSums;
                double v = 0.0;
                if (!poll(mbox_r, (unsigned)(((b * NG + g) * kSums + k) * 16), v)) *fail = 1;
                sums_t[k * kRows + g] = v;
            }
        } else {
            constexpr int PER = MODE == 1 ? 2 : 4;
            const int pairs = ng * kSums;
            const int e0 = tid * PER;
            if (e0 < pairs) {
                unsigned long long lo[PER], hi[PER];
                unsigned spins = 0;
                bool done = false;
                while (!done) {
#pragma unroll
                    for (int u = 0; u < PER; ++u) load_pair(grp_r, (unsigned)(min(e0 + u, pairs - 1) * 16), lo[u], hi[u]);
                    done = true;
#pragma unroll
                    for (int u = 0; u < PER; ++u) done &= (unsigned)(lo[u] >> 32) == epoch && (unsigned)(hi[u] >> 32) == epoch;
                    if (!done) {
                        if (++spins > P.spin_limit) {
                            *fail = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(SLEEP);
                    }
                }
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int e = e0 + u;
                    if (e < pairs) sums_t[(e % kSums) * kRows + e / kSums] = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi[u] << 32) | (unsigned)lo[u]));
                }
            }
        }
        __syncthreads();
        const unsigned s4 = STAMPS ? (unsigned)(wall_clock64() - t_start) : 0u;
        if (MODE != 4 && MODE != 5 && MODE != 8 && tid < kSums) tot[tid] = row_sum(sums_t + tid * kRows, ng);
        if (*fail) {
            if (tid == 0) atomicOr(P.err, 1);
            break;
        }
        __syncthreads();
        if (tid < (kSums < 18 ? kSums : 18)) {
            const double want = (double)(tid + 1) * (double)G * (double)(G + 1) * 0.5 + (double)G * (double)it;
            if (tot[tid] != want) atomicOr(P.err, 2);
        }
        busy(P.solve_ticks);
        const unsigned s5 = STAMPS ? (unsigned)(wall_clock64() - t_start) : 0u;
        if (STAMPS && b == 0 && tid == 0) {
            unsigned *r = P.stamps + (size_t)it * 6;
            r[0] = s0;
            r[1] = s1;
            r[2] = s2;
            r[3] = s3;
            r[4] = s4;
            r[5] = s5;
        }
        __syncthreads();
    }
}

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

static unsigned g_epoch = 1000;

template <int NG, int MODE, int SLEEP, bool STAMPS = true, int KS = kSumsMax, int ST = KS, bool GM = false>
static void run(const char *name, int G, int iters, unsigned work, unsigned jitter, unsigned solve, Args A) {
    const int lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xchg<NG, MODE, SLEEP, STAMPS, KS, ST, GM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    A.G = G;
    A.iters = iters;
    A.work_ticks = work;
    A.jitter_ticks = jitter;
    A.solve_ticks = solve;
    A.spin_limit = 1u << 18;
    double best = 1e30;
    std::vector<unsigned> st((size_t)iters * 6);
    int err = 0;
    for (int rep = 0; rep < 3; ++rep) {
        A.epoch_base = g_epoch;
        g_epoch += (unsigned)iters + 8u;
        CK(hipMemset(A.err, 0, sizeof(int)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_xchg<NG, MODE, SLEEP, STAMPS, KS, ST, GM>), dim3(G), dim3(kThreads), lds, 0, A);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&err, A.err, sizeof(int), hipMemcpyDeviceToHost));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(st.data(), A.stamps, st.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        }
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
        if (err) break;
    }
    // medians of workgroup 0's stages over the rounds (the first 20 left out), in microseconds
    double med[6];
    for (int c = 0; c < 6; ++c) {
        std::vector<unsigned> v;
        for (int it = 20; it < iters; ++it) v.push_back(st[(size_t)it * 6 + c] - (c ? st[(size_t)it * 6 + c - 1] : 0u));
        std::sort(v.begin(), v.end());
        med[c] = v.empty() ? 0.0 : v[v.size() / 2] * 0.01;
    }
    printf("%-34s G %3d work %4.1f jitter %4.1f solve %3.1f | round %6.3f us  exchange %6.3f | wg0: work %5.2f publish %5.2f hop1 %5.2f group-publish %5.2f hop2 %5.2f tail %5.2f%s\n",
           name, G, work * 0.01, jitter * 0.01, solve * 0.01, best * 1e3 / iters, best * 1e3 / iters - (work + jitter * 0.5 + solve) * 0.01, med[0], med[1],
           med[2], med[3], med[4], med[5], err ? "  ** FAILED **" : "");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 224;
    const int iters = argc > 2 ? atoi(argv[2]) : 400;
    Args A;
    memset(&A, 0, sizeof A);
    const size_t gran_bytes = (size_t)2 * kMaxBlocks * 2 * 32 * 8, grp_bytes = (size_t)2 * 64 * 2 * 32 * 8,
                 mbox_bytes = (size_t)2 * kMaxBlocks * 64 * 2 * kSumsMax * 8;
    CK(hipMalloc(&A.gran, gran_bytes));
    CK(hipMalloc(&A.grp_gran, grp_bytes));
    CK(hipMalloc(&A.mailbox, mbox_bytes));
    CK(hipMalloc(&A.stamps, (size_t)iters * 6 * sizeof(unsigned)));
    CK(hipMalloc(&A.err, sizeof(int)));
    CK(hipMemset(A.gran, 0, gran_bytes));
    CK(hipMemset(A.grp_gran, 0, grp_bytes));
    CK(hipMemset(A.mailbox, 0, mbox_bytes));
    A.tail_permille = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned jitter = pass == 0 ? 0u : 150u;
        run<16, 9, 2, false, 18, 24>("product form now", G, iters, 200, jitter, 140, A);
        run<16, 12, 2, false, 18, 24>("  the leader's gather inside waves", G, iters, 200, jitter, 140, A);
        run<16, 9, 2, false, 18, 24>("product form now, again", G, iters, 200, jitter, 140, A);
        run<16, 12, 2, false, 18, 24>("  inside waves, again", G, iters, 200, jitter, 140, A);
    }
    run<16, 12, 2, true, 18, 24>("the leader's gather inside waves, stamps", G, iters, 200, 0, 140, A);
    return 0;
}
