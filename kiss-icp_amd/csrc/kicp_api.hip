// kicp_api.hip -- the C-ABI of libkicp (include/kicp.h): handle management, HBM buffers, launch
// sequencing on one HIP stream per handle.  No CPU fallback exists: without a gfx950 device
// every create call fails loudly with KICP_ERR_NO_DEVICE.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "kicp_launch.hpp"
#include "kicp_numa.hpp"

namespace kicp {

// ---- errors / options ------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }
Options &options() {
    static Options o;
    return o;
}

static inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
// KICP_HOST_TRACE=1: one line per queued frame on stderr (where the host side of a frame spent its time)
static bool host_trace_on() {
    static const bool on = [] {
        const char *e = getenv("KICP_HOST_TRACE");
        return e && *e && *e != '0';
    }();
    return on;
}

// ---- bounded waits --------------------------------------------------------------------------------------------------
// (declared in kicp_internal.hpp, which says why)
static thread_local bool t_gave_up = false;  // this thread's most recent wait ended in KICP_ERR_TIMEOUT
// relaxed: nobody is waiting for the RESULT of what is waited for -- queue back-pressure: the caller is frames ahead of the
// device -- so the wait goes to sleep at once instead of yielding in a loop: a yield loop is a whole core per stream, and eight
// ranks of it on the 16 CPUs a GPU box's container gets run into the cgroup's quota (round 5's closing session: 2.4 cores
// busy per stream, half of it this loop).
template <class Query>
static int wait_poll(Query query, const char *kind, const char *what, bool relaxed = false, long limit_ms = -1) {
    const double t0 = now_ms();
    const double limit = (double)(limit_ms >= 0 ? limit_ms : options().wait_timeout_ms);
    for (;;) {
        const hipError_t e = query();
        if (e == hipSuccess) {
            t_gave_up = false;
            return KICP_OK;
        }
        (void)hipGetLastError();  // hipErrorNotReady is an answer, not an error
        if (e != hipErrorNotReady) {
            set_error("%s(%s) failed: %s", kind, what, hipGetErrorString(e));
            return KICP_ERR_HIP;
        }
        const double dt = now_ms() - t0;
        if (limit > 0.0 && dt > limit) {
            t_gave_up = true;
            set_error("%s: the device did not finish within %ld ms (%s); the work is still queued", what, limit_ms >= 0 ? limit_ms : options().wait_timeout_ms,
                      limit_ms >= 0 ? "collective_timeout_ms" : "wait_timeout_ms");
            return KICP_ERR_TIMEOUT;
        }
        // a registration is a fraction of a millisecond: poll closely at first, then leave the core to others
        if (relaxed) {
            std::this_thread::sleep_for(std::chrono::microseconds(dt < 5.0 ? 40 : 500));
        } else if (dt < 0.05) {
            for (int k = 0; k < 8; ++k) _mm_pause();
        } else if (dt < 2.0) {
            std::this_thread::yield();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(dt < 50.0 ? 50 : 500));
        }
    }
}
static bool wait_blocking() {  // KICP_WAIT_BLOCKING=1 (diagnostics): the runtime's own unbounded waits instead of the polls
    static const bool on = [] {
        const char *e = getenv("KICP_WAIT_BLOCKING");
        return e && *e && *e != '0';
    }();
    return on;
}
int wait_stream(hipStream_t s, const char *what) {
    if (wait_blocking()) {
        KICP_HIP(hipStreamSynchronize(s));
        return KICP_OK;
    }
    return wait_poll([s] { return hipStreamQuery(s); }, "hipStreamQuery", what);
}
// ... for work that waits on PEERS (an all-gather on the exchange stream): the collective's deadline, not the device's
int wait_stream_peers(hipStream_t s, const char *what) {
    if (wait_blocking()) {
        KICP_HIP(hipStreamSynchronize(s));
        return KICP_OK;
    }
    const long ms = options().collective_timeout_ms > 0 ? options().collective_timeout_ms : options().wait_timeout_ms;
    return wait_poll([s] { return hipStreamQuery(s); }, "hipStreamQuery", what, false, ms);
}
int wait_event(hipEvent_t e, const char *what, bool relaxed) {
    if (wait_blocking()) {
        KICP_HIP(hipEventSynchronize(e));
        return KICP_OK;
    }
    return wait_poll([e] { return hipEventQuery(e); }, "hipEventQuery", what, relaxed && options().relaxed_backpressure != 0);
}

struct StreamRegistry {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> v;
};
static StreamRegistry &stream_registry() {
    static StreamRegistry *r = new StreamRegistry();  // (never destroyed: handles may be released from static destructors)
    return *r;
}
void stream_register(int device_id, hipStream_t s) {
    if (!s) return;
    StreamRegistry &r = stream_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    r.v.emplace_back(device_id, s);
}
void stream_forget(hipStream_t s) {
    StreamRegistry &r = stream_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    for (size_t i = 0; i < r.v.size(); ++i)
        if (r.v[i].second == s) {
            r.v[i] = r.v.back();
            r.v.pop_back();
            return;
        }
}
// A POINT-IN-TIME wait, like the hipFree / hipDeviceSynchronize it stands in for: an event is recorded on every stream the
// library has on the device, under the registry's lock, and the EVENTS are polled -- work another thread queues on its own
// pipeline after this call started is not waited for.  (Round 5 polled hipStreamQuery per stream, i.e. "until the stream is
// idle": a second pipeline that is kept fed on the same device never is, and the buffer growth of the first starved into a
// spurious KICP_ERR_TIMEOUT.)  null_stream: also what was queued on the NULL stream (kicp_device_synchronize: a caller's own
// kernels that produced a buffer handed to the *_device entries).
int wait_device(int device_id, const char *what, bool null_stream) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device_id) KICP_HIP(hipSetDevice(device_id));
    std::vector<hipEvent_t> evs;
    std::vector<hipStream_t> unmarked;  // (an event could not be had for these: polled as streams)
    {
        StreamRegistry &r = stream_registry();
        std::lock_guard<std::mutex> lk(r.mu);
        for (const auto &e : r.v) {
            if (e.first != device_id) continue;
            if (hipStreamQuery(e.second) == hipSuccess) continue;  // nothing pending: nothing to mark
            (void)hipGetLastError();
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, e.second) == hipSuccess) {
                evs.push_back(ev);
            } else {
                (void)hipGetLastError();
                if (ev) (void)hipEventDestroy(ev);
                unmarked.push_back(e.second);
            }
        }
    }
    int st = KICP_OK;
    for (hipEvent_t ev : evs) {
        if (st == KICP_OK) st = wait_event(ev, what, false);
        (void)hipEventDestroy(ev);  // (deferred by the runtime while the event is pending)
    }
    for (hipStream_t s : unmarked)
        if (st == KICP_OK) st = wait_stream(s, what);
    if (st == KICP_OK && null_stream) st = wait_stream(nullptr, what);
    if (prev >= 0 && prev != device_id) (void)hipSetDevice(prev);
    return st;
}
struct UtilStreams {
    std::mutex mu;
    hipStream_t s[64] = {nullptr};
};
hipStream_t util_stream(int device_id) {
    static UtilStreams *u = new UtilStreams();
    std::lock_guard<std::mutex> lk(u->mu);
    hipStream_t &s = u->s[device_id & 63];
    if (!s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            s = nullptr;
            return nullptr;
        }
        stream_register(device_id, s);
    }
    return s;
}
int stream_destroy(hipStream_t s) {
    if (!s) return KICP_OK;
    const int st = wait_stream(s, "stream teardown");
    // leaked: hipStreamDestroy would wait for the work that does not end.  The stream STAYS in the registry: whoever frees
    // memory on this device later must keep finding it busy (and give up in time) rather than walk into hipFree's own wait.
    if (st == KICP_ERR_TIMEOUT) return st;
    stream_forget(s);
    (void)hipStreamDestroy(s);
    return KICP_OK;
}

// test hook: a kernel that occupies the stream for a while (the dependency that is not signalled in time)
__global__ void k_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
void inject_stall(hipStream_t s) {
    const long ms = options().inject_stall_ms;
    if (ms <= 0) return;
    options().inject_stall_ms = 0;  // once
    hipLaunchKernelGGL(k_stall, dim3(1), dim3(64), 0, s, (unsigned long long)ms * 100000ull);
}

static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}

int DevBuf::reserve(size_t need, bool keep, hipStream_t s) {
    if (need <= bytes && p) return KICP_OK;
    size_t nb = need < 256 ? 256 : need;
    if (keep && bytes) nb = nb < bytes * 2 ? bytes * 2 : nb;
    if (!s) s = util_stream(current_device());
    if (!s) {
        set_error("no utility stream on device %d", current_device());
        return KICP_ERR_HIP;
    }
    void *np = nullptr;
    KICP_HIP(hipMalloc(&np, nb));
    // zeros, not whatever the allocator hands out: the contract of this type (see its declaration)
    hipError_t e = hipMemsetAsync(np, 0, nb, s);
    if (e == hipSuccess && keep && p && bytes) e = hipMemcpyAsync(np, p, bytes, hipMemcpyDeviceToDevice, s);
    int st = e == hipSuccess ? wait_stream(s, "buffer initialisation") : KICP_ERR_HIP;
    // the old buffer: hipFree waits for the whole device, so that wait is made here, with a deadline
    if (st == KICP_OK && p) st = wait_device(current_device(), "buffer growth (work that may still read the old buffer)");
    if (st != KICP_OK) {
        if (e != hipSuccess) set_error("device buffer initialisation failed: %s", hipGetErrorString(e));
        if (st != KICP_ERR_TIMEOUT) (void)hipFree(np);  // (timeout: both buffers stay -- hipFree would not return)
        return st;
    }
    if (p) KICP_HIP(hipFree(p));
    p = np;
    bytes = nb;
    return KICP_OK;
}
void DevBuf::release() {
    // hipFree synchronises with the device: only behind a bounded wait (a device that does not answer keeps the buffer)
    if (p) drop(wait_device(current_device(), "buffer release") != KICP_ERR_TIMEOUT);
}
void DevBuf::drop(bool device_idle) {
    if (p && device_idle) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

int check_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s); libkicp has no CPU fallback",
                  e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return KICP_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        set_error("device_id %d out of range (have %d)", device_id, n);
        return KICP_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    KICP_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libkicp is built for gfx950 (MI355X) only", device_id,
                  prop.gcnArchName);
        return KICP_ERR_NO_DEVICE;
    }
    KICP_HIP(hipSetDevice(device_id));
    return KICP_OK;
}

static uint32_t next_pow2(size_t v) {
    uint32_t c = 1024;
    while (c < v && c < (1u << 30)) c <<= 1;
    return c;
}

// the device's record of the bounds-asserting build
struct BoundsSlot {
    std::mutex mu;
    BoundsRec *p = nullptr;
};
static BoundsSlot &bounds_slot(int device_id) {
    static BoundsSlot slots[64];
    return slots[device_id & 63];
}
BoundsRec *bounds_rec(int device_id) {
    BoundsSlot &b = bounds_slot(device_id);
    std::lock_guard<std::mutex> lk(b.mu);
    if (!b.p) {
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(BoundsRec)) != hipSuccess) return nullptr;
        if (hipMemset(q, 0, sizeof(BoundsRec)) != hipSuccess) {
            (void)hipFree(q);
            return nullptr;
        }
        b.p = static_cast<BoundsRec *>(q);
    }
    return b.p;
}

static int err_bits_to_status(int bits) {
    if (!bits) return KICP_OK;
    if (bits & E_BOUNDS) {  // (only the bounds-asserting build raises it)
        BoundsRec r;
        memset(&r, 0, sizeof r);
        BoundsRec *d = bounds_rec(current_device());
        if (d && hipMemcpy(&r, d, sizeof r, hipMemcpyDeviceToHost) == hipSuccess) (void)hipMemset(d, 0, sizeof r);
        set_error("bounds check failed on the device: source line %d (tag %d), index %lld, limit %lld, workgroup %d, thread %d (bits 0x%x)", r.line, r.tag,
                  r.idx, r.limit, r.block, r.thread, bits);
        return KICP_ERR_HIP;
    }
    if (bits & E_TIMEOUT) {
        set_error("a bounded in-kernel wait gave up (ICP workgroups not co-resident?)");
        return KICP_ERR_TIMEOUT;
    }
    if (bits & E_RANGE) {
        set_error("voxel coordinate outside +-2^20 (point too far from the origin for this voxel size)");
        return KICP_ERR_RANGE;
    }
    set_error("device table full (bits 0x%x)", bits);
    return KICP_ERR_CAPACITY;
}

// ---- the persistent ICP kernel and the device it shares -------------------------------------------
// k_icp's workgroups exchange partial sums inside the launch, so all of them must be resident at the
// same time.  Two things guarantee that: (1) the grids that can be in flight together never exceed what the device can
// hold (occupancy query x CU count of THIS device or partition, at the launch's LDS size, divided among the streams that
// share the device: option "icp_device_streams"); (2) launches of k_icp on one device pass a gate of as many LANES as
// there are shares.  A stream keeps its lane from launch to launch; a stream without one takes a free lane, or queues
// behind the last launch of the lane it takes over (hipStreamWaitEvent) -- so at most `lanes` registrations run at
// once, each with 1 / lanes of the grid, whatever the number of handles.  One share (the default): the launches of all
// handles are ordered behind each other, each with the whole grid -- two half-resident grids would wait for each other
// until their bounded spins give up.
constexpr int kIcpMaxLanes = 8;
struct IcpDeviceGate {
    std::mutex mu;
    int lanes = 1;  // the largest share count among the LIVE handles of this device (recomputed when one is created or destroyed)
    int share_live[kIcpMaxLanes + 1] = {0};  // live handles by share
    struct Lane {
        hipStream_t stream = nullptr;  // stream of the most recent k_icp launch through this lane
        unsigned long stamp = 0;       // (least recently used lane is taken over first)
    } lane[kIcpMaxLanes];
    unsigned long clock = 0;
    hipEvent_t ev = nullptr;  // scratch event (a wait captures the record it was issued behind)
    // co-resident workgroups per dynamic-LDS size of the launch (the occupancy query is per size: a cache keyed on
    // anything coarser would hand a grid sized for one LDS setting to a launch with another)
    struct Entry {
        int lds_bytes = -1, blocks = 0;
    } max_blocks[8];
};
static IcpDeviceGate &icp_gate(int device_id) {
    static IcpDeviceGate gates[64];
    return gates[device_id & 63];
}
static int icp_max_blocks(int device_id, int lds_bytes) {
    IcpDeviceGate &g = icp_gate(device_id);
    std::lock_guard<std::mutex> lk(g.mu);
    int free_slot = 0;
    for (int i = 0; i < 8; ++i) {
        if (g.max_blocks[i].lds_bytes == lds_bytes) return g.max_blocks[i].blocks;
        if (g.max_blocks[i].lds_bytes < 0) {
            free_slot = i;
            break;
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus <= 0) cus = 1;
    int per_cu = icp_blocks_per_cu(lds_bytes);
    if (per_cu < 1) per_cu = 1;
    long b = (long)per_cu * cus;
    g.max_blocks[free_slot].lds_bytes = lds_bytes;
    g.max_blocks[free_slot].blocks = (int)(b < kIcpMaxBlocks ? b : kIcpMaxBlocks);
    return g.max_blocks[free_slot].blocks;
}
// a handle that shares its device with share - 1 others is being created (+1) / destroyed (-1)
static void icp_gate_count_share(int device_id, int share, int delta) {
    IcpDeviceGate &g = icp_gate(device_id);
    std::lock_guard<std::mutex> lk(g.mu);
    share = share < 1 ? 1 : (share > kIcpMaxLanes ? kIcpMaxLanes : share);
    g.share_live[share] += delta;
    if (g.share_live[share] < 0) g.share_live[share] = 0;
    g.lanes = 1;
    for (int k = kIcpMaxLanes; k >= 1; --k)
        if (g.share_live[k] > 0) {
            g.lanes = k;
            break;
        }
}
// launch k_icp on `s` through the device's gate.  share: the part of the co-resident grid the launch was sized for (1 / share
// of it).  A launch with the lanes' own share takes ONE lane; a launch that takes more of the device than a lane stands for
// (a whole-grid registration while pipelines with shares exist: kicp_align_points_to_map, a pipeline created before them)
// takes ALL lanes -- it is ordered behind whatever any lane last launched and everything is ordered behind it: two grids
// that together exceed what the device holds are never in flight at once (round 4's review: the gate did not look at the
// launch's own share, and the lane count was never lowered).
static int icp_launch_ordered(int device_id, IcpParams &P, int grid, bool profile, hipStream_t s, int share, hipEvent_t start = nullptr, hipEvent_t stop = nullptr) {
    IcpDeviceGate &g = icp_gate(device_id);
    std::lock_guard<std::mutex> lk(g.mu);
    if (share < g.lanes) {  // more than a lane's worth of the device
        for (int i = 0; i < kIcpMaxLanes; ++i) {
            if (!g.lane[i].stream || g.lane[i].stream == s) continue;
            bool seen = false;  // (several lanes may hold the same stream: one wait)
            for (int k = 0; k < i; ++k) seen = seen || g.lane[k].stream == g.lane[i].stream;
            if (seen) continue;
            if (!g.ev) KICP_HIP(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
            KICP_HIP(hipEventRecord(g.ev, g.lane[i].stream));
            KICP_HIP(hipStreamWaitEvent(s, g.ev, 0));
        }
        launch_icp(P, grid, profile, P.use_wide != 0, s, start, stop);
        ++g.clock;
        for (int i = 0; i < kIcpMaxLanes; ++i) {
            g.lane[i].stream = i < g.lanes ? s : nullptr;
            g.lane[i].stamp = g.clock;
        }
        return KICP_OK;
    }
    int mine = -1, pick = 0;
    for (int i = 0; i < g.lanes; ++i) {
        if (g.lane[i].stream == s) mine = i;
        if (g.lane[i].stream == nullptr && g.lane[pick].stream != nullptr) pick = i;
        if (g.lane[i].stream != nullptr && g.lane[pick].stream != nullptr && g.lane[i].stamp < g.lane[pick].stamp) pick = i;
    }
    if (mine < 0) {
        mine = pick;
        if (g.lane[mine].stream) {  // behind whatever that lane's last launch (and the work queued after it) is
            if (!g.ev) KICP_HIP(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
            KICP_HIP(hipEventRecord(g.ev, g.lane[mine].stream));
            KICP_HIP(hipStreamWaitEvent(s, g.ev, 0));
        }
    }
    launch_icp(P, grid, profile, P.use_wide != 0, s, start, stop);
    g.lane[mine].stream = s;
    g.lane[mine].stamp = ++g.clock;
    return KICP_OK;
}
// a stream is about to be destroyed: nobody may record on it any more
static void icp_forget_stream(int device_id, hipStream_t s) {
    IcpDeviceGate &g = icp_gate(device_id);
    std::lock_guard<std::mutex> lk(g.mu);
    for (int i = 0; i < kIcpMaxLanes; ++i)
        if (g.lane[i].stream == s) g.lane[i].stream = nullptr;
}

// The grid is launched at the device's co-resident maximum; the kernel itself picks how many of the
// workgroups take part from the actual N_src (surplus workgroups exit at once).  `cap` > 0 lowers the
// maximum (replay after a timeout).
static int icp_fill_policy(int device_id, IcpParams &P, size_t n_hint, int cap, int share = 1) {
    P.force_blocks = (int)options().icp_blocks;
    P.points_per_group = (int)(options().icp_points_per_group > 0 ? options().icp_points_per_group : 1);
    P.use_lds = options().icp_use_lds != 0;
    P.bulk_fill = options().icp_bulk_fill != 0;
    // A workgroup owns its CU: 160 KiB of LDS for the candidate pool, and its eight waves fill the CU's
    // vector register file, so nothing else runs beside it.  The front stages of the NEXT frame run
    // concurrently on a second stream; a few CUs are left out of the grid for them (they are small
    // streaming kernels hidden under the registration either way).
    long lds = options().icp_lds_kib > 0 ? options().icp_lds_kib * 1024 : kIcpLdsBytesMax;
    if (lds > kIcpLdsBytesMax) lds = kIcpLdsBytesMax;
    if (lds < 96 * 1024) lds = 96 * 1024;  // (the fixed part of the layout -- records, table, chunk buffers -- needs ~88 KiB)
    P.lds_bytes = (int)lds;
    int grid = icp_max_blocks(device_id, P.lds_bytes);
    const int reserve = (int)options().icp_reserve_cus;
    if (grid > 4 * reserve) grid -= reserve;
    if (share > 1) grid = grid / share > 0 ? grid / share : 1;  // this handle's part of the device (see the gate)
    if (cap > 0 && cap < grid) grid = cap;
    if (P.force_blocks > grid) P.force_blocks = grid;
    // Which form of the association: a 32-lane group per source point (few points per workgroup, neighbourhoods of
    // hundreds of map points) or a thread per source point (kicp_icp_wide.hpp: hundreds of points per workgroup).  Both
    // give the same pose bit for bit, so going by a HINT of the cloud's size (the previous frame's) is safe.
    const long per_wg = (long)(n_hint / (size_t)(P.force_blocks > 0 ? P.force_blocks : grid));
    P.use_wide = options().icp_wide >= 0 ? (options().icp_wide != 0) : (per_wg > kIcpListRunMax);
    P.wide_prune = (int)options().icp_wide_prune;
    P.schur_solve = (int)options().icp_schur_solve;
    P.wide_prefill = (int)options().icp_wide_prefill;
    P.wide_per_round = (int)options().icp_wide_per_round;
    P.wide_flat = (int)options().icp_wide_flat;
    P.wide_group_max = (int)options().icp_wide_group_max;
    P.wide_stable = (int)options().icp_wide_stable;
    P.group_stable = (int)options().icp_group_stable;
    P.wide_promote_from = (int)options().icp_wide_promote_from;
    P.wide_load_eighths = (int)options().icp_wide_load_eighths;
    return grid;
}

// ---- helper threads for host-side staging copies ------------------------------------------------------
// RegisterFrame takes pageable host memory (std::vector<Eigen::Vector3d>, numpy arrays).  A 130k-point
// scan is 3.1 MB; one core copies that in ~0.3 ms -- longer than the GPU needs for the whole frame.  A few
// helper threads (plus the caller) split the copy into pinned staging memory.
class StagePool {
public:
    // node >= 0: the helpers run on the CPUs of that NUMA node (the GPU's: what they write is read from there)
    explicit StagePool(int helpers, int node = -1) {
        for (int i = 0; i < helpers; ++i)
            threads_.emplace_back([this, node] {
                if (node >= 0 && numa::bind_thread_to_node(pthread_self(), node)) bound_.fetch_add(1, std::memory_order_relaxed);
                started_.fetch_add(1, std::memory_order_release);
                loop();
            });
        while (started_.load(std::memory_order_acquire) < helpers) std::this_thread::yield();
    }
    int helpers() const { return (int)threads_.size(); }
    int bound() const { return bound_.load(std::memory_order_relaxed); }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // fn(chunk) for every chunk in [0, nchunks), on the helpers and the calling thread; returns when all are done
    void run(int nchunks, const std::function<void(int)> &fn) {
        if (nchunks <= 0) return;
        if (nchunks == 1 || threads_.empty()) {
            for (int c = 0; c < nchunks; ++c) fn(c);
            return;
        }
        Job job;
        job.total = nchunks;
        job.fn = &fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            cur_ = &job;
            ++gen_;
            gen_hint_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        work(job);
        while (job.done.load(std::memory_order_acquire) < nchunks) std::this_thread::yield();
        {
            std::lock_guard<std::mutex> lk(mu_);
            cur_ = nullptr;  // no helper can pick the job up any more ...
        }
        while (job.active.load(std::memory_order_acquire) > 0) std::this_thread::yield();  // ... and none still holds it
    }

private:
    struct Job {
        std::atomic<int> next{0}, done{0}, active{0};
        int total = 0;
        const std::function<void(int)> *fn = nullptr;
    };
    static void work(Job &j) {
        for (;;) {
            const int c = j.next.fetch_add(1, std::memory_order_relaxed);
            if (c >= j.total) break;
            (*j.fn)(c);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void loop() {
        unsigned long seen = 0;  // generation of the last job this helper worked on
        for (;;) {
            Job *j;
            // a copy is often two jobs back to back (the narrowing attempt, then the timestamps): look for the next one
            // briefly (~30 us) before going to sleep.  Not longer: the helpers of an idle pipeline must not burn cores
            // (GPU boxes are shared, containers carry CPU quotas), and with "queue_depth" frames queued on the device
            // the caller has a whole frame time for a copy that one core finishes in a third of it.
            for (int spin = 0; spin < 1000 && gen_hint_.load(std::memory_order_acquire) == seen; ++spin) _mm_pause();
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || (cur_ && gen_ != seen); });
                if (quit_) return;
                j = cur_;
                seen = gen_;
                j->active.fetch_add(1, std::memory_order_acq_rel);  // taken under the lock: the job is still alive
            }
            work(*j);
            j->active.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    Job *cur_ = nullptr;
    unsigned long gen_ = 0;
    std::atomic<unsigned long> gen_hint_{0};  // copy of gen_ the helpers may read without the lock
    std::atomic<int> started_{0}, bound_{0};
    bool quit_ = false;
};

// the NUMA node a device hangs off (sysfs, by its PCI address); -1: unknown, or the platform has one node
int device_numa_node(int device) {
    if (options().staging_numa_pretend >= 0) return (int)options().staging_numa_pretend;  // (test hook: see kicp.h)
    static std::mutex mu;
    static int cache[64];
    static bool have[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    const int k = device & 63;
    if (!have[k]) {
        char bdf[64] = {0};
        cache[k] = hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) == hipSuccess ? numa::pci_numa_node(bdf) : -1;
        (void)hipGetLastError();
        have[k] = true;
    }
    return cache[k];
}

// CPUs this process may really use: its affinity mask, capped by the cgroup's CPU quota (containers on GPU boxes: 16 of 256)
static long available_cpus() {
    long n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (long)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) {
                const long c = (quota + period - 1) / period;
                if (c < n || n <= 0) n = c;
            }
        }
        fclose(f);
    }
    return n > 0 ? n : 1;
}
static std::atomic<int> g_live_pipelines{0};  // pipelines of this process (all devices): they share the host's cores

// f64 -> f32 narrowing of a block of coordinates; true iff every value survives the round trip, i.e. the
// data came from a float32 sensor file (datasets/kitti.py:66, ROS PointCloud2) and nothing is lost
static bool narrow_exact_scalar(const double *src, float *dst, size_t count) {
    bool ok = true;
    for (size_t i = 0; i < count; ++i) {
        const float f = (float)src[i];
        dst[i] = f;
        ok &= ((double)f == src[i]);
    }
    return ok;
}
__attribute__((target("avx2"))) static bool narrow_exact_avx2(const double *src, float *dst, size_t count) {
    __m256d all = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));
    size_t i = 0;
    for (; i + 8 <= count; i += 8) {
        const __m256d a = _mm256_loadu_pd(src + i), b = _mm256_loadu_pd(src + i + 4);
        const __m128 fa = _mm256_cvtpd_ps(a), fb = _mm256_cvtpd_ps(b);
        _mm_storeu_ps(dst + i, fa);
        _mm_storeu_ps(dst + i + 4, fb);
        all = _mm256_and_pd(all, _mm256_and_pd(_mm256_cmp_pd(_mm256_cvtps_pd(fa), a, _CMP_EQ_OQ), _mm256_cmp_pd(_mm256_cvtps_pd(fb), b, _CMP_EQ_OQ)));
    }
    bool ok = _mm256_movemask_pd(all) == 0xF;
    return narrow_exact_scalar(src + i, dst + i, count - i) && ok;
}
static bool narrow_exact(const double *src, float *dst, size_t count) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? narrow_exact_avx2(src, dst, count) : narrow_exact_scalar(src, dst, count);
}

}  // namespace kicp

using namespace kicp;

// ==============================================================================================
// kicp_map
// ==============================================================================================
// behind the map counters and the PipeState: the first-level sign-off words of a frame's last kernel (idle = 0: each is
// reset by the workgroup that completes it)
static constexpr size_t kDoneSubOffset = ((sizeof(int) * C_COUNT + sizeof(PipeState) + 127) / 128) * 128;
MapView kicp_map::view() const {
    MapView v;
    v.slots = slots.as<Slot>();
    v.mask = slot_cap - 1;
    v.blocks = blocks.as<char>();
    v.stride = stride;
    v.max_points = (int)max_points;
    v.blocks_cap = blocks_cap;
    v.ctr = ctr.as<int>();
    v.free_ids = free_ids.as<int>();
    v.free_cap = blocks_cap;
    v.doomed = doomed.as<int>();
    v.done_sub = reinterpret_cast<int *>(ctr.as<char>() + kDoneSubOffset);
    v.heads = heads.as<int>();
    v.z_off = kBlockHeader + 16 * (int)max_points;
    v.voxel_size = voxel_size;
    v.max_distance = max_distance;
    v.map_resolution = sqrt(voxel_size * voxel_size / (double)max_points);
    v.dbg = kDebugBounds ? bounds_rec(device) : nullptr;
    return v;
}

int kicp_map::refresh_counters() {
    const double t0 = now_ms();
    // (the wait first, bounded; then the copy: a device-to-host copy into pageable memory BLOCKS inside the runtime until the
    // stream has reached it -- queued in front of the wait it would be the unbounded wait this library does not have)
    KICP_TRY(wait_stream(stream, "map counters"));
    KICP_HIP(hipMemcpy(h_ctr, ctr.p, sizeof h_ctr, hipMemcpyDeviceToHost));
    n_refresh++;
    wait_ms += now_ms() - t0;
    used_ub = h_ctr[C_USED];
    live_ub = h_ctr[C_LIVE];
    bump_ub = h_ctr[C_BUMP] < blocks_cap ? h_ctr[C_BUMP] : blocks_cap;
    return KICP_OK;
}

int kicp_map::scratch_reserve(size_t n_max, InsertScratch &sc) {
    if (n_max == 0) n_max = 1;
    KICP_TRY(world.reserve(n_max * 3 * sizeof(double)));
    KICP_TRY(next.reserve(n_max * sizeof(int)));
    KICP_TRY(rec_slot.reserve(n_max * sizeof(int)));
    KICP_TRY(rec_block.reserve(n_max * sizeof(int)));
    KICP_TRY(rec_list.reserve(n_max * kRecList * sizeof(int)));
    // idle state of a record: count 0, head -1 (every insert leaves its records idle again)
    if (rec_count.bytes < n_max * sizeof(int)) {
        KICP_TRY(rec_count.reserve(n_max * sizeof(int)));
        KICP_HIP(hipMemsetAsync(rec_count.p, 0, rec_count.bytes, stream));
    }
    if (rec_head.bytes < n_max * sizeof(int)) {
        KICP_TRY(rec_head.reserve(n_max * sizeof(int)));
        KICP_HIP(hipMemsetAsync(rec_head.p, 0xFF, rec_head.bytes, stream));
    }
    sc.world = world.as<double>();
    sc.next = next.as<int>();
    sc.rec_slot = rec_slot.as<int>();
    sc.rec_block = rec_block.as<int>();
    sc.rec_count = rec_count.as<int>();
    sc.rec_head = rec_head.as<int>();
    sc.rec_list = rec_list.as<int>();
    sc.parity = (int)(insert_seq++ & 1u);
    return KICP_OK;
}

int kicp_map::check_errors() {
    KICP_TRY(refresh_counters());
    if (h_ctr[C_ERR]) {
        const int bits = h_ctr[C_ERR];
        int zero = 0;
        KICP_HIP(hipMemcpyAsync(ctr.as<int>() + C_ERR, &zero, sizeof zero, hipMemcpyHostToDevice, stream));
        KICP_TRY(wait_stream(stream, "map error word"));
        return err_bits_to_status(bits);
    }
    return KICP_OK;
}

static int map_rehash(kicp_map *m, uint32_t new_cap) {
    // (re)build the slot array from the live blocks; drops tombstones
    m->n_rehash++;
    if (new_cap != m->slot_cap) m->n_grow++;
    if (new_cap != m->slot_cap) {
        DevBuf ns;
        KICP_TRY(ns.reserve((size_t)new_cap * sizeof(Slot)));
        KICP_TRY(wait_stream(m->stream, "map rehash"));
        m->slots.release();
        m->slots = ns;
        m->slot_cap = new_cap;
        KICP_TRY(m->heads.reserve((size_t)new_cap * sizeof(int)));
    }
    KICP_HIP(hipMemsetAsync(m->slots.p, 0xFF, (size_t)m->slot_cap * sizeof(Slot), m->stream));
    KICP_HIP(hipMemsetAsync(m->heads.p, 0xFF, (size_t)m->slot_cap * sizeof(int), m->stream));
    launch_map_rehash(m->view(), m->bump_ub, m->stream);
    KICP_HIP(hipGetLastError());
    KICP_TRY(m->refresh_counters());
    return KICP_OK;
}

bool kicp_map::capacity_ok(size_t incoming) const {
    return 2 * ((size_t)used_ub + incoming) <= slot_cap && (size_t)bump_ub + incoming <= (size_t)blocks_cap;
}

int kicp_map::ensure_capacity(size_t incoming) {
    // Every incoming point may open a new voxel.  Keep the load factor (live + tombstones) <= 1/2
    // and one block per possible new voxel.  The host only knows upper bounds of the device
    // counters between refreshes; refresh (one small D2H) before deciding to grow.
    if (capacity_ok(incoming)) return KICP_OK;
    KICP_TRY(refresh_counters());
    const size_t live = (size_t)h_ctr[C_LIVE], tomb = (size_t)h_ctr[C_TOMB];
    // Growth targets leave room for the frames that may be queued behind the newest one whose exact counters the
    // host has seen ("queue_depth", plus the incoming one and one to spare): each of them is accounted with one
    // new voxel per raw point until its record arrives, and growing is a stream synchronisation.
    const size_t kQueueSlack = (size_t)(options().queue_depth > 0 ? options().queue_depth : 8) + 2;
    if (2 * ((size_t)used_ub + incoming) > slot_cap) {
        const size_t want = 2 * (live + kQueueSlack * incoming);
        uint32_t cap = slot_cap;
        if (want > cap) cap = next_pow2(want * 2);
        if (cap != slot_cap || tomb > 0) KICP_TRY(map_rehash(this, cap));
        if (2 * ((size_t)used_ub + incoming) > slot_cap) {
            set_error("voxel hash cannot grow to %zu slots", want);
            return KICP_ERR_CAPACITY;
        }
    }
    if ((size_t)bump_ub + incoming > (size_t)blocks_cap) {
        size_t want = ((size_t)bump_ub + kQueueSlack * incoming) * 2;
        if (want > (size_t)0x7FFFFFF0) want = (size_t)0x7FFFFFF0;
        if (want < (size_t)bump_ub + incoming) {
            set_error("voxel pool cannot grow beyond %d blocks", blocks_cap);
            return KICP_ERR_CAPACITY;
        }
        const size_t old_bytes = (size_t)blocks_cap * stride;
        n_grow++;
        KICP_TRY(blocks.reserve(want * stride + 64, true, stream));
        KICP_HIP(hipMemsetAsync(blocks.as<char>() + old_bytes, 0, want * stride + 64 - old_bytes, stream));
        // the free-block ring is indexed modulo its capacity: re-linearise the live entries
        // [head, pend) at the front of the larger ring (h_ctr was refreshed above)
        {
            unsigned head = (unsigned)h_ctr[C_FHEAD];
            const unsigned tail = (unsigned)h_ctr[C_FTAIL], pend = (unsigned)h_ctr[C_FPEND];
            if ((int)(tail - head) < 0) head = tail;
            const size_t count = (size_t)(pend - head);
            std::vector<int> old_ring((size_t)blocks_cap), lin(count ? count : 1);
            KICP_TRY(wait_stream(stream, "map growth"));  // (the copy below goes into a local buffer: nothing may be pending when it starts)
            KICP_HIP(hipMemcpy(old_ring.data(), free_ids.p, (size_t)blocks_cap * sizeof(int), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < count; ++i) lin[i] = old_ring[(head + (unsigned)i) % (unsigned)blocks_cap];
            KICP_TRY(free_ids.reserve(want * sizeof(int)));
            KICP_TRY(doomed.reserve(want * sizeof(int)));  // (a list per frame: nothing to keep)
            if (count) KICP_HIP(hipMemcpyAsync(free_ids.p, lin.data(), count * sizeof(int), hipMemcpyHostToDevice, stream));
            const int cur[3] = {0, (int)count, (int)count};
            KICP_HIP(hipMemcpyAsync(ctr.as<int>() + C_FHEAD, &cur[0], sizeof(int), hipMemcpyHostToDevice, stream));
            KICP_HIP(hipMemcpyAsync(ctr.as<int>() + C_FTAIL, &cur[1], sizeof(int), hipMemcpyHostToDevice, stream));
            KICP_HIP(hipMemcpyAsync(ctr.as<int>() + C_FPEND, &cur[2], sizeof(int), hipMemcpyHostToDevice, stream));
            KICP_TRY(wait_stream(stream, "map growth"));
        }
        blocks_cap = (int)want;
    }
    return KICP_OK;
}

static int map_alloc(kicp_map *m) {
    m->stride = ((kBlockHeader + 24 * (int)m->max_points + 127) / 128) * 128;
    m->slot_cap = 1u << 16;
    m->blocks_cap = 1 << 14;
    KICP_TRY(m->slots.reserve((size_t)m->slot_cap * sizeof(Slot)));
    KICP_TRY(m->heads.reserve((size_t)m->slot_cap * sizeof(int)));
    KICP_HIP(hipMemsetAsync(m->heads.p, 0xFF, (size_t)m->slot_cap * sizeof(int), m->stream));
    KICP_TRY(m->blocks.reserve((size_t)m->blocks_cap * m->stride + 64));
    KICP_TRY(m->free_ids.reserve((size_t)m->blocks_cap * sizeof(int)));
    KICP_TRY(m->doomed.reserve((size_t)m->blocks_cap * sizeof(int)));
    KICP_TRY(m->ctr.reserve(kDoneSubOffset + sizeof(int) * kCtrStride * kDoneSub));
    KICP_HIP(hipMemsetAsync(m->slots.p, 0xFF, (size_t)m->slot_cap * sizeof(Slot), m->stream));
    KICP_HIP(hipMemsetAsync(m->blocks.p, 0, m->blocks.bytes, m->stream));
    KICP_HIP(hipMemsetAsync(m->ctr.p, 0, m->ctr.bytes, m->stream));
    KICP_TRY(wait_stream(m->stream, "map creation"));
    m->used_ub = m->bump_ub = m->live_ub = 0;
    return KICP_OK;
}

// the small PipeState that carries a pose into k_map_link for the standalone map calls
static PipeState *map_mini_state(kicp_map *m) {
    return reinterpret_cast<PipeState *>(m->ctr.as<char>() + sizeof(int) * C_COUNT);
}

static int map_create_on_stream(double voxel_size, double max_distance, unsigned max_points,
                                int device_id, hipStream_t stream, kicp_map **out) {
    if (!out) return KICP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!(voxel_size > 0.0) || max_points == 0 || max_points > 4096) {
        set_error("invalid map parameters (voxel_size %g, max_points_per_voxel %u)", voxel_size, max_points);
        return KICP_ERR_INVALID_ARG;
    }
    KICP_TRY(check_device(device_id));
    kicp_map *m = new (std::nothrow) kicp_map();
    if (!m) return KICP_ERR_OOM;
    m->device = device_id;
    m->voxel_size = voxel_size;
    m->max_distance = max_distance;
    m->max_points = max_points;
    if (stream) {
        m->stream = stream;
        m->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete m;
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return KICP_ERR_HIP;
        }
        stream_register(device_id, m->stream);
    }
    int s = map_alloc(m);
    if (s != KICP_OK) {
        kicp_map_destroy(m);
        return s;
    }
    *out = m;
    return KICP_OK;
}

extern "C" {

int kicp_map_create(double voxel_size, double max_distance, unsigned max_points_per_voxel,
                    int device_id, kicp_map **out) {
    return map_create_on_stream(voxel_size, max_distance, max_points_per_voxel, device_id, nullptr, out);
}

int kicp_map_destroy(kicp_map *m) {
    if (!m) return KICP_OK;
    (void)hipSetDevice(m->device);
    // (one bounded wait for everything the map owns; a device that does not answer keeps the memory)
    int idle = wait_stream(m->stream, "map teardown");
    if (idle == KICP_OK) idle = wait_device(m->device, "map teardown");
    for (DevBuf *b : {&m->slots, &m->heads, &m->blocks, &m->free_ids, &m->doomed, &m->ctr, &m->pts_in, &m->world, &m->next, &m->rec_slot, &m->rec_block, &m->rec_count,
                      &m->rec_head, &m->rec_list})
        b->drop(idle == KICP_OK);
    if (m->own_stream && m->stream) {
        if (idle == KICP_OK) (void)stream_destroy(m->stream);  // (else leaked with the rest, and kept in the registry)
    }
    delete m;
    return idle;
}

int kicp_map_clone(const kicp_map *csrc, kicp_map **out) {
    kicp_map *src = const_cast<kicp_map *>(csrc);
    if (!src || !out) return KICP_ERR_INVALID_ARG;
    *out = nullptr;
    KICP_HIP(hipSetDevice(src->device));
    KICP_TRY(wait_stream(src->stream, "map copy (source)"));  // everything queued on the source has happened
    kicp_map *m = nullptr;
    KICP_TRY(map_create_on_stream(src->voxel_size, src->max_distance, src->max_points, src->device, nullptr, &m));
    int s = KICP_OK;
    auto copy = [&](DevBuf &dst, const DevBuf &from) {
        if (s != KICP_OK || !from.p) return;
        dst.release();
        s = dst.reserve(from.bytes);
        if (s == KICP_OK && hipMemcpyAsync(dst.p, from.p, from.bytes, hipMemcpyDeviceToDevice, m->stream) != hipSuccess) {
            set_error("kicp_map_clone: device copy failed");
            s = KICP_ERR_HIP;
        }
    };
    copy(m->slots, src->slots);
    copy(m->heads, src->heads);
    copy(m->blocks, src->blocks);
    copy(m->free_ids, src->free_ids);
    copy(m->doomed, src->doomed);  // (for its size: the list is per frame)
    copy(m->ctr, src->ctr);
    if (s == KICP_OK) s = wait_stream(m->stream, "map copy");
    if (s != KICP_OK) {
        kicp_map_destroy(m);
        return s;
    }
    m->stride = src->stride;
    m->slot_cap = src->slot_cap;
    m->blocks_cap = src->blocks_cap;
    m->used_ub = src->used_ub;
    m->bump_ub = src->bump_ub;
    m->live_ub = src->live_ub;
    memcpy(m->h_ctr, src->h_ctr, sizeof m->h_ctr);
    m->insert_seq = src->insert_seq;
    *out = m;
    return KICP_OK;
}

int kicp_map_clear(kicp_map *m) {
    if (!m) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(m->refresh_counters());
    KICP_HIP(hipMemsetAsync(m->slots.p, 0xFF, (size_t)m->slot_cap * sizeof(Slot), m->stream));
    KICP_HIP(hipMemsetAsync(m->heads.p, 0xFF, (size_t)m->slot_cap * sizeof(int), m->stream));
    KICP_HIP(hipMemsetAsync(m->blocks.p, 0, (size_t)m->bump_ub * m->stride, m->stream));
    KICP_HIP(hipMemsetAsync(m->ctr.p, 0, sizeof(int) * C_COUNT, m->stream));
    KICP_TRY(wait_stream(m->stream, "map clear"));
    m->used_ub = m->bump_ub = m->live_ub = 0;
    return KICP_OK;
}

int kicp_map_empty(const kicp_map *cm, int *empty) {
    kicp_map *m = const_cast<kicp_map *>(cm);
    if (!m || !empty) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(m->refresh_counters());
    *empty = (m->h_ctr[C_LIVE] == 0);
    return KICP_OK;
}

int kicp_map_size(const kicp_map *cm, size_t *n_voxels, size_t *n_points) {
    kicp_map *m = const_cast<kicp_map *>(cm);
    if (!m) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(m->refresh_counters());
    if (n_points) {
        KICP_HIP(hipMemsetAsync(m->ctr.as<int>() + C_NPTS, 0, sizeof(int), m->stream));
        launch_map_count_points(m->view(), m->bump_ub, m->stream);
        KICP_HIP(hipGetLastError());
        KICP_TRY(m->refresh_counters());
        *n_points = (size_t)m->h_ctr[C_NPTS];
    }
    if (n_voxels) *n_voxels = (size_t)m->h_ctr[C_LIVE];
    return KICP_OK;
}

// AddPoints on points already in HBM.  origin / origin_state (one of them, or neither): RemovePointsFarFromLocation around that
// point rides along -- the fused update of kicp_map.hip ("map_fused_update"; an insert of no points has nothing to ride on)
static int map_insert_device(kicp_map *m, const double *d_in, const int *n_ptr, int n_imm,
                             size_t n_max, const PipeState *state, int use_pose, const double *origin = nullptr,
                             const PipeState *origin_state = nullptr) {
    if (n_max == 0) return KICP_OK;
    KICP_TRY(m->ensure_capacity(n_max));
    InsertScratch sc;
    KICP_TRY(m->scratch_reserve(n_max, sc));
    const MapView v = m->view();
    MapPrune mp;
    memset(&mp, 0, sizeof mp);
    const bool fused = origin || origin_state;
    if (fused) {
        mp.bump_ub = m->bump_ub;
        mp.state = origin_state;
        mp.use_state_origin = origin_state ? 1 : 0;
        for (int k = 0; k < 3; ++k) mp.origin[k] = origin ? origin[k] : 0.0;
    }
    launch_map_link(v, sc, d_in, n_ptr, n_imm, (int)n_max, state, use_pose, m->stream, fused ? &mp : nullptr);
    launch_map_apply(v, sc, (int)n_max, m->stream, fused ? &mp : nullptr);
    KICP_HIP(hipGetLastError());
    m->used_ub += (long)n_max;
    m->live_ub += (long)n_max;
    m->bump_ub += (long)n_max;
    if (m->bump_ub > m->blocks_cap) m->bump_ub = m->blocks_cap;
    return KICP_OK;
}

static int map_upload(kicp_map *m, const double *xyz, size_t n) {
    if (n > (size_t)0x7FFFFFF0 / 3) {
        set_error("too many points (%zu)", n);
        return KICP_ERR_INVALID_ARG;
    }
    KICP_TRY(m->pts_in.reserve((n ? n : 1) * 3 * sizeof(double)));
    inject_stall(m->stream);
    if (n) KICP_HIP(hipMemcpyAsync(m->pts_in.p, xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, m->stream));
    return KICP_OK;
}

int kicp_map_add_points(kicp_map *m, const double *xyz, size_t n) {
    if (!m || (!xyz && n)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(map_upload(m, xyz, n));
    KICP_TRY(map_insert_device(m, m->pts_in.as<double>(), nullptr, (int)n, n, nullptr, 0));
    return m->check_errors();
}

int kicp_map_remove_far(kicp_map *m, const double origin[3]) {
    if (!m || !origin) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    inject_stall(m->stream);
    launch_map_prune(m->view(), m->bump_ub, nullptr, 0, origin, nullptr, 0, m->stream);
    KICP_HIP(hipGetLastError());
    return m->check_errors();
}

int kicp_map_update_origin(kicp_map *m, const double *xyz, size_t n, const double origin[3]) {
    if (!m || (!xyz && n) || !origin) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(map_upload(m, xyz, n));
    const bool fused = options().map_fused_update != 0 && n > 0;
    KICP_TRY(map_insert_device(m, m->pts_in.as<double>(), nullptr, (int)n, n, nullptr, 0, fused ? origin : nullptr));
    if (!fused) launch_map_prune(m->view(), m->bump_ub, nullptr, 0, origin, nullptr, 0, m->stream);
    KICP_HIP(hipGetLastError());
    return m->check_errors();
}

int kicp_map_update_pose(kicp_map *m, const double *xyz, size_t n, const double pose[16]) {
    if (!m || (!xyz && n) || !pose) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    SE3 T;
    if (!se3_from_matrix(pose, T)) {
        set_error("pose is not a rigid transform (SOPHUS_ENSURE: R not orthogonal or det <= 0)");
        return KICP_ERR_INVALID_ARG;
    }
    PipeState *ms = map_mini_state(m);
    KICP_HIP(hipMemcpyAsync(&ms->new_pose, &T, sizeof T, hipMemcpyHostToDevice, m->stream));
    KICP_TRY(map_upload(m, xyz, n));
    const bool fused = options().map_fused_update != 0 && n > 0;
    KICP_TRY(map_insert_device(m, m->pts_in.as<double>(), nullptr, (int)n, n, ms, 1, nullptr, fused ? ms : nullptr));
    if (!fused) launch_map_prune(m->view(), m->bump_ub, ms, 1, nullptr, nullptr, 0, m->stream);
    KICP_HIP(hipGetLastError());
    return m->check_errors();
}

int kicp_map_pointcloud(const kicp_map *cm, double *out_xyz, size_t cap, size_t *n_out) {
    kicp_map *m = const_cast<kicp_map *>(cm);
    if (!m || !n_out || (!out_xyz && cap)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(m->refresh_counters());
    const size_t nb = (size_t)m->bump_ub;
    std::vector<char> host(nb * m->stride);
    if (nb) KICP_HIP(hipMemcpy(host.data(), m->blocks.p, nb * m->stride, hipMemcpyDeviceToHost));  // (the stream is idle: refresh_counters waited)
    size_t k = 0;
    for (size_t b = 0; b < nb; ++b) {
        const BlockHdr *h = reinterpret_cast<const BlockHdr *>(host.data() + b * m->stride);
        const double *xy = reinterpret_cast<const double *>(host.data() + b * m->stride + kBlockHeader);
        const double *z = reinterpret_cast<const double *>(host.data() + b * m->stride + kBlockHeader + 16 * m->max_points);
        for (int i = 0; i < h->count; ++i, ++k)
            if (k < cap) {
                out_xyz[3 * k] = xy[2 * i];
                out_xyz[3 * k + 1] = xy[2 * i + 1];
                out_xyz[3 * k + 2] = z[i];
            }
    }
    *n_out = k;
    return KICP_OK;
}

int kicp_map_closest_neighbor(const kicp_map *cm, const double *q, size_t nq, double *nn, double *dist) {
    kicp_map *m = const_cast<kicp_map *>(cm);
    if (!m || ((!q || !nn || !dist) && nq)) return KICP_ERR_INVALID_ARG;
    if (nq == 0) return KICP_OK;
    KICP_HIP(hipSetDevice(m->device));
    KICP_TRY(map_upload(m, q, nq));
    KICP_TRY(m->world.reserve(nq * 4 * sizeof(double)));
    double *d_nn = m->world.as<double>();
    double *d_dist = d_nn + 3 * nq;
    launch_closest_neighbor(m->view(), m->pts_in.as<double>(), (int)nq, d_nn, d_dist, m->stream);
    KICP_HIP(hipGetLastError());
    // (the caller's buffers are only written once the kernel is known to have ended: a wait that gives up leaves no copy pending)
    KICP_TRY(wait_stream(m->stream, "closest-neighbour search"));
    KICP_HIP(hipMemcpy(nn, d_nn, nq * 3 * sizeof(double), hipMemcpyDeviceToHost));
    KICP_HIP(hipMemcpy(dist, d_dist, nq * sizeof(double), hipMemcpyDeviceToHost));
    return KICP_OK;
}

}  // extern "C"

// ==============================================================================================
// Registration
// ==============================================================================================
static PrepState idle_prep_state() {
    PrepState h;
    memset(&h, 0, sizeof h);
    h.tmin_bits = ~0ull;
    h.tmax_bits = 0ull;
    return h;
}

static void init_state(PipeState &s, double initial_threshold) {
    memset(&s, 0, sizeof s);
    s.last_pose = se3_identity();
    s.last_delta = se3_identity();
    s.new_pose = se3_identity();
    s.guess = se3_identity();
    s.model_sse = initial_threshold * initial_threshold;  // Threshold.cpp:35
    s.num_samples = 1;
    s.epoch_base = 1;
}

static constexpr unsigned kSpinLimit = 1u << 21;

extern "C" {

int kicp_registration_create(int max_num_iterations, double convergence_criterion, int max_num_threads,
                             int device_id, kicp_registration **out) {
    (void)max_num_threads;
    if (!out) return KICP_ERR_INVALID_ARG;
    *out = nullptr;
    KICP_TRY(check_device(device_id));
    kicp_registration *r = new (std::nothrow) kicp_registration();
    if (!r) return KICP_ERR_OOM;
    r->device = device_id;
    r->max_iters = max_num_iterations;
    r->conv = convergence_criterion;
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&r->ev0) != hipSuccess || hipEventCreate(&r->ev1) != hipSuccess) {
        set_error("stream/event creation failed");
        kicp_registration_destroy(r);
        return KICP_ERR_HIP;
    }
    stream_register(device_id, r->stream);
    int s = r->state.reserve(sizeof(PipeState));
    if (s == KICP_OK && (icp_prepare(device_id) != 0 || tile_sort_prepare(device_id) != 0)) {
        set_error("hipFuncSetAttribute(k_icp, 160 KiB LDS) failed");
        s = KICP_ERR_HIP;
    }
    if (s == KICP_OK) s = r->granules.reserve(icp_granule_words(kIcpMaxBlocks) * sizeof(unsigned long long));
    if (s != KICP_OK) {
        kicp_registration_destroy(r);
        return s;
    }
    PipeState st;
    init_state(st, 0.0);
    KICP_HIP(hipMemcpyAsync(r->state.p, &st, sizeof st, hipMemcpyHostToDevice, r->stream));
    KICP_HIP(hipMemsetAsync(r->granules.p, 0, r->granules.bytes, r->stream));
    {
        const int ws = wait_stream(r->stream, "registration creation");
        if (ws != KICP_OK) {
            kicp_registration_destroy(r);
            return ws;
        }
    }
    *out = r;
    return KICP_OK;
}

int kicp_registration_destroy(kicp_registration *r) {
    if (!r) return KICP_OK;
    (void)hipSetDevice(r->device);
    int idle = r->stream ? wait_stream(r->stream, "registration teardown") : KICP_OK;
    if (idle == KICP_OK) idle = wait_device(r->device, "registration teardown");
    for (DevBuf *b : {&r->frame, &r->work, &r->sort_in, &r->sort_out, &r->sort_tmp, &r->run_wts, &r->run_wts32, &r->granules, &r->state}) b->drop(idle == KICP_OK);
    if (idle == KICP_OK) {
        if (r->ev0) (void)hipEventDestroy(r->ev0);
        if (r->ev1) (void)hipEventDestroy(r->ev1);
    }
    if (r->stream && idle == KICP_OK) {
        // (a stream that is leaked with its k_icp possibly in flight keeps its lane of the device's launch gate: later
        // registrations are ordered behind it instead of being launched beside it)
        icp_forget_stream(r->device, r->stream);
        (void)stream_destroy(r->stream);
    }
    delete r;
    return idle;
}

int kicp_align_points_to_map(kicp_registration *r, const double *frame_xyz, size_t n,
                             const kicp_map *cmap, const double initial_guess[16],
                             double max_correspondence_distance, double kernel_scale,
                             double T_out[16], kicp_icp_stats *stats) {
    kicp_map *map = const_cast<kicp_map *>(cmap);
    if (!r || !map || !initial_guess || !T_out || (!frame_xyz && n)) return KICP_ERR_INVALID_ARG;
    if (map->device != r->device) {
        set_error("map lives on device %d, registration on device %d", map->device, r->device);
        return KICP_ERR_INVALID_ARG;
    }
    if (n > (size_t)0x7FFFFFF0 / 3) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(r->device));
    SE3 guess;
    if (!se3_from_matrix(initial_guess, guess)) {
        set_error("initial_guess is not a rigid transform (SOPHUS_ENSURE)");
        return KICP_ERR_INVALID_ARG;
    }
    KICP_TRY(wait_stream(map->stream, "AlignPointsToMap (the map's pending work)"));  // the map's own work is done before we read it
    KICP_TRY(r->frame.reserve((n ? n : 1) * 3 * sizeof(double)));
    inject_stall(r->stream);
    KICP_TRY(r->work.reserve((n ? n : 1) * 3 * sizeof(double)));
    if (n) KICP_HIP(hipMemcpyAsync(r->frame.p, frame_xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, r->stream));
    PipeState *st = r->state.as<PipeState>();
    KICP_HIP(hipMemcpyAsync(&st->guess, &guess, sizeof guess, hipMemcpyHostToDevice, r->stream));
    // spatial order of the source cloud (see kicp_sort.hip)
    const bool sorted = n > 0 && n <= ((size_t)1 << 24);
    if (sorted) {
        KICP_TRY(r->sort_in.reserve(n * sizeof(unsigned long long)));
        KICP_TRY(r->sort_out.reserve(n * sizeof(unsigned long long)));
        const size_t tb = tile_sort_temp_bytes(n);
        KICP_TRY(r->sort_tmp.reserve(tb));
        const int se = launch_tile_sort(r->frame.as<double>(), nullptr, (int)n, n, map->voxel_size, r->sort_in.as<unsigned long long>(),
                                        r->sort_out.as<unsigned long long>(), n, r->stream);
        if (se != 0) {
            set_error("device sort of the source cloud failed (%s)", hipGetErrorString((hipError_t)se));
            return KICP_ERR_HIP;
        }
        if (r->run_wts.bytes < n * sizeof(unsigned long long)) {  // tagged granules: a new buffer starts with tags no launch uses
            KICP_TRY(r->run_wts.reserve(n * sizeof(unsigned long long)));
            KICP_HIP(hipMemsetAsync(r->run_wts.p, 0, r->run_wts.bytes, r->stream));
        }
        KICP_TRY(r->run_wts32.reserve((size_t)kIcpListRunMax * kIcpMaxBlocks * sizeof(unsigned)));  // (k_icp_weights: short runs only)
    }
    PipeState h;
    for (int attempt = 0, cap = 0;; ++attempt) {
        IcpParams P;
        memset(&P, 0, sizeof P);
        const int G = icp_fill_policy(r->device, P, n, cap);
        P.frame = r->frame.as<double>();
        P.order = sorted ? r->sort_out.as<unsigned long long>() : nullptr;
        P.wts = sorted ? r->run_wts.as<unsigned long long>() : nullptr;
        P.wts32 = sorted && options().icp_weights_kernel != 0 ? r->run_wts32.as<unsigned>() : nullptr;
        P.weight_base = (int)options().icp_weight_base;
        P.weight_quad = (int)options().icp_weight_quad;
        P.weight_long_base = (int)options().icp_weight_long_base;
        P.weight_long_emul = (int)options().icp_weight_long_emul;
        P.weight_dense_min = (int)options().icp_weight_dense_min;
        P.weight_dense_div = (int)options().icp_weight_dense_div;
        P.work = r->work.as<double>();
        P.n_ptr = nullptr;
        P.n_imm = (int)n;
        P.map = map->view();
        P.state = st;
        P.pipeline_mode = 0;
        P.max_dist = max_correspondence_distance;
        P.kernel_scale = kernel_scale;
        P.max_iters = r->max_iters;
        P.conv = r->conv;
        P.granules = r->granules.as<unsigned long long>();
        P.spin_limit = kSpinLimit;
        if (P.wts32) launch_icp_weights(P, G, n, r->stream);
        KICP_HIP(hipEventRecord(r->ev0, r->stream));
        KICP_TRY(icp_launch_ordered(r->device, P, G, options().icp_profile != 0, r->stream, 1));
        KICP_HIP(hipGetLastError());
        KICP_HIP(hipEventRecord(r->ev1, r->stream));
        // (the result is fetched only once the launch is known to have ended: `h` is a local, and a wait that gives up must
        // leave no copy pending into it)
        KICP_TRY(wait_stream(r->stream, "AlignPointsToMap"));
        KICP_HIP(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
        if (!h.err) break;
        int zero = 0;
        KICP_HIP(hipMemcpy(&st->err, &zero, sizeof zero, hipMemcpyHostToDevice));
        // the workgroups were not all resident (something else holds CUs of this device): the launch
        // committed nothing, so it is simply repeated with half as many workgroups
        if ((h.err & E_TIMEOUT) && attempt < 3 && G > 1) {
            cap = G / 2;
            continue;
        }
        return err_bits_to_status(h.err);
    }
    se3_matrix(h.new_pose, T_out);
    memcpy(r->last_sums, h.icp_last_sums, sizeof r->last_sums);
    if (stats) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r->ev0, r->ev1);
        stats->iterations = h.icp_iterations;
        stats->converged = h.icp_converged;
        stats->n_source = n;
        stats->n_corr_last = h.icp_ncorr_last;
        stats->points_examined = h.icp_examined;
        stats->n_corr_total = h.icp_ncorr_total;
        stats->kernel_ms = ms;
    }
    return KICP_OK;
}

int kicp_registration_last_system(const kicp_registration *r, double JTJ[36], double JTr[6], uint64_t *n_corr) {
    if (!r || !JTJ || !JTr) return KICP_ERR_INVALID_ARG;
    // the 16 unique sums -> the 6x6 / 6x1 of Registration.cpp:80-121 (J = [I | -hat(s)], row-major)
    const double *S = r->last_sums;
    for (int i = 0; i < 36; ++i) JTJ[i] = 0.0;
    JTJ[0] = JTJ[7] = JTJ[14] = S[0];
    JTJ[0 * 6 + 4] = JTJ[4 * 6 + 0] = S[3];
    JTJ[0 * 6 + 5] = JTJ[5 * 6 + 0] = -S[2];
    JTJ[1 * 6 + 3] = JTJ[3 * 6 + 1] = -S[3];
    JTJ[1 * 6 + 5] = JTJ[5 * 6 + 1] = S[1];
    JTJ[2 * 6 + 3] = JTJ[3 * 6 + 2] = S[2];
    JTJ[2 * 6 + 4] = JTJ[4 * 6 + 2] = -S[1];
    JTJ[3 * 6 + 3] = S[4];
    JTJ[3 * 6 + 4] = JTJ[4 * 6 + 3] = S[5];
    JTJ[3 * 6 + 5] = JTJ[5 * 6 + 3] = S[6];
    JTJ[4 * 6 + 4] = S[7];
    JTJ[4 * 6 + 5] = JTJ[5 * 6 + 4] = S[8];
    JTJ[5 * 6 + 5] = S[9];
    for (int i = 0; i < 6; ++i) JTr[i] = S[10 + i];
    if (n_corr) *n_corr = (uint64_t)S[16];
    return KICP_OK;
}

}  // extern "C"

// ==============================================================================================
// VoxelDownsample / Preprocess as free functions (temporary HBM buffers per call)
// ==============================================================================================
struct ScopedStream {
    hipStream_t s = nullptr;
    int create(int device_id) {
        KICP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        stream_register(device_id, s);
        return KICP_OK;
    }
    ~ScopedStream() {
        if (s && hipStreamQuery(s) == hipSuccess) (void)stream_destroy(s);  // (still busy -- the call gave up on it: leaked, and kept in the registry)
        (void)hipGetLastError();
    }
};
struct ScopedBufs {  // (declared AFTER the call's ScopedStream: destroyed first, and its one wait covers the stream's teardown too)
    std::vector<DevBuf *> v;
    ~ScopedBufs() {
        // (a call that has just given up on the device does not wait for it a second time: its buffers are leaked)
        const bool idle = !t_gave_up && wait_device(current_device(), "temporary buffers") == KICP_OK;
        for (auto *b : v) b->drop(idle);
    }
};

static int downsample_device(const double *d_in, const int *n_ptr, int n_imm, int n_max, double voxel,
                             DevBuf &tab, uint32_t tab_cap, int *slot_of, int *blk_counts, double *d_out,
                             int *d_nout, int *d_err, bool claim, int order, int *rb_elem, int *rb_home, hipStream_t s) {
    DsParams P;
    memset(&P, 0, sizeof P);
    P.order = order;
    P.tab_cap = (int)tab_cap;
    P.rb_elem = rb_elem;
    P.rb_home = rb_home;
    P.in = d_in;
    P.n_ptr = n_ptr;
    P.n_imm = n_imm;
    P.n_max = n_max;
    P.voxel = voxel;
    P.tab = tab.as<DsSlot>();
    P.mask = tab_cap - 1;
    P.slot_of = slot_of;
    P.blk_counts = blk_counts;
    P.out = d_out;
    P.n_out = d_nout;
    P.err = d_err;
    P.dbg = kDebugBounds ? bounds_rec(current_device()) : nullptr;
    if (claim) launch_ds_claim(P, s);
    if (order) {
        launch_ds_arrange(P, s);
        launch_ds_scatter_rb(P, s);
    } else {
        launch_ds_flags(P, s);
        launch_ds_scatter(P, s);
    }
    KICP_HIP(hipGetLastError());
    return KICP_OK;
}

static int init_ds_table(DevBuf &tab, uint32_t cap, hipStream_t s) {
    // key = EMPTY (all ones), minidx = INT_MAX: fill with 0xFF then fix minidx by a pattern fill
    KICP_TRY(tab.reserve((size_t)cap * sizeof(DsSlot)));
    std::vector<DsSlot> h(cap);
    for (auto &e : h) {
        e.key = kKeyEmpty;
        e.minidx = 0x7FFFFFFF;
        e.pad = 0;
    }
    KICP_HIP(hipMemcpyAsync(tab.p, h.data(), (size_t)cap * sizeof(DsSlot), hipMemcpyHostToDevice, s));
    return wait_stream(s, "downsample table initialisation");
}

extern "C" {

int kicp_voxel_downsample(const double *xyz, size_t n, double voxel_size, int device_id, double *out_xyz,
                          size_t *n_out) {
    if ((!xyz || !out_xyz) && n) return KICP_ERR_INVALID_ARG;
    if (!n_out || !(voxel_size > 0.0) || n > (size_t)0x7FFFFFF0 / 3) return KICP_ERR_INVALID_ARG;
    KICP_TRY(check_device(device_id));
    *n_out = 0;
    if (n == 0) return KICP_OK;
    ScopedStream ss;
    KICP_TRY(ss.create(device_id));
    DevBuf in, out, tab, slot_of, counts, misc, rb;
    ScopedBufs sb;
    sb.v = {&in, &out, &tab, &slot_of, &counts, &misc, &rb};
    const int order = options().downsample_order != 0;
    const uint32_t cap = next_pow2(2 * n);
    KICP_TRY(in.reserve(n * 3 * sizeof(double)));
    KICP_TRY(out.reserve(n * 3 * sizeof(double)));
    KICP_TRY(slot_of.reserve(n * sizeof(int)));
    KICP_TRY(counts.reserve(((size_t)cap / 1024 + (n + 1023) / 1024 + 2) * sizeof(int)));
    KICP_TRY(misc.reserve(2 * sizeof(int)));
    if (order) KICP_TRY(rb.reserve((size_t)cap * 2 * sizeof(int)));
    KICP_TRY(init_ds_table(tab, cap, ss.s));
    inject_stall(ss.s);
    KICP_HIP(hipMemsetAsync(misc.p, 0, 2 * sizeof(int), ss.s));
    KICP_HIP(hipMemcpyAsync(in.p, xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ss.s));
    KICP_TRY(downsample_device(in.as<double>(), nullptr, (int)n, (int)n, voxel_size, tab, cap, slot_of.as<int>(),
                               counts.as<int>(), out.as<double>(), misc.as<int>(), misc.as<int>() + 1, true, order,
                               order ? rb.as<int>() : nullptr, order ? rb.as<int>() + cap : nullptr, ss.s));
    int h[2];
    KICP_TRY(wait_stream(ss.s, "VoxelDownsample"));  // (then the results: nothing is left pending into locals or the caller's buffers)
    KICP_HIP(hipMemcpy(h, misc.p, sizeof h, hipMemcpyDeviceToHost));
    if (h[1]) return err_bits_to_status(h[1]);
    KICP_HIP(hipMemcpy(out_xyz, out.p, (size_t)h[0] * 3 * sizeof(double), hipMemcpyDeviceToHost));
    *n_out = (size_t)h[0];
    return KICP_OK;
}

int kicp_preprocess(const double *xyz, size_t n, const double *timestamps, size_t n_ts,
                    const double relative_motion[16], double max_range, double min_range, int deskew,
                    int device_id, double *out_xyz, size_t *n_out) {
    if ((!xyz || !out_xyz) && n) return KICP_ERR_INVALID_ARG;
    if (!n_out || !relative_motion || n > (size_t)0x7FFFFFF0 / 3) return KICP_ERR_INVALID_ARG;
    KICP_TRY(check_device(device_id));
    *n_out = 0;
    SE3 motion;
    if (!se3_from_matrix(relative_motion, motion)) {
        set_error("relative_motion is not a rigid transform (SOPHUS_ENSURE)");
        return KICP_ERR_INVALID_ARG;
    }
    const bool do_deskew = deskew && n_ts > 0 && timestamps;
    if (do_deskew && n_ts < n) {
        set_error("timestamps (%zu) shorter than frame (%zu)", n_ts, n);
        return KICP_ERR_TIMESTAMPS;
    }
    if (n == 0) return KICP_OK;
    ScopedStream ss;
    KICP_TRY(ss.create(device_id));
    DevBuf in, ts, tmp, out, counts, st, prep;
    ScopedBufs sb;
    sb.v = {&in, &ts, &tmp, &out, &counts, &st, &prep};
    KICP_TRY(in.reserve(n * 3 * sizeof(double)));
    KICP_TRY(tmp.reserve(n * 3 * sizeof(double)));
    KICP_TRY(out.reserve(n * 3 * sizeof(double)));
    KICP_TRY(counts.reserve(((n + 1023) / 1024 + 1) * sizeof(int)));
    KICP_TRY(st.reserve(sizeof(PipeState)));
    PipeState hs;
    init_state(hs, 0.0);
    inject_stall(ss.s);
    KICP_HIP(hipMemcpyAsync(st.p, &hs, sizeof hs, hipMemcpyHostToDevice, ss.s));
    KICP_TRY(prep.reserve(sizeof(PrepState)));
    PrepState hp = idle_prep_state();
    KICP_HIP(hipMemcpyAsync(prep.p, &hp, sizeof hp, hipMemcpyHostToDevice, ss.s));
    KICP_HIP(hipMemcpyAsync(in.p, xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ss.s));
    if (do_deskew) {
        KICP_TRY(ts.reserve(n_ts * sizeof(double)));
        KICP_HIP(hipMemcpyAsync(ts.p, timestamps, n_ts * sizeof(double), hipMemcpyHostToDevice, ss.s));
        launch_ts_minmax(ts.as<double>(), (int)n_ts, prep.as<PrepState>(), ss.s);
    }
    PreParams P;
    memset(&P, 0, sizeof P);
    P.xyz = in.as<double>();
    P.ts = do_deskew ? ts.as<double>() : nullptr;
    P.n = (int)n;
    P.deskew = do_deskew ? 1 : 0;
    P.use_state_motion = 0;
    P.motion = motion;
    P.state = st.as<PipeState>();
    P.prep = prep.as<PrepState>();
    P.max_range = max_range;
    P.min_range = min_range;
    P.tmp = tmp.as<double>();
    P.blk_counts = counts.as<int>();
    P.out = out.as<double>();
    P.n_out = &prep.as<PrepState>()->n_pre;
    P.ds_tab = nullptr;
    P.err = &st.as<PipeState>()->err;
    P.dbg = kDebugBounds ? bounds_rec(device_id) : nullptr;
    launch_pre_flags(P, ss.s);
    launch_pre_scatter(P, ss.s);
    KICP_HIP(hipGetLastError());
    KICP_TRY(wait_stream(ss.s, "Preprocess"));
    KICP_HIP(hipMemcpy(&hs, st.p, sizeof hs, hipMemcpyDeviceToHost));
    KICP_HIP(hipMemcpy(&hp, prep.p, sizeof hp, hipMemcpyDeviceToHost));
    if (hs.err) return err_bits_to_status(hs.err);
    KICP_HIP(hipMemcpy(out_xyz, out.p, (size_t)hp.n_pre * 3 * sizeof(double), hipMemcpyDeviceToHost));
    *n_out = (size_t)hp.n_pre;
    return KICP_OK;
}

}  // extern "C"

// ==============================================================================================
// pipeline::KissICP
// ==============================================================================================
struct FrameRecord {  // the first kRecWords words mirror the device layout [map counters | PipeState]
    int map_ctr[C_COUNT];
    PipeState st;
    uint64_t n_raw;
};
static_assert(offsetof(FrameRecord, st) == sizeof(int) * C_COUNT, "PipeState sits right behind the map counters");
constexpr int kRecWords = (int)((sizeof(int) * C_COUNT + sizeof(PipeState)) / sizeof(unsigned));

// what was enqueued last (enough to run the frame again after a registration that gave up)
struct FrameInput {
    bool valid = false;
    bool host_mapped = false;  // d_xyz / d_ts are device addresses of pinned HOST memory (the zero-copy staging slot)
    const void *d_xyz = nullptr;
    int xyz_f32 = 0;
    size_t n = 0;
    const double *d_ts = nullptr;
    size_t n_ts = 0;
};

// Two streams per pipeline.  `stream` carries the frame's serial chain: AlignPointsToMap, then the
// map update (frame k+1's registration needs frame k's points in the map).  `prep_stream` carries
// the upload of the scan and the stages in front of the registration (Preprocess + Voxelize): for
// frame k+1 they only need frame k's pose (for the deskew motion), so they run while frame k's
// registration and map update are still in flight.
struct kicp_pipeline {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t prep_stream = nullptr;
    hipStream_t copy_stream = nullptr;     // early download of the preprocessed frame (kicp_pipeline_register_frame_outputs), created on first use
    hipEvent_t icp_done_event = nullptr;   // the event recorded behind the most recent ICP launch (or none)
    hipEvent_t ev_prep_done[2] = {nullptr, nullptr};  // recorded on `prep_stream`, by frame parity
    uint64_t frames_enqueued = 0;
    kicp_config cfg;
    kicp_map *map = nullptr;
    // fd (the 0.5 v cloud, read by the map update) and src (the 1.5 v cloud, read by the registration)
    // exist twice, indexed by frame parity; so do the upload targets raw / ts of the host-input path
    DevBuf raw[2], ts[2], tmp, pre, fd[2], src[2], work, slot1, slot2, tab1, tab2, rb1, rb2, counts, granules, prof_groups, prep;
    int ds_order = 1;  // VoxelDownsample output order ("downsample_order" option, read at create)
    DevBuf sort_in, sort_out[2], sort_tmp;  // spatial order of the source cloud (keys; sorted keys by frame parity; rocPRIM scratch)
    DevBuf run_wts;                         // run weights of the sorted points, one tagged granule each (written by k_icp's prologue)
    DevBuf run_wts32;                       // ... as plain 32-bit words, written by k_icp_weights in front of the launch (short runs)
    size_t sort_tmp_bytes = 0;
    size_t cap_points = 0;
    uint32_t tab_cap = 0;
    // per-frame records land in pinned host memory, one slot per frame in flight
    static constexpr int kRing = 256;
    FrameRecord *ring = nullptr;  // hipHostMalloc
    hipEvent_t ev[kRing][2];
    // Option "frame_events" (off by default): the three per-frame events attached to dispatches (hipExtLaunchKernel) instead
    // of being packets of their own in the queue.  The idea: the kernel timeline shows 0 us between k_map_link / apply /
    // prune, which have nothing between them, 7 us behind k_icp with one recorded event, 13 - 16 in front of it with a wait
    // and a record (profiles/r04_final_timeline.txt).  The measurement: a dispatch that carries a completion signal costs
    // more than the packet it saves -- 0.387 against 0.360 ms per frame (profiles/r04_av_frame_events_sweep.txt).  ev[i][0] /
    // [1] are then the start / stop of frame i's k_icp dispatch (timing; [1] also "the pose is ready"), ev_done[i] the
    // completion of its k_map_prune: "frame i is completely done" -- what ev[i + 1][0] says when events are recorded.
    hipEvent_t ev_done[kRing];
    bool ext_events = false;
    bool ev_ok = false;
    int in_flight = 0;
    int done_upto = 0;    // frames [0, done_upto) of the ring are known to be complete (their successor's launch event fired)
    uint64_t frames_done = 0;
    FrameRecord last;  // most recent completed frame
    bool have_last = false;
    // host-input path: pinned staging slots, reused round-robin once their upload has completed
    static constexpr int kStage = 4;
    char *stage[kStage] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_points = 0;
    hipEvent_t ev_h2d[kStage] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_busy[kStage] = {false, false, false, false};
    uint64_t staged = 0;
    int f32_skip = 0;  // frames for which the lossless-narrowing attempt is skipped (the last attempt failed)
    StagePool *pool = nullptr;
    size_t stage_bytes = 0;
    bool stage_registered = false;  // the slots are node-bound pages of our own, registered with the runtime (else: hipHostMalloc)
    int stage_node = -1;            // NUMA node the slots lie on (-1: unknown)
    char *out_stage = nullptr;  // pinned bounce buffer of kicp_pipeline_output
    size_t out_stage_bytes = 0;
    // registration replay (timeout): co-residency cap for the ICP grid, the last frame's inputs
    int icp_cap = 0;
    int icp_share = 1;  // streams this device's persistent grid is divided among (option "icp_device_streams" when the pipeline was created)
    bool share_counted = false;  // ... and the device's gate knows about it
    int inject_timeouts = 0;  // test hook ("icp_inject_timeout" option, read at create)
    int inject_skip = 0;      // ... after this many untouched registrations ("icp_inject_timeout_skip")
    FrameInput last_in;
    // ICP timing accumulators
    double icp_ms = 0.0;
    uint64_t icp_launches = 0, icp_iters = 0, icp_bytes = 0;
    kicp_host_stats hs = {};            // where the host side of the queued frames went (kicp_pipeline_host_stats)
    uint64_t map_refresh0 = 0, map_grow0 = 0, map_rehash0 = 0;  // the map's counters at the last reset
    double map_wait0 = 0.0;
    std::vector<double> pending_poses;  // row-major 4x4 per frame completed since the caller's last sync
    bool poses_stale = false;           // the caller has seen them: the next queued frame starts a new list
};

static int pipe_sync(kicp_pipeline *p, bool user_call);

// The pipeline's PipeState is the one behind its map's counters, so that one contiguous block
// [map counters | PipeState] is the whole per-frame record.
static PipeState *pipe_state(kicp_pipeline *p) { return map_mini_state(p->map); }

static int pipe_reserve(kicp_pipeline *p, size_t n) {
    if (p->cap_points && n <= p->cap_points) return KICP_OK;
    if (p->cap_points) p->hs.buffer_grows++;
    if (p->in_flight) KICP_TRY(pipe_sync(p, false));
    KICP_TRY(wait_stream(p->prep_stream, "pipeline buffers (front stages)"));
    // never less than a minimum: an EMPTY first scan must still find its buffers (the front-stage
    // kernels write their counts even for zero points)
    size_t cap = (n + n / 8 + 1024 + 1) & ~(size_t)1;  // (even: the timestamps behind cap x 24 bytes of a staging slot stay 16-byte aligned for k_stage_in's 16-byte loads)
    const size_t b3 = cap * 3 * sizeof(double);
    for (int i = 0; i < 2; ++i) {
        KICP_TRY(p->raw[i].reserve(b3));
        KICP_TRY(p->ts[i].reserve(cap * sizeof(double)));
    }
    KICP_TRY(p->tmp.reserve(b3));
    // pre/fd/src keep their contents (last frame's outputs) across a growth
    KICP_TRY(p->pre.reserve(b3, true, p->stream));
    KICP_TRY(p->fd[0].reserve(b3, true, p->stream));
    KICP_TRY(p->fd[1].reserve(b3, true, p->stream));
    KICP_TRY(p->src[0].reserve(b3, true, p->stream));
    KICP_TRY(p->src[1].reserve(b3, true, p->stream));
    KICP_TRY(p->work.reserve(b3));
    KICP_TRY(p->slot1.reserve(cap * sizeof(int)));
    KICP_TRY(p->slot2.reserve(cap * sizeof(int)));
    const uint32_t tcap = next_pow2(2 * cap);
    KICP_TRY(p->counts.reserve(3 * ((cap + 1023) / 1024 + (size_t)tcap / 1024 + 2) * sizeof(int)));
    KICP_TRY(p->sort_in.reserve(cap * sizeof(unsigned long long)));
    KICP_TRY(p->sort_out[0].reserve(cap * sizeof(unsigned long long)));
    KICP_TRY(p->sort_out[1].reserve(cap * sizeof(unsigned long long)));
    p->sort_tmp_bytes = tile_sort_temp_bytes(cap);
    KICP_TRY(p->sort_tmp.reserve(p->sort_tmp_bytes));
    if (p->run_wts.bytes < cap * sizeof(unsigned long long)) {  // tagged granules: a new buffer starts with tags no launch uses
        KICP_TRY(p->run_wts.reserve(cap * sizeof(unsigned long long)));
        KICP_HIP(hipMemsetAsync(p->run_wts.p, 0, p->run_wts.bytes, p->stream));
    }
    KICP_TRY(p->run_wts32.reserve((size_t)kIcpListRunMax * kIcpMaxBlocks * sizeof(unsigned)));
    KICP_TRY(init_ds_table(p->tab1, tcap, p->stream));
    KICP_TRY(init_ds_table(p->tab2, tcap, p->stream));
    if (p->ds_order) {
        KICP_TRY(p->rb1.reserve((size_t)tcap * 2 * sizeof(int)));
        KICP_TRY(p->rb2.reserve((size_t)tcap * 2 * sizeof(int)));
    }
    p->tab_cap = tcap;
    p->cap_points = cap;
    p->last_in.valid = false;  // the upload targets moved
    return KICP_OK;
}

// (the caller has made sure the device no longer reads the slot)
static void pipe_free_stage_slot(kicp_pipeline *p, int i) {
    if (!p->stage[i]) return;
    if (p->stage_registered) {
        (void)hipHostUnregister(p->stage[i]);
        numa::free_on_node(p->stage[i], p->stage_bytes);
    } else {
        (void)hipHostFree(p->stage[i]);
    }
    p->stage[i] = nullptr;
}

// pinned staging slots of the host-input path: [cap x 3 doubles | cap timestamps] each
static int pipe_reserve_staging(kicp_pipeline *p) {
    if (p->stage_points >= p->cap_points && p->stage[0]) return KICP_OK;
    for (int i = 0; i < kicp_pipeline::kStage; ++i) {
        if (p->stage_busy[i]) KICP_TRY(wait_event(p->ev_h2d[i], "staging slot"));
        p->stage_busy[i] = false;
        if (p->stage[i]) {
            KICP_TRY(wait_device(p->device, "staging slot release"));  // (hipHostFree waits for the whole device)
            pipe_free_stage_slot(p, i);
        }
    }
    // Where the slots lie: the runtime's pinned allocation first -- ROCm places it near the current device -- and a look at
    // the node its first page really landed on (move_pages), reported in kicp_host_stats.  With "staging_numa" = 2, slots on another
    // node than the GPU's (a two-socket 8-GPU box; a runtime that placed by the calling thread instead) are replaced by pages of
    // our own, bound to the GPU's node, registered with the runtime.  Wherever a step is refused (no NUMA information in the container, mbind not permitted, a registered
    // range whose device address differs from the host's), the runtime's allocation stays.
    const size_t bytes = ((p->cap_points * 4 * sizeof(double)) + 4095) & ~(size_t)4095;
    const int want_node = options().staging_numa != 0 ? device_numa_node(p->device) : -1;
    p->stage_registered = false;
    p->stage_node = -1;
    for (int i = 0; i < kicp_pipeline::kStage; ++i) {
        if (!p->stage_registered) {
            if (hipHostMalloc((void **)&p->stage[i], bytes) != hipSuccess) {
                set_error("pinned staging allocation of %zu bytes failed", bytes);
                return KICP_ERR_OOM;
            }
            if (i == 0) {
                memset(p->stage[0], 0, 4096);
                p->stage_node = numa::node_of_address(p->stage[0]);
                // (level 2 only: on every box this library has run on the runtime's allocation WAS on the GPU's node, so the
                // replacement below has never executed on a device -- it stays out of the default path until it has)
                if (options().staging_numa >= 2 && want_node >= 0 && p->stage_node >= 0 && p->stage_node != want_node) {
                    void *own = numa::alloc_on_node(bytes, want_node), *dev = nullptr;
                    if (own && hipHostRegister(own, bytes, hipHostRegisterDefault) == hipSuccess) {
                        if (hipHostGetDevicePointer(&dev, own, 0) == hipSuccess && dev == own) {
                            (void)hipHostFree(p->stage[0]);
                            p->stage[0] = (char *)own;
                            p->stage_registered = true;
                            p->stage_node = numa::node_of_address(own);
                            own = nullptr;
                        } else {
                            (void)hipHostUnregister(own);
                        }
                    }
                    (void)hipGetLastError();
                    if (own) numa::free_on_node(own, bytes);
                }
            }
        } else {
            void *own = numa::alloc_on_node(bytes, want_node), *dev = nullptr;
            const bool ok = own && hipHostRegister(own, bytes, hipHostRegisterDefault) == hipSuccess;
            if (!ok || hipHostGetDevicePointer(&dev, own, 0) != hipSuccess || dev != own) {
                if (ok) (void)hipHostUnregister(own);
                if (own) numa::free_on_node(own, bytes);
                (void)hipGetLastError();
                set_error("pinned staging allocation of %zu bytes on NUMA node %d failed", bytes, want_node);
                p->stage_bytes = bytes;  // (what the slots already made are freed by)
                return KICP_ERR_OOM;
            }
            p->stage[i] = (char *)own;
        }
        if (!p->ev_h2d[i]) KICP_HIP(hipEventCreateWithFlags(&p->ev_h2d[i], hipEventDisableTiming));
    }
    p->stage_bytes = bytes;
    p->stage_points = p->cap_points;
    if (!p->pool) {
        // Helper threads only as far as the host has cores for them: every pipeline of the process has a caller (or a batch
        // worker) that stages too, and the cores are the container's, not the box's -- 8 streams x (1 + 3) threads on the 16
        // CPUs a GPU box's container gets would fight each other (round 4's review).  Each pipeline's share of the usable
        // CPUs, minus one for its caller and one to spare, caps its helpers: 1 stream on 16 CPUs -> the option's 3, 8 -> 0.
        long helpers = options().staging_threads;
        const int live = g_live_pipelines.load(std::memory_order_relaxed);
        long peers = 1;  // processes of one job on this box (torch.distributed.run / MPI launchers say how many): they share the cores too
        for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"})
            if (const char *e = getenv(name)) {
                const long v = atol(e);
                if (v > peers && v <= 1024) peers = v;
            }
        const long share = available_cpus() / ((live > 0 ? live : 1) * peers) - 2;
        if (helpers > share) helpers = share;
        if (helpers < 0) helpers = 0;
        p->pool = new (std::nothrow) StagePool((int)helpers, options().staging_numa >= 2 ? device_numa_node(p->device) : -1);
        if (!p->pool) return KICP_ERR_OOM;
    }
    return KICP_OK;
}

// Tighten the host-side upper bounds of the map counters WITHOUT synchronising: the newest frame
// whose last kernel has completed left its exact counters in the pinned ring; frames queued behind
// it can each have added at most one voxel per raw point.
// the event that says "frame i (of the current batch) is completely done"; recorded events: the one in front of frame i + 1's
// registration, which therefore must have been queued
static hipEvent_t pipe_frame_done_event(kicp_pipeline *p, int i) { return p->ext_events ? p->ev_done[i] : p->ev[i + 1][0]; }

static void pipe_refresh_bounds(kicp_pipeline *p) {
    kicp_map *m = p->map;
    // ev[i + 1][0] sits in front of frame i+1's registration launch, i.e. behind frame i's last kernel
    long pending = p->in_flight > 0 ? (long)p->ring[p->in_flight - 1].n_raw : 0;
    for (int i = p->in_flight - 2; i >= 0; --i) {
        if (i + 1 <= p->done_upto || hipEventQuery(pipe_frame_done_event(p, i)) == hipSuccess) {
            if (i + 1 > p->done_upto) p->done_upto = i + 1;
            const FrameRecord &r = p->ring[i];
            const long used = (long)r.map_ctr[C_USED] + pending, bump = (long)r.map_ctr[C_BUMP] + pending,
                       live = (long)r.map_ctr[C_LIVE] + pending;
            if (used < m->used_ub) m->used_ub = used;
            if (bump < m->bump_ub) m->bump_ub = bump;
            if (live < m->live_ub) m->live_ub = live;
            return;
        }
        pending += (long)p->ring[i].n_raw;
    }
    (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
}

// Back-pressure of the asynchronous entries: at most "queue_depth" frames are kept queued on the device.  A host
// that produces scans faster than the device registers them would otherwise run arbitrarily far ahead (the HIP
// runtime then blocks inside some later call for milliseconds at a time -- measured, profiles/r03_b -- and the
// map's capacity bounds, one voxel per raw point and queued frame, grow without need); with a few frames
// queued the device never idles, and the wait, when it comes, is for work that is ahead of the caller anyway.
static int pipe_backpressure(kicp_pipeline *p) {
    const int depth = (int)options().queue_depth;
    if (depth < 2 || p->in_flight < depth) return KICP_OK;
    const int k = p->in_flight - depth + 1;  // frame k - 1 must be done = the event in front of frame k's registration
    if (k <= p->done_upto) return KICP_OK;
    if (hipEventQuery(pipe_frame_done_event(p, k - 1)) != hipSuccess) {
        (void)hipGetLastError();
        const double t0 = now_ms();
        KICP_TRY(wait_event(pipe_frame_done_event(p, k - 1), "queue back-pressure", true));
        p->hs.backpressure_waits++;
        p->hs.backpressure_ms += now_ms() - t0;
    }
    p->done_upto = k;
    return KICP_OK;
}

// Drop the tombstones of the map's slot array IN STREAM ORDER, without the host waiting for anything: used when the
// slots ever claimed (live + tombstones) approach the load limit while the live voxels alone are far from it --
// the steady state of a moving sensor, whose old voxels die as fast as new ones appear.
static int pipe_rehash_in_stream(kicp_pipeline *p) {
    kicp_map *m = p->map;
    KICP_HIP(hipMemsetAsync(m->slots.p, 0xFF, (size_t)m->slot_cap * sizeof(Slot), m->stream));
    KICP_HIP(hipMemsetAsync(m->heads.p, 0xFF, (size_t)m->slot_cap * sizeof(int), m->stream));
    launch_map_rehash(m->view(), m->bump_ub, m->stream);
    KICP_HIP(hipGetLastError());
    m->n_rehash++;
    m->used_ub = m->live_ub;  // the rebuilt array holds exactly the live voxels (records of earlier frames stay valid,
                              // if loose, upper bounds)
    return KICP_OK;
}

// queue one frame whose scan is (or will be, in prep_stream order) in HBM at d_xyz: float64 xyz
// triples, or float32 ones when xyz_f32 is set
static int pipe_enqueue(kicp_pipeline *p, const void *d_xyz, int xyz_f32, size_t n, const double *d_ts, size_t n_ts, bool host_mapped = false) {
    const kicp_config &c = p->cfg;
    const bool do_deskew = c.deskew && n_ts > 0 && d_ts;  // Preprocessing.cpp:59
    if (do_deskew && n_ts < n) {
        set_error("timestamps (%zu) shorter than frame (%zu)", n_ts, n);
        return KICP_ERR_TIMESTAMPS;
    }
    if (n > (size_t)0x7FFFFFF0 / 3) return KICP_ERR_INVALID_ARG;
    if (p->poses_stale) {
        p->pending_poses.clear();
        p->poses_stale = false;
    }
    if (p->in_flight >= kicp_pipeline::kRing) {
        const double t0 = now_ms();
        KICP_TRY(pipe_sync(p, false));
        p->hs.ring_syncs++;
        p->hs.wait_ms += now_ms() - t0;
    }
    KICP_TRY(pipe_reserve(p, n));
    kicp_map *m = p->map;
    KICP_TRY(pipe_backpressure(p));
    if (options().map_rehash_every > 0 && p->frames_enqueued > 0 && p->frames_enqueued % (uint64_t)options().map_rehash_every == 0)
        KICP_TRY(pipe_rehash_in_stream(p));  // test hook: rebuild the slot array every N frames, frames in flight or not
    if (!m->capacity_ok(n)) {
        pipe_refresh_bounds(p);
        // live voxels far from the limit, tombstones in the way: rebuild the slot array in stream order
        const size_t slack = (size_t)(options().queue_depth > 0 ? options().queue_depth : 8) + 2;
        if (!m->capacity_ok(n) && (size_t)m->bump_ub + n <= (size_t)m->blocks_cap &&
            2 * ((size_t)m->live_ub + slack * n) <= m->slot_cap)
            KICP_TRY(pipe_rehash_in_stream(p));
        if (!m->capacity_ok(n) && p->in_flight > 2 && options().queue_depth == 0) {
            // (without a queue depth the bounds can be arbitrarily loose) throttle instead of draining: wait until frame in_flight-3 is done (the
            // launch-start event of the frame behind it), i.e. at most two frames are still queued; then the bound
            // (exact counters of the newest finished frame + two frames of slack) fits
            const double t0 = now_ms();
            KICP_TRY(wait_event(pipe_frame_done_event(p, p->in_flight - 3), "map capacity"));
            p->hs.capacity_waits++;
            p->hs.wait_ms += now_ms() - t0;
            pipe_refresh_bounds(p);
        }
    }
    KICP_TRY(m->ensure_capacity(n));
    hipStream_t s = p->stream, sp = p->prep_stream;
    PipeState *st = pipe_state(p);
    const int par = (int)(p->frames_enqueued & 1u);
    PrepState *prep = p->prep.as<PrepState>() + par;
    double *fd = p->fd[par].as<double>();
    const int nblk = (int)((p->cap_points + 1023) / 1024 + p->tab_cap / 1024 + 2);
    int *cnt0 = p->counts.as<int>(), *cnt1 = cnt0 + nblk, *cnt2 = cnt1 + nblk;
    const int n_i = (int)n;

    // ===== prep_stream: everything in front of the registration ===================================
    // needs the previous frame's pose bookkeeping (last_delta for the deskew, written by its ICP
    // launch) and the buffers that launch read (src); nothing of the previous frame's map update.
    // Frame k's front stages reuse the buffers of frame k-2 (parity), so that frame must be completely
    // done: the event in front of frame k-1's registration launch says so (frames of earlier batches
    // are done anyway: the host synchronised on them).  The pose of frame k-1 is only needed to
    // deskew; without timestamps the front stages run under frame k-1's registration.
    if (p->in_flight >= 2) KICP_HIP(hipStreamWaitEvent(sp, pipe_frame_done_event(p, p->in_flight - 2), 0));
    // What depends on no pose comes BEFORE the wait for the previous frame's: with deskewing the rest of this stream is on the
    // frame's serial chain.  A scan that lies in the host's staging slot is brought into HBM here (raw / ts of this parity:
    // last read by frame k-2's front stages), and the timestamps' minimum and maximum are taken.
    const void *xyz_in = d_xyz;
    const double *ts_in = d_ts;
    if (do_deskew && host_mapped && options().stage_in != 0 && n) {
        launch_stage_in(d_xyz, p->raw[par].p, n * 3 * (xyz_f32 ? sizeof(float) : sizeof(double)), sp);
        launch_stage_in(d_ts, p->ts[par].p, n_ts * sizeof(double), sp);
        xyz_in = p->raw[par].p;
        ts_in = p->ts[par].as<double>();
    }
    if (do_deskew) launch_ts_minmax(ts_in, (int)n_ts, prep, sp);
    if (do_deskew && p->icp_done_event) KICP_HIP(hipStreamWaitEvent(sp, p->icp_done_event, 0));
    // --- Preprocess (KissICP.cpp:38) + first VoxelDownsample claim -----------------------------
    PreParams P;
    memset(&P, 0, sizeof P);
    P.xyz = xyz_in;
    P.xyz_f32 = xyz_f32;
    P.ts = do_deskew ? ts_in : nullptr;
    P.n = n_i;
    P.deskew = do_deskew ? 1 : 0;
    P.use_state_motion = 1;
    P.motion = se3_identity();
    P.state = st;
    P.prep = prep;
    P.max_range = c.max_range;
    P.min_range = c.min_range;
    P.tmp = p->tmp.as<double>();
    P.blk_counts = cnt0;
    P.out = p->pre.as<double>();
    P.n_out = &prep->n_pre;
    P.ds_tab = p->tab1.as<DsSlot>();
    P.ds_order = p->ds_order;
    P.ds_mask = p->tab_cap - 1;
    P.ds_voxel = c.voxel_size * 0.5;  // KissICP.cpp:72
    P.ds_slot_of = p->slot1.as<int>();
    P.err = &st->err;
    P.dbg = kDebugBounds ? bounds_rec(p->device) : nullptr;
    launch_pre_flags(P, sp);
    launch_pre_scatter(P, sp);

    // --- Voxelize (KissICP.cpp:70-75) ---------------------------------------------------------
    DsParams D1;
    memset(&D1, 0, sizeof D1);
    D1.order = p->ds_order;
    D1.tab_cap = (int)p->tab_cap;
    D1.rb_elem = p->rb1.as<int>();
    D1.rb_home = p->rb1.as<int>() + p->tab_cap;
    D1.in = p->pre.as<double>();
    D1.n_ptr = &prep->n_pre;
    D1.n_max = n_i;
    D1.voxel = c.voxel_size * 0.5;
    D1.tab = p->tab1.as<DsSlot>();
    D1.mask = p->tab_cap - 1;
    D1.slot_of = p->slot1.as<int>();
    D1.blk_counts = cnt1;
    D1.out = fd;
    D1.n_out = &prep->n_fd;
    D1.next_tab = p->tab2.as<DsSlot>();
    D1.next_mask = p->tab_cap - 1;
    D1.next_voxel = c.voxel_size * 1.5;  // KissICP.cpp:73
    D1.next_slot_of = p->slot2.as<int>();
    D1.err = &st->err;
    D1.dbg = kDebugBounds ? bounds_rec(p->device) : nullptr;
    DsParams D2;
    memset(&D2, 0, sizeof D2);
    D2.order = p->ds_order;
    D2.tab_cap = (int)p->tab_cap;
    D2.rb_elem = p->rb2.as<int>();
    D2.rb_home = p->rb2.as<int>() + p->tab_cap;
    D2.in = fd;
    D2.n_ptr = &prep->n_fd;
    D2.n_max = n_i;
    D2.voxel = c.voxel_size * 1.5;
    D2.tab = p->tab2.as<DsSlot>();
    D2.mask = p->tab_cap - 1;
    D2.slot_of = p->slot2.as<int>();
    D2.blk_counts = cnt2;
    D2.out = p->src[par].as<double>();
    D2.n_out = &prep->n_src;
    D2.err = &st->err;
    D2.dbg = kDebugBounds ? bounds_rec(p->device) : nullptr;
    // --- spatial order of the source cloud: the ICP kernel hands every workgroup a compact patch of it ----
    // (how many points there will be is known on the device only; the previous frame's count says how to sort: a small cloud
    // by rank in one launch, its keys written beside it by the scatter that emits it)
    const bool sorted = n <= ((size_t)1 << 24);
    const size_t sort_hint = p->have_last ? (size_t)p->last.st.n_src : n / 8;
    const bool keys_ready = sorted && n && tile_sort_by_rank(n, sort_hint);
    if (keys_ready) {
        D2.sort_keys = p->sort_in.as<unsigned long long>();
        D2.sort_inv_cell = tile_sort_inv_cell(c.voxel_size);
    }
    if (p->ds_order) {  // the reference's output order: arrange the grid's clusters, then compact bucket by bucket
        launch_ds_arrange(D1, sp);
        launch_ds_scatter_rb(D1, sp);
        launch_ds_arrange(D2, sp);
        launch_ds_scatter_rb(D2, sp);
    } else {
        launch_ds_flags(D1, sp);
        launch_ds_scatter(D1, sp);
        launch_ds_flags(D2, sp);
        launch_ds_scatter(D2, sp);
    }
    KICP_HIP(hipGetLastError());
    if (sorted && n) {
        const int se = launch_tile_sort(p->src[par].as<double>(), &prep->n_src, 0, n, c.voxel_size, p->sort_in.as<unsigned long long>(),
                                        p->sort_out[par].as<unsigned long long>(), sort_hint, sp, keys_ready);
        if (se != 0) {
            set_error("device sort of the source cloud failed (%s)", hipGetErrorString((hipError_t)se));
            return KICP_ERR_HIP;
        }
    }
    KICP_HIP(hipEventRecord(p->ev_prep_done[par], sp));

    // ===== stream: the serial chain =================================================================
    inject_stall(s);
    KICP_HIP(hipStreamWaitEvent(s, p->ev_prep_done[par], 0));
    // --- AlignPointsToMap + threshold / pose bookkeeping (KissICP.cpp:44-63) ---------------------
    IcpParams I;
    memset(&I, 0, sizeof I);
    // the source cloud is at most the scan; in practice ~1/60 of it (two voxel downsamples): the LDS policy
    // goes by the previous frame's count when there is one
    const size_t n_src_hint = p->have_last ? (size_t)p->last.st.n_src : n / 32;
    const int G = icp_fill_policy(p->device, I, n_src_hint, p->icp_cap, p->icp_share);
    I.frame = p->src[par].as<double>();
    I.order = sorted ? p->sort_out[par].as<unsigned long long>() : nullptr;
    if (sorted && n) {  // runs of equal weight, settled by the kernel's own prologue
        I.wts = p->run_wts.as<unsigned long long>();
        I.wts32 = options().icp_weights_kernel != 0 ? p->run_wts32.as<unsigned>() : nullptr;
        I.weight_base = (int)options().icp_weight_base;
        I.weight_quad = (int)options().icp_weight_quad;
        I.weight_long_base = (int)options().icp_weight_long_base;
        I.weight_long_emul = (int)options().icp_weight_long_emul;
        I.weight_dense_min = (int)options().icp_weight_dense_min;
        I.weight_dense_div = (int)options().icp_weight_dense_div;
    }
    I.work = p->work.as<double>();
    I.n_ptr = &prep->n_src;
    I.prep = prep;
    I.map = m->view();
    I.state = st;
    I.pipeline_mode = 1;
    I.min_motion_th = c.min_motion_th;
    I.max_iters = c.max_num_iterations;
    I.conv = c.convergence_criterion;
    I.granules = p->granules.as<unsigned long long>();
    I.spin_limit = kSpinLimit;
    if (p->inject_skip > 0) {
        p->inject_skip--;
    } else if (p->inject_timeouts > 0) {
        I.inject_timeout = 1;
        p->inject_timeouts--;
    }
    if (options().icp_profile != 0) {
        KICP_TRY(p->prof_groups.reserve(kIcpGroupProfileWords * sizeof(unsigned)));
        I.prof_groups = p->prof_groups.as<unsigned>();
    }
    // the run weights of a cloud of short runs, by a kernel of their own in front of the launch ("icp_weights_kernel", kicp_icp.hip)
    if (I.wts && I.wts32) launch_icp_weights(I, G, n_src_hint, s);
    else I.wts32 = nullptr;
    const int slot = p->in_flight;
    // Two events per frame, each doing double duty: ev[slot][0] in front of the launch opens the timing
    // bracket and marks "the previous frame is completely done" (buffer reuse, capacity bounds);
    // ev[slot][1] behind it closes the bracket and tells prep_stream the pose is ready.
    // (the pose-ready event is skipped when nobody would look at it: timing off and a configuration that never deskews)
    const bool want_done = options().icp_timing != 0 || c.deskew;
    p->icp_done_event = nullptr;
    if (p->ext_events) {
        KICP_TRY(icp_launch_ordered(p->device, I, G, options().icp_profile != 0, s, p->icp_share, options().icp_timing != 0 ? p->ev[slot][0] : nullptr,
                                    want_done ? p->ev[slot][1] : nullptr));
        if (want_done) p->icp_done_event = p->ev[slot][1];
    } else {
        KICP_HIP(hipEventRecord(p->ev[slot][0], s));
        KICP_TRY(icp_launch_ordered(p->device, I, G, options().icp_profile != 0, s, p->icp_share));
        if (want_done) {
            p->icp_done_event = p->ev[slot][1];
            KICP_HIP(hipEventRecord(p->icp_done_event, s));
        }
    }

    // --- local_map_.Update(frame_downsample, new_pose) (KissICP.cpp:61) --------------------------
    InsertScratch sc;
    KICP_TRY(m->scratch_reserve(p->cap_points, sc));
    const MapView v = m->view();
    // (RemovePointsFarFromLocation rides in the two kernels of AddPoints -- "map_fused_update" --, or follows as a third)
    FrameRecord *rec = p->ring + slot;
    rec->n_raw = n;
    MapPrune mp;
    memset(&mp, 0, sizeof mp);
    mp.bump_ub = m->bump_ub;
    mp.state = st;
    mp.use_state_origin = 1;
    mp.host_rec = reinterpret_cast<unsigned *>(rec);  // the frame record, written by the frame's last kernel itself into the pinned host ring
    mp.rec_words = kRecWords;
    const bool fused = options().map_fused_update != 0;
    launch_map_link(v, sc, fd, &prep->n_fd, 0, n_i, st, 1, s, fused ? &mp : nullptr);
    launch_map_apply(v, sc, n_i, s, fused ? &mp : nullptr, fused && p->ext_events ? p->ev_done[slot] : nullptr);
    m->used_ub += (long)n;
    m->live_ub += (long)n;
    m->bump_ub += (long)n;
    if (m->bump_ub > m->blocks_cap) m->bump_ub = m->blocks_cap;
    if (!fused) launch_map_prune(v, m->bump_ub, st, 1, nullptr, mp.host_rec, kRecWords, s, p->ext_events ? p->ev_done[slot] : nullptr);
    KICP_HIP(hipGetLastError());
    p->in_flight++;
    p->frames_enqueued++;
    p->hs.frames++;
    p->last_in.valid = true;
    p->last_in.host_mapped = host_mapped;
    p->last_in.d_xyz = d_xyz;
    p->last_in.xyz_f32 = xyz_f32;
    p->last_in.n = n;
    p->last_in.d_ts = d_ts;
    p->last_in.n_ts = n_ts;
    return KICP_OK;
}

// Host-input RegisterFrame, first half: copy the caller's scan into a pinned staging slot (the caller's
// buffer is free again when this returns), start its upload on prep_stream -- where it runs under the
// previous frame's registration -- and queue the frame behind it.  Exactly one of xyz64 / xyz32 is set.
// float64 scans whose values are all exactly representable in float32 (they came from a float32 sensor
// file) are narrowed on the way: half the bytes over PCIe, bit-identical points on the device.
static int pipe_stage_and_enqueue(kicp_pipeline *p, const double *xyz64, const float *xyz32, size_t n, const double *ts,
                                  size_t n_ts) {
    const bool have_ts = ts && n_ts > 0;
    if (!have_ts) n_ts = 0;
    if (p->cfg.deskew && have_ts && n_ts < n) {
        set_error("timestamps (%zu) shorter than frame (%zu)", n_ts, n);
        return KICP_ERR_TIMESTAMPS;
    }
    if (n > (size_t)0x7FFFFFF0 / 3 || n_ts > (size_t)0x7FFFFFF0) return KICP_ERR_INVALID_ARG;
    const double t_call = now_ms();
    if (p->in_flight >= kicp_pipeline::kRing) {
        KICP_TRY(pipe_sync(p, false));
        p->hs.ring_syncs++;
        p->hs.wait_ms += now_ms() - t_call;
    }
    KICP_TRY(pipe_backpressure(p));
    KICP_TRY(pipe_reserve(p, n > n_ts ? n : n_ts));
    KICP_TRY(pipe_reserve_staging(p));
    const int slot = (int)(p->staged % kicp_pipeline::kStage);
    if (p->stage_busy[slot]) {  // its previous upload (four frames ago) must have left the slot
        if (hipEventQuery(p->ev_h2d[slot]) != hipSuccess) {
            (void)hipGetLastError();
            const double t0 = now_ms();
            KICP_TRY(wait_event(p->ev_h2d[slot], "staging slot"));
            p->hs.staging_waits++;
            p->hs.wait_ms += now_ms() - t0;
        }
        p->stage_busy[slot] = false;
    }
    const double t_stage = now_ms();
    const double wait_before = p->hs.wait_ms + p->map->wait_ms;
    char *h = p->stage[slot];
    double *h_ts = reinterpret_cast<double *>(h + p->stage_points * 3 * sizeof(double));
    constexpr size_t kChunk = 16384;  // values per task
    const size_t n_val = n * 3;
    const int n_chunks = (int)((n_val + kChunk - 1) / kChunk);
    const int ts_chunks = (int)((n_ts + kChunk - 1) / kChunk);
    bool as_f32 = xyz32 != nullptr;
    if (xyz32) {
        p->pool->run(n_chunks + ts_chunks, [&](int c) {
            if (c < n_chunks) {
                const size_t a = (size_t)c * kChunk, b = a + kChunk < n_val ? a + kChunk : n_val;
                memcpy(reinterpret_cast<float *>(h) + a, xyz32 + a, (b - a) * sizeof(float));
            } else {
                const size_t a = (size_t)(c - n_chunks) * kChunk, b = a + kChunk < n_ts ? a + kChunk : n_ts;
                memcpy(h_ts + a, ts + a, (b - a) * sizeof(double));
            }
        });
    } else {
        bool try_f32 = options().staging_f32 != 0 && p->f32_skip == 0;
        if (!try_f32 && p->f32_skip > 0) p->f32_skip--;
        if (try_f32) {
            std::atomic<int> inexact{0};
            p->pool->run(n_chunks + ts_chunks, [&](int c) {
                if (c < n_chunks) {
                    if (inexact.load(std::memory_order_relaxed)) return;  // the attempt is void anyway
                    const size_t a = (size_t)c * kChunk, b = a + kChunk < n_val ? a + kChunk : n_val;
                    if (!narrow_exact(xyz64 + a, reinterpret_cast<float *>(h) + a, b - a)) inexact.store(1, std::memory_order_relaxed);
                } else {
                    const size_t a = (size_t)(c - n_chunks) * kChunk, b = a + kChunk < n_ts ? a + kChunk : n_ts;
                    memcpy(h_ts + a, ts + a, (b - a) * sizeof(double));
                }
            });
            if (inexact.load()) {
                try_f32 = false;
                p->f32_skip = 32;  // genuinely float64 data: do not pay for the attempt on every frame
            } else {
                as_f32 = true;
            }
            if (!as_f32)
                p->pool->run(n_chunks, [&](int c) {
                    const size_t a = (size_t)c * kChunk, b = a + kChunk < n_val ? a + kChunk : n_val;
                    memcpy(reinterpret_cast<double *>(h) + a, xyz64 + a, (b - a) * sizeof(double));
                });
        } else {
            p->pool->run(n_chunks + ts_chunks, [&](int c) {
                if (c < n_chunks) {
                    const size_t a = (size_t)c * kChunk, b = a + kChunk < n_val ? a + kChunk : n_val;
                    memcpy(reinterpret_cast<double *>(h) + a, xyz64 + a, (b - a) * sizeof(double));
                } else {
                    const size_t a = (size_t)(c - n_chunks) * kChunk, b = a + kChunk < n_ts ? a + kChunk : n_ts;
                    memcpy(h_ts + a, ts + a, (b - a) * sizeof(double));
                }
            });
        }
    }
    const int par = (int)(p->frames_enqueued & 1u);
    hipStream_t sp = p->prep_stream;
    const double t_copied = now_ms();
    const int in_flight_before = p->in_flight;
    int st;
    double t_enq;
    if (options().staging_zero_copy != 0) {
        // No upload at all: the staging slot is pinned, device-mapped host memory, and the only kernels that read the
        // raw scan (k_ts_minmax, k_pre_flags) read it once, front to back -- they fetch it over PCIe themselves.  One
        // DMA hand-off less per frame, and the copy engines stay out of the frame path (their queues are created
        // lazily, ~5-8 ms each, the first time the runtime finds the ones it has busy: measured as a one-off stall of
        // the 10th frame of a process, profiles/r03_c).
        void *d_h = nullptr;
        KICP_HIP(hipHostGetDevicePointer(&d_h, h, 0));
        const double *d_hts = reinterpret_cast<const double *>(static_cast<char *>(d_h) + p->stage_points * 3 * sizeof(double));
        t_enq = now_ms();
        st = pipe_enqueue(p, d_h, as_f32 ? 1 : 0, n, n_ts ? d_hts : nullptr, n_ts, true);
        KICP_HIP(hipEventRecord(p->ev_h2d[slot], sp));  // behind the front stages of this frame: the slot is free again
    } else {
        // upload on the stream that consumes it.  raw[par] / ts[par] were last read by the front stages of
        // frame k-2, which precede this copy in stream order.
        if (n) KICP_HIP(hipMemcpyAsync(p->raw[par].p, h, n_val * (as_f32 ? sizeof(float) : sizeof(double)), hipMemcpyHostToDevice, sp));
        if (n_ts) KICP_HIP(hipMemcpyAsync(p->ts[par].p, h_ts, n_ts * sizeof(double), hipMemcpyHostToDevice, sp));
        KICP_HIP(hipEventRecord(p->ev_h2d[slot], sp));
        t_enq = now_ms();
        st = pipe_enqueue(p, p->raw[par].p, as_f32 ? 1 : 0, n, n_ts ? p->ts[par].as<double>() : nullptr, n_ts);
    }
    p->stage_busy[slot] = true;
    p->staged++;
    const double t_end = now_ms();
    const double waited = (p->hs.wait_ms + p->map->wait_ms) - wait_before;  // inside pipe_enqueue
    p->hs.stage_ms += t_copied - t_stage;
    p->hs.enqueue_ms += (t_end - t_copied) - waited;
    if (t_end - t_call > p->hs.max_call_ms) p->hs.max_call_ms = t_end - t_call;
    if (host_trace_on())
        fprintf(stderr, "[kicp host] frame %llu in_flight %d: call %.3f ms = pre %.3f + copy %.3f (%s) + h2d calls %.3f + enqueue %.3f (waited %.3f); waits backpressure %llu cap %llu stage %llu refresh %llu grow %llu rehash %llu\n",
                (unsigned long long)p->frames_enqueued - 1, in_flight_before, t_end - t_call, t_stage - t_call, t_copied - t_stage, as_f32 ? "f32" : "f64",
                t_enq - t_copied, t_end - t_enq, waited, (unsigned long long)p->hs.backpressure_waits, (unsigned long long)p->hs.capacity_waits, (unsigned long long)p->hs.staging_waits,
                (unsigned long long)p->map->n_refresh, (unsigned long long)p->map->n_grow, (unsigned long long)p->map->n_rehash);
    return st;
}

extern "C" {

int kicp_config_default(kicp_config *c) {
    if (!c) return KICP_ERR_INVALID_ARG;
    c->voxel_size = 1.0;
    c->max_range = 100.0;
    c->min_range = 0.0;
    c->max_points_per_voxel = 20;
    c->min_motion_th = 0.1;
    c->initial_threshold = 2.0;
    c->max_num_iterations = 500;
    c->convergence_criterion = 0.0001;
    c->max_num_threads = 0;
    c->deskew = 1;
    return KICP_OK;
}

int kicp_pipeline_create(const kicp_config *cfg, int device_id, kicp_pipeline **out) {
    return kicp::pipeline_create_shared(cfg, device_id, 0, out);
}

}  // extern "C"

// share > 0: the pipeline's part of its device's persistent grid (1 / share), whatever option "icp_device_streams" says -- the
// batch entry knows how many of its streams sit on each device and says so per pipeline (round 4's review: it used to
// overwrite the process-wide option around the creation of its pipelines)
int kicp::pipeline_create_shared(const kicp_config *cfg, int device_id, int share, kicp_pipeline **out) {
    if (!cfg || !out) return KICP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!(cfg->voxel_size > 0.0) || cfg->max_points_per_voxel <= 0 || cfg->max_num_iterations < 0) {
        set_error("invalid KISSConfig");
        return KICP_ERR_INVALID_ARG;
    }
    KICP_TRY(check_device(device_id));
    kicp_pipeline *p = new (std::nothrow) kicp_pipeline();
    if (!p) return KICP_ERR_OOM;
    p->device = device_id;
    p->cfg = *cfg;
    g_live_pipelines.fetch_add(1, std::memory_order_relaxed);
    p->inject_timeouts = (int)options().icp_inject_timeout;
    p->inject_skip = (int)options().icp_inject_timeout_skip;
    p->ds_order = options().downsample_order != 0 ? 1 : 0;
    p->icp_share = share > 0 ? share : (int)(options().icp_device_streams > 1 ? options().icp_device_streams : 1);
    if (p->icp_share > kIcpMaxLanes) p->icp_share = kIcpMaxLanes;
    icp_gate_count_share(device_id, p->icp_share, +1);
    p->share_counted = true;
    int s = KICP_OK;
    // The two streams of a pipeline must sit on DIFFERENT hardware queues: the registration is one persistent launch, and a
    // front-stage kernel queued behind it on the same queue waits for it to end.  The runtime hands streams of one priority
    // to a few hardware queues in turn, so two streams of a pipeline could share one -- measured: whichever pipeline was
    // created second in a process lost the overlap, +87 us between its registrations, 2140 instead of 2585 scans/s
    // (profiles/r04_aa_bench_200_10.json, r04_ac_bench_swapped.json: it is the position, not the input path).  Streams of
    // different priorities never share a queue: the registration's stream gets the higher one.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&p->prep_stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_prep_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_prep_done[1], hipEventDisableTiming) != hipSuccess)
        s = KICP_ERR_HIP;
    stream_register(device_id, p->stream);
    stream_register(device_id, p->prep_stream);
    p->ext_events = options().frame_events != 0;
    if (s == KICP_OK) {
        p->ev_ok = true;
        for (int i = 0; i < kicp_pipeline::kRing && p->ev_ok; ++i) {
            for (int j = 0; j < 2; ++j)
                if (hipEventCreate(&p->ev[i][j]) != hipSuccess) p->ev_ok = false;
            if (hipEventCreate(&p->ev_done[i]) != hipSuccess) p->ev_ok = false;
        }
    }
    if (s == KICP_OK && !p->ev_ok) s = KICP_ERR_HIP;  // the events order buffer reuse between the two streams
    if (s == KICP_OK && hipHostMalloc((void **)&p->ring, sizeof(FrameRecord) * kicp_pipeline::kRing) != hipSuccess)
        s = KICP_ERR_OOM;
    // KissICP.hpp:62-68: local_map_(voxel_size, max_range, max_points_per_voxel)
    if (s == KICP_OK)
        s = map_create_on_stream(cfg->voxel_size, cfg->max_range, (unsigned)cfg->max_points_per_voxel, device_id,
                                 p->stream, &p->map);
    if (s == KICP_OK && (icp_prepare(device_id) != 0 || tile_sort_prepare(device_id) != 0)) {
        set_error("hipFuncSetAttribute(k_icp, 160 KiB LDS) failed");
        s = KICP_ERR_HIP;
    }
    if (s == KICP_OK) s = p->granules.reserve(icp_granule_words(kIcpMaxBlocks) * sizeof(unsigned long long));
    if (s == KICP_OK) s = p->prep.reserve(2 * sizeof(PrepState));
    if (s == KICP_OK) s = pipe_reserve(p, 0);  // minimum buffers: the first scan may be empty
    if (s != KICP_OK) {
        if (s == KICP_ERR_HIP) set_error("pipeline resource creation failed");
        kicp_pipeline_destroy(p);
        return s;
    }
    PipeState st;
    init_state(st, cfg->initial_threshold);
    KICP_HIP(hipMemcpyAsync(pipe_state(p), &st, sizeof st, hipMemcpyHostToDevice, p->stream));
    KICP_HIP(hipMemsetAsync(p->granules.p, 0, p->granules.bytes, p->stream));
    const PrepState idle[2] = {idle_prep_state(), idle_prep_state()};
    KICP_HIP(hipMemcpyAsync(p->prep.p, idle, sizeof idle, hipMemcpyHostToDevice, p->stream));
    {
        const int ws = wait_stream(p->stream, "pipeline creation");
        if (ws != KICP_OK) {
            kicp_pipeline_destroy(p);
            return ws;
        }
    }
    memset(&p->last, 0, sizeof p->last);
    p->last.st = st;
    *out = p;
    return KICP_OK;
}

extern "C" {

int kicp_pipeline_destroy(kicp_pipeline *p) {
    if (!p) return KICP_OK;
    (void)hipSetDevice(p->device);
    // Everything the pipeline has queued must have ended before its memory goes; every wait has its deadline, and a device
    // that does not answer keeps what it may still touch (device and pinned memory, streams, events are leaked): the call
    // returns KICP_ERR_TIMEOUT instead of never.
    int idle = KICP_OK;
    for (hipStream_t s : {p->prep_stream, p->copy_stream, p->stream})
        if (s && idle == KICP_OK) idle = wait_stream(s, "pipeline teardown");
    if (idle == KICP_OK) idle = wait_device(p->device, "pipeline teardown");  // (hipHostFree below waits for the whole device)
    const bool gone = idle == KICP_OK;  // nothing of this pipeline is in flight any more
    g_live_pipelines.fetch_sub(1, std::memory_order_relaxed);
    if (p->share_counted) icp_gate_count_share(p->device, p->icp_share, -1);
    delete p->pool;
    p->pool = nullptr;
    if (p->map) {  // (the map lives on the pipeline's stream: the wait above was its wait too)
        kicp_map *m = p->map;
        for (DevBuf *b : {&m->slots, &m->heads, &m->blocks, &m->free_ids, &m->doomed, &m->ctr, &m->pts_in, &m->world, &m->next, &m->rec_slot, &m->rec_block, &m->rec_count,
                          &m->rec_head, &m->rec_list})
            b->drop(gone);
        delete m;
        p->map = nullptr;
    }
    for (DevBuf *b : {&p->raw[0], &p->raw[1], &p->ts[0], &p->ts[1], &p->tmp, &p->pre, &p->fd[0], &p->fd[1], &p->src[0], &p->src[1],
                      &p->work, &p->slot1, &p->slot2, &p->tab1, &p->tab2, &p->rb1, &p->rb2, &p->counts, &p->granules, &p->prof_groups, &p->prep,
                      &p->sort_in, &p->sort_out[0], &p->sort_out[1], &p->sort_tmp, &p->run_wts, &p->run_wts32})
        b->drop(gone);
    if (gone) {
        for (int i = 0; i < 2; ++i)
            if (p->ev_prep_done[i]) (void)hipEventDestroy(p->ev_prep_done[i]);
        for (int i = 0; i < kicp_pipeline::kStage; ++i) {
            if (p->ev_h2d[i]) (void)hipEventDestroy(p->ev_h2d[i]);
            pipe_free_stage_slot(p, i);
        }
        if (p->out_stage) (void)hipHostFree(p->out_stage);
        if (p->ev_ok)
            for (int i = 0; i < kicp_pipeline::kRing; ++i) {
                for (int j = 0; j < 2; ++j) (void)hipEventDestroy(p->ev[i][j]);
                (void)hipEventDestroy(p->ev_done[i]);
            }
        if (p->ring) (void)hipHostFree(p->ring);
    }
    for (hipStream_t s : {p->copy_stream, p->prep_stream, p->stream}) {
        if (!s) continue;
        if (!gone) continue;  // (leaked with the rest: kept in the registry, and -- its k_icp may be in flight -- in its lane of the launch gate)
        if (s == p->stream) icp_forget_stream(p->device, s);
        (void)stream_destroy(s);
    }
    delete p;
    return gone ? KICP_OK : idle;
}

// fold the completed frames [0, count) of the ring into the accumulators
static void pipe_collect(kicp_pipeline *p, int count, int &err_bits) {
    for (int i = 0; i < count; ++i) {
        const FrameRecord &r = p->ring[i];
        err_bits |= r.st.err | r.map_ctr[C_ERR];
        if (options().icp_timing && p->ev_ok) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p->ev[i][0], p->ev[i][1]) == hipSuccess) p->icp_ms += ms;
            else (void)hipGetLastError();  // e.g. the option was switched on while frames were queued
            if (i > 0) {  // device time between two registrations: the previous frame's map update + the front of this one
                float gap = 0.f;
                hipError_t ge;
                if (p->ext_events) {  // (stop to stop, minus this registration: a start event reads as its dispatch's end outside its own pair)
                    ge = hipEventElapsedTime(&gap, p->ev[i - 1][1], p->ev[i][1]);
                    gap -= ms;
                    if (gap < 0.f) gap = 0.f;
                } else {
                    ge = hipEventElapsedTime(&gap, p->ev[i - 1][1], p->ev[i][0]);
                }
                if (ge == hipSuccess) {
                    p->hs.device_gap_ms += gap;
                    if (gap > p->hs.max_device_gap_ms) p->hs.max_device_gap_ms = gap;
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        if (host_trace_on() && p->ev_ok) {
            float t_start = 0.f, t_icp = 0.f;  // device-side timeline of the batch: registration start relative to the first frame's
            if (hipEventElapsedTime(&t_start, p->ev[0][0], p->ev[i][0]) != hipSuccess) (void)hipGetLastError();
            if (hipEventElapsedTime(&t_icp, p->ev[i][0], p->ev[i][1]) != hipSuccess) (void)hipGetLastError();
            fprintf(stderr, "[kicp device] batch frame %d: registration starts at %.3f ms, runs %.3f ms (%d iterations, n_src %d, map %d voxels)\n", i,
                    t_start, t_icp, r.st.icp_iterations, r.st.n_src, r.map_ctr[C_LIVE]);
        }
        p->icp_launches++;
        p->icp_iters += (uint64_t)r.st.icp_iterations;
        // algorithmic bytes of AlignPointsToMap:
        //   per iteration N_src*(24+24) + N_src*27*16 + E*24 + 336
        p->icp_bytes += (uint64_t)r.st.icp_iterations * ((uint64_t)r.st.n_src * (48 + 27 * 16) + 336) +
                        r.st.icp_examined * 24;
        double T[16];
        se3_matrix(r.st.last_pose, T);
        p->pending_poses.insert(p->pending_poses.end(), T, T + 16);
    }
    if (count) {
        p->last = p->ring[count - 1];
        p->have_last = true;
        p->frames_done += (uint64_t)count;
        kicp_map *m = p->map;
        memcpy(m->h_ctr, p->last.map_ctr, sizeof m->h_ctr);
        m->used_ub = m->h_ctr[C_USED];
        m->live_ub = m->h_ctr[C_LIVE];
        m->bump_ub = m->h_ctr[C_BUMP] < m->blocks_cap ? m->h_ctr[C_BUMP] : m->blocks_cap;
    }
}

int kicp_pipeline_sync(kicp_pipeline *p) {
    if (!p) return KICP_ERR_INVALID_ARG;
    return pipe_sync(p, true);
}

}  // extern "C"

// user_call = false: a wait the library inserted on its own (ring full, buffers growing); the poses it
// completes stay on the caller's list
static int pipe_sync_impl(kicp_pipeline *p);
static int pipe_sync(kicp_pipeline *p, bool user_call) {
    KICP_HIP(hipSetDevice(p->device));
    if (p->poses_stale) {
        p->pending_poses.clear();
        p->poses_stale = false;
    }
    // The list is marked "seen by the caller" only when the caller's sync RETURNS: a frame replayed inside it (after a
    // registration that gave up) goes through pipe_enqueue, which must neither drop the poses collected a moment ago nor
    // leave the flag cleared behind it.
    const int st = pipe_sync_impl(p);
    if (user_call) p->poses_stale = true;
    return st;
}
static int pipe_sync_impl(kicp_pipeline *p) {
    for (int attempt = 0;; ++attempt) {
        KICP_TRY(wait_stream(p->stream, "pipeline sync"));  // (gives up with everything still queued: a later sync collects it)
        // a registration whose workgroups were not all resident gives up (bounded spin), commits nothing
        // and poisons the frames queued behind it (they do nothing either): the frames in front are good
        int good = p->in_flight;
        for (int i = 0; i < p->in_flight; ++i)
            if (p->ring[i].st.err & E_TIMEOUT) {
                good = i;
                break;
            }
        int err_bits = 0;
        const int queued = p->in_flight;
        pipe_collect(p, good, err_bits);
        p->in_flight = 0;
        p->done_upto = 0;
        if (good < queued) {
            PipeState *st = pipe_state(p);
            const int zero = 0;
            KICP_HIP(hipMemcpy(&st->err, &zero, sizeof zero, hipMemcpyHostToDevice));
            // the map's host-side bounds counted inserts that never happened: refresh them
            KICP_TRY(p->map->refresh_counters());
            // next time with half as many workgroups
            IcpParams probe;
            memset(&probe, 0, sizeof probe);
            const int grid = icp_fill_policy(p->device, probe, p->have_last ? (size_t)p->last.st.n_src : 0, p->icp_cap, p->icp_share);
            p->icp_cap = grid > 1 ? grid / 2 : 1;
            p->frames_enqueued -= (uint64_t)(queued - good);
            if (queued - good == 1 && p->last_in.valid && attempt < 3 && !(err_bits & ~E_TIMEOUT)) {
                // exactly the last frame is missing and its scan is still where it was: run it again
                const FrameInput in = p->last_in;
                KICP_TRY(pipe_enqueue(p, in.d_xyz, in.xyz_f32, in.n, in.d_ts, in.n_ts, in.host_mapped));
                continue;
            }
            set_error("registration gave up waiting for its workgroups (%d frame(s) not processed; re-submit them)", queued - good);
            return KICP_ERR_TIMEOUT;
        }
        if (err_bits) {
            PipeState *st = pipe_state(p);
            int zero = 0;
            KICP_HIP(hipMemcpy(&st->err, &zero, sizeof zero, hipMemcpyHostToDevice));
            KICP_HIP(hipMemcpy(p->map->ctr.as<int>() + C_ERR, &zero, sizeof zero, hipMemcpyHostToDevice));
            return err_bits_to_status(err_bits);
        }
        return KICP_OK;
    }
}

extern "C" {

int kicp_pipeline_register_frame_device(kicp_pipeline *p, const double *d_xyz, size_t n, const double *d_ts,
                                        size_t n_ts) {
    if (!p || (!d_xyz && n)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    return pipe_enqueue(p, d_xyz, 0, n, d_ts, n_ts);
}

int kicp_pipeline_register_frame_async(kicp_pipeline *p, const double *xyz, size_t n, const double *timestamps,
                                       size_t n_ts) {
    if (!p || (!xyz && n)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    return pipe_stage_and_enqueue(p, xyz, nullptr, n, timestamps, n_ts);
}

int kicp_pipeline_register_frame_async_f32(kicp_pipeline *p, const float *xyz, size_t n, const double *timestamps,
                                           size_t n_ts) {
    if (!p || (!xyz && n)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    static const float kNone[3] = {0.f, 0.f, 0.f};
    return pipe_stage_and_enqueue(p, nullptr, xyz ? xyz : kNone, n, timestamps, n_ts);
}

int kicp_pipeline_register_frame(kicp_pipeline *p, const double *xyz, size_t n, const double *timestamps,
                                 size_t n_ts) {
    if (!p || (!xyz && n)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    static const double kNone[3] = {0.0, 0.0, 0.0};
    KICP_TRY(pipe_stage_and_enqueue(p, xyz ? xyz : kNone, nullptr, n, timestamps, n_ts));
    return kicp_pipeline_sync(p);
}


static int pipe_state_get(kicp_pipeline *p, PipeState &h) {
    if (p->in_flight) KICP_TRY(kicp_pipeline_sync(p));
    h = p->last.st;
    return KICP_OK;
}

int kicp_pipeline_pose(kicp_pipeline *p, double T[16]) {
    if (!p || !T) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    se3_matrix(h.last_pose, T);
    return KICP_OK;
}

int kicp_pipeline_delta(kicp_pipeline *p, double T[16]) {
    if (!p || !T) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    se3_matrix(h.last_delta, T);
    return KICP_OK;
}

static int pipe_set_se3(kicp_pipeline *p, const double T[16], bool delta) {
    if (!p || !T) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    SE3 x;
    if (!se3_from_matrix(T, x)) {
        set_error("not a rigid transform (SOPHUS_ENSURE)");
        return KICP_ERR_INVALID_ARG;
    }
    if (p->in_flight) KICP_TRY(kicp_pipeline_sync(p));
    PipeState *st = pipe_state(p);
    KICP_HIP(hipMemcpy(delta ? &st->last_delta : &st->last_pose, &x, sizeof x, hipMemcpyHostToDevice));
    (delta ? p->last.st.last_delta : p->last.st.last_pose) = x;
    return KICP_OK;
}
int kicp_pipeline_set_pose(kicp_pipeline *p, const double T[16]) { return pipe_set_se3(p, T, false); }
int kicp_pipeline_set_delta(kicp_pipeline *p, const double T[16]) { return pipe_set_se3(p, T, true); }

int kicp_pipeline_map(kicp_pipeline *p, kicp_map **map) {
    if (!p || !map) return KICP_ERR_INVALID_ARG;
    if (p->in_flight) KICP_TRY(kicp_pipeline_sync(p));
    *map = p->map;
    return KICP_OK;
}

int kicp_pipeline_output_size(kicp_pipeline *p, int which, size_t *n) {
    if (!p || !n || which < 0 || which > 2) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    if (!p->have_last) {
        *n = 0;
        return KICP_OK;
    }
    *n = (size_t)(which == KICP_OUT_PREPROCESSED ? h.n_pre : which == KICP_OUT_SOURCE ? h.n_src : h.n_fd);
    return KICP_OK;
}

// pinned bounce buffer of the output downloads
static int pipe_out_stage(kicp_pipeline *p, size_t bytes) {
    if (p->out_stage_bytes >= bytes) return KICP_OK;
    if (p->out_stage) {
        KICP_TRY(wait_device(p->device, "output buffer growth"));  // (hipHostFree waits for the whole device)
        KICP_HIP(hipHostFree(p->out_stage));
    }
    p->out_stage = nullptr;
    p->out_stage_bytes = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc((void **)&p->out_stage, want) != hipSuccess) {
        set_error("pinned output buffer of %zu bytes failed", want);
        return KICP_ERR_OOM;
    }
    p->out_stage_bytes = want;
    return KICP_OK;
}
// pinned -> the caller's pageable memory, spread over the helper threads
static void pipe_spread_copy(kicp_pipeline *p, void *dst_, const void *src_, size_t bytes) {
    constexpr size_t kPiece = 256 * 1024;
    const int pieces = (int)((bytes + kPiece - 1) / kPiece);
    char *dst = static_cast<char *>(dst_);
    const char *src = static_cast<const char *>(src_);
    p->pool->run(pieces, [&](int k) {
        const size_t a = (size_t)k * kPiece, e = a + kPiece < bytes ? a + kPiece : bytes;
        memcpy(dst + a, src + a, e - a);
    });
}

int kicp_pipeline_output(kicp_pipeline *p, int which, double *out, size_t cap, size_t *n) {
    if (!p || !n || (!out && cap)) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    size_t cnt = 0;
    KICP_TRY(kicp_pipeline_output_size(p, which, &cnt));
    *n = cnt;
    const size_t c = cnt < cap ? cnt : cap;
    if (c) {
        const unsigned par = (unsigned)((p->frames_enqueued - 1) & 1u);  // the last frame's buffers
        const DevBuf &b = which == KICP_OUT_PREPROCESSED ? p->pre : which == KICP_OUT_SOURCE ? p->src[par] : p->fd[par];
        const size_t bytes = c * 3 * sizeof(double);
        if (bytes < (size_t)256 * 1024) {  // small clouds: one direct copy (the stream is idle: the size query above synchronised)
            KICP_HIP(hipMemcpy(out, b.p, bytes, hipMemcpyDeviceToHost));
            return KICP_OK;
        }
        // large clouds (the preprocessed frame is ~3 MB): DMA into a pinned bounce buffer, then the helper
        // threads spread it into the caller's pageable memory
        KICP_TRY(pipe_out_stage(p, bytes));
        KICP_TRY(pipe_reserve_staging(p));  // (creates the helper threads)
        KICP_HIP(hipMemcpyAsync(p->out_stage, b.p, bytes, hipMemcpyDeviceToHost, p->stream));
        KICP_TRY(wait_stream(p->stream, "output download"));
        pipe_spread_copy(p, out, p->out_stage, bytes);
    }
    return KICP_OK;
}

// RegisterFrame as the reference declares it (KissICP.cpp:35-68): blocks, and hands back BOTH clouds.  The
// preprocessed frame -- the big one, as large as the scan -- is final long before the pose is: it leaves the device
// (DMA into pinned memory on a third stream) and reaches the caller's buffer while the registration is still
// running, so the call costs what the registration costs, not that plus a 3 MB download.
// views: the clouds stay in the pipeline's pinned buffer and the caller gets pointers into it.
// (enqueue = false: the frame is the one the caller has queued itself, a moment ago, through the asynchronous entry -- and
// nothing else: kicp_pipeline_collect_outputs)
static int pipe_register_outputs(kicp_pipeline *p, const double *xyz, size_t n, const double *timestamps, size_t n_ts, bool views,
                                 double *pre_out, size_t pre_cap, const double **pre_view, size_t *n_pre, double *src_out,
                                 size_t src_cap, const double **src_view, size_t *n_src, bool enqueue = true) {
    KICP_HIP(hipSetDevice(p->device));
    static const double kNone[3] = {0.0, 0.0, 0.0};
    if (enqueue) {
        if (p->in_flight) KICP_TRY(pipe_sync(p, false));  // `pre` exists once: no frame may be queued behind this one
        KICP_TRY(pipe_stage_and_enqueue(p, xyz ? xyz : kNone, nullptr, n, timestamps, n_ts));
    } else {
        if (p->in_flight != 1) {
            set_error("kicp_pipeline_collect_outputs: exactly one frame must be in flight (%d are)", p->in_flight);
            return KICP_ERR_INVALID_ARG;
        }
        n = p->last_in.n;
    }
    const int par = (int)((p->frames_enqueued - 1) & 1u);
    if (!p->copy_stream) {
        KICP_HIP(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
        stream_register(p->device, p->copy_stream);
    }
    // everything the front stages may have produced (at most n points) plus their counts, in one go: the counts are
    // only known on the device, and a second round trip to size the copy would cost more than the few bytes saved
    const size_t want = views ? n : (n < pre_cap ? n : pre_cap);
    const size_t bytes = want * 3 * sizeof(double);
    const size_t off_prep = (bytes + 63) & ~(size_t)63, off_src = off_prep + 256;
    // (the source cloud as a view needs room behind the counts whether or not the preprocessed frame is one)
    KICP_TRY(pipe_out_stage(p, off_src + ((views || src_view) ? n * 3 * sizeof(double) : 0)));
    PrepState *h_prep = reinterpret_cast<PrepState *>(p->out_stage + off_prep);
    KICP_HIP(hipStreamWaitEvent(p->copy_stream, p->ev_prep_done[par], 0));
    KICP_HIP(hipMemcpyAsync(h_prep, p->prep.as<PrepState>() + par, sizeof(PrepState), hipMemcpyDeviceToHost, p->copy_stream));
    if (bytes) KICP_HIP(hipMemcpyAsync(p->out_stage, p->pre.p, bytes, hipMemcpyDeviceToHost, p->copy_stream));
    KICP_TRY(wait_stream(p->copy_stream, "early download of the preprocessed frame"));
    const size_t got_pre = (size_t)h_prep->n_pre;
    const size_t c = got_pre < want ? got_pre : want;
    if (!views && c) pipe_spread_copy(p, pre_out, p->out_stage, c * 3 * sizeof(double));
    // ... and now the pose
    KICP_TRY(pipe_sync(p, true));
    *n_pre = (size_t)p->last.st.n_pre;
    if (*n_pre != got_pre) {  // (a replayed registration re-runs the front stages on the same scan: same counts)
        set_error("preprocessed count changed under the download (%zu vs %zu)", got_pre, *n_pre);
        return KICP_ERR_HIP;
    }
    if (!src_view) return kicp_pipeline_output(p, KICP_OUT_SOURCE, src_out, src_cap, n_src);
    if (pre_view) *pre_view = reinterpret_cast<const double *>(p->out_stage);
    *n_src = (size_t)p->last.st.n_src;
    *src_view = reinterpret_cast<const double *>(p->out_stage + off_src);
    if (*n_src) {
        KICP_HIP(hipMemcpyAsync(p->out_stage + off_src, p->src[par].p, *n_src * 3 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        KICP_TRY(wait_stream(p->stream, "download of the source cloud"));
    }
    return KICP_OK;
}

int kicp_pipeline_register_frame_outputs(kicp_pipeline *p, const double *xyz, size_t n, const double *timestamps, size_t n_ts,
                                         double *pre_out, size_t pre_cap, size_t *n_pre, double *src_out, size_t src_cap,
                                         size_t *n_src) {
    if (!p || (!xyz && n) || !n_pre || !n_src || (!pre_out && pre_cap) || (!src_out && src_cap)) return KICP_ERR_INVALID_ARG;
    return pipe_register_outputs(p, xyz, n, timestamps, n_ts, false, pre_out, pre_cap, nullptr, n_pre, src_out, src_cap, nullptr, n_src);
}

int kicp_pipeline_register_frame_views(kicp_pipeline *p, const double *xyz, size_t n, const double *timestamps, size_t n_ts,
                                       const double **pre_view, size_t *n_pre, const double **src_view, size_t *n_src) {
    if (!p || (!xyz && n) || !n_pre || !n_src || !pre_view || !src_view) return KICP_ERR_INVALID_ARG;
    return pipe_register_outputs(p, xyz, n, timestamps, n_ts, true, nullptr, 0, pre_view, n_pre, nullptr, 0, src_view, n_src);
}

int kicp_pipeline_collect_outputs(kicp_pipeline *p, double *pre_out, size_t pre_cap, size_t *n_pre, const double **src_view, size_t *n_src) {
    if (!p || !n_pre || !n_src || !src_view || (!pre_out && pre_cap)) return KICP_ERR_INVALID_ARG;
    return pipe_register_outputs(p, nullptr, 0, nullptr, 0, false, pre_out, pre_cap, nullptr, n_pre, nullptr, 0, src_view, n_src, false);
}

int kicp_pipeline_voxelize(kicp_pipeline *p, const double *xyz, size_t n, double *source_xyz, size_t *n_source,
                           double *fd_xyz, size_t *n_fd) {
    if (!p || !n_source || !n_fd) return KICP_ERR_INVALID_ARG;
    // KissICP.cpp:70-75
    KICP_TRY(kicp_voxel_downsample(xyz, n, p->cfg.voxel_size * 0.5, p->device, fd_xyz, n_fd));
    return kicp_voxel_downsample(fd_xyz, *n_fd, p->cfg.voxel_size * 1.5, p->device, source_xyz, n_source);
}

int kicp_pipeline_last_stats(kicp_pipeline *p, kicp_frame_stats *s) {
    if (!p || !s) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    memset(s, 0, sizeof *s);
    s->n_raw = p->last.n_raw;
    s->n_preprocessed = (uint64_t)h.n_pre;
    s->n_frame_downsample = (uint64_t)h.n_fd;
    s->n_source = (uint64_t)h.n_src;
    s->map_voxels = (uint64_t)p->last.map_ctr[C_LIVE];
    s->sigma = h.sigma;
    s->icp.iterations = h.icp_iterations;
    s->icp.converged = h.icp_converged;
    s->icp.n_source = (uint64_t)h.n_src;
    s->icp.n_corr_last = h.icp_ncorr_last;
    s->icp.points_examined = h.icp_examined;
    s->icp.n_corr_total = h.icp_ncorr_total;
    return KICP_OK;
}

int kicp_pipeline_icp_timing(kicp_pipeline *p, double *total_ms, uint64_t *launches, uint64_t *iterations,
                             uint64_t *bytes, int reset) {
    if (!p) return KICP_ERR_INVALID_ARG;
    if (p->in_flight) KICP_TRY(kicp_pipeline_sync(p));
    if (total_ms) *total_ms = p->icp_ms;
    if (launches) *launches = p->icp_launches;
    if (iterations) *iterations = p->icp_iters;
    if (bytes) *bytes = p->icp_bytes;
    if (reset) {
        p->icp_ms = 0.0;
        p->icp_launches = p->icp_iters = p->icp_bytes = 0;
    }
    return KICP_OK;
}

int kicp_pipeline_icp_profile(kicp_pipeline *p, uint64_t cycles[4], int *workgroups) {
    if (!p || !cycles) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    for (int i = 0; i < 4; ++i) cycles[i] = h.prof[i];
    if (workgroups) *workgroups = h.icp_blocks_used;
    return KICP_OK;
}

int kicp_pipeline_icp_clock(kicp_pipeline *p, uint64_t *cycles, uint64_t *ticks) {
    if (!p || !cycles || !ticks) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    *cycles = h.prof_clock[0];
    *ticks = h.prof_clock[1];
    return KICP_OK;
}

int kicp_pipeline_icp_first_iteration(kicp_pipeline *p, uint64_t *ticks_first, uint64_t *ticks_total, int *iterations) {
    if (!p || !ticks_first || !ticks_total || !iterations) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    *ticks_first = h.icp_iterations > 0 ? h.prof_it0_ticks : 0;
    *ticks_total = h.prof_clock[1];
    *iterations = h.icp_iterations;
    return KICP_OK;
}

int kicp_pipeline_icp_iteration_profile(kicp_pipeline *p, uint32_t *out, int cap_iters, int *n_iters) {
    if (!p || !n_iters || (!out && cap_iters > 0)) return KICP_ERR_INVALID_ARG;
    PipeState h;
    KICP_TRY(pipe_state_get(p, h));
    int n = h.icp_iterations < kIcpProfIters ? h.icp_iterations : kIcpProfIters;
    *n_iters = n;
    if (n > cap_iters) n = cap_iters;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 6; ++j) out[i * 6 + j] = h.prof_iter[i][j];
    return KICP_OK;
}

int kicp_pipeline_icp_group_profile(kicp_pipeline *p, uint32_t *out, size_t cap_words, int *n_iters, int *n_groups) {
    if (!p || !out || !n_iters || !n_groups) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(p->device));
    if (p->in_flight) KICP_TRY(kicp_pipeline_sync(p));
    *n_iters = *n_groups = 0;
    if (!p->have_last || !p->prof_groups.p) return KICP_OK;
    const int iters = p->last.st.icp_iterations < kIcpProfIters ? p->last.st.icp_iterations : kIcpProfIters;
    const int groups = p->last.st.icp_blocks_used * kIcpGroupsPerBlock;
    if ((size_t)iters * groups * 4 > cap_words) return KICP_ERR_INVALID_ARG;
    for (int it = 0; it < iters; ++it)
        KICP_HIP(hipMemcpy(out + (size_t)it * groups * 4,
                           p->prof_groups.as<unsigned>() + (size_t)it * kIcpMaxBlocks * kIcpGroupsPerBlock * 4,
                           (size_t)groups * 4 * sizeof(unsigned), hipMemcpyDeviceToHost));
    *n_iters = iters;
    *n_groups = groups;
    return KICP_OK;
}

int kicp_pipeline_host_stats(kicp_pipeline *p, kicp_host_stats *out, int reset) {
    if (!p) return KICP_ERR_INVALID_ARG;
    const kicp_map *m = p->map;
    if (out) {
        *out = p->hs;
        out->counter_refreshes = m->n_refresh - p->map_refresh0;
        out->map_grows = m->n_grow - p->map_grow0;
        out->map_rehashes = m->n_rehash - p->map_rehash0;
        out->wait_ms += m->wait_ms - p->map_wait0;
        out->device_numa_node = device_numa_node(p->device);
        out->staging_numa_node = p->stage[0] ? p->stage_node : -1;
        out->staging_helpers = p->pool ? p->pool->helpers() : -1;
        out->helpers_bound = p->pool ? p->pool->bound() : 0;
    }
    if (reset) {
        memset(&p->hs, 0, sizeof p->hs);
        p->map_refresh0 = m->n_refresh;
        p->map_grow0 = m->n_grow;
        p->map_rehash0 = m->n_rehash;
        p->map_wait0 = m->wait_ms;
    }
    return KICP_OK;
}

int kicp_pipeline_stream(kicp_pipeline *p, void **stream) {
    if (!p || !stream) return KICP_ERR_INVALID_ARG;
    *stream = (void *)p->stream;
    return KICP_OK;
}

/* poses of the frames completed by the most recent kicp_pipeline_sync (row-major 4x4 each) */
int kicp_pipeline_synced_poses(kicp_pipeline *p, double *T_out, size_t cap_frames, size_t *n_frames) {
    if (!p || !n_frames) return KICP_ERR_INVALID_ARG;
    const size_t nf = p->pending_poses.size() / 16;
    *n_frames = nf;
    const size_t c = nf < cap_frames ? nf : cap_frames;
    if (c && T_out) memcpy(T_out, p->pending_poses.data(), c * 16 * sizeof(double));
    return KICP_OK;
}

// ---- misc ---------------------------------------------------------------------------------------
const char *kicp_status_string(int s) {
    switch (s) {
        case KICP_OK: return "ok";
        case KICP_ERR_INVALID_ARG: return "invalid argument";
        case KICP_ERR_HIP: return "HIP runtime error";
        case KICP_ERR_OOM: return "out of memory";
        case KICP_ERR_CAPACITY: return "device table capacity exceeded";
        case KICP_ERR_RANGE: return "voxel coordinate out of range";
        case KICP_ERR_TIMEOUT: return "in-kernel wait timed out";
        case KICP_ERR_NO_DEVICE: return "no gfx950 device";
        case KICP_ERR_TIMESTAMPS: return "timestamps shorter than frame";
        default: return "unknown status";
    }
}
const char *kicp_last_error(void) { return get_error(); }
int kicp_version(int *major, int *minor) {
    if (major) *major = KICP_VERSION_MAJOR;
    if (minor) *minor = KICP_VERSION_MINOR;
    return KICP_OK;
}
int kicp_device_count(int *count) {
    if (!count) return KICP_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return KICP_OK;
}
int kicp_device_name(int device_id, char *buf, size_t len) {
    if (!buf || !len) return KICP_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    KICP_HIP(hipGetDeviceProperties(&prop, device_id));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return KICP_OK;
}
int kicp_device_alloc(int device_id, size_t bytes, void **d_ptr) {
    if (!d_ptr) return KICP_ERR_INVALID_ARG;
    KICP_TRY(check_device(device_id));
    KICP_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return KICP_OK;
}
int kicp_device_free(int device_id, void *d_ptr) {
    KICP_HIP(hipSetDevice(device_id));
    if (d_ptr) {
        KICP_TRY(wait_device(device_id, "kicp_device_free"));  // (hipFree waits for the whole device)
        KICP_HIP(hipFree(d_ptr));
    }
    return KICP_OK;
}
int kicp_device_upload(int device_id, void *d_dst, const void *h_src, size_t bytes) {
    if ((!d_dst || !h_src) && bytes) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(device_id));
    if (bytes) KICP_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return KICP_OK;
}
int kicp_device_download(int device_id, void *h_dst, const void *d_src, size_t bytes) {
    if ((!h_dst || !d_src) && bytes) return KICP_ERR_INVALID_ARG;
    KICP_HIP(hipSetDevice(device_id));
    if (bytes) KICP_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return KICP_OK;
}
int kicp_device_synchronize(int device_id) {
    KICP_HIP(hipSetDevice(device_id));
    return wait_device(device_id, "kicp_device_synchronize", true);  // (the library's own streams and the null stream, with the deadline)
}
// Host self-test: the staging path's float64 -> float32 narrowing (AVX2 or scalar, as the staging threads run it).
// *exact = 1 iff every value survives the round trip -- the only case in which a scan is uploaded as float32.
int kicp_selftest_narrow(const double *src, size_t count, float *dst, int *exact) {
    if ((!src || !dst) && count) return KICP_ERR_INVALID_ARG;
    if (!exact) return KICP_ERR_INVALID_ARG;
    *exact = narrow_exact(src, dst, count) ? 1 : 0;
    return KICP_OK;
}

int kicp_selftest_tile_sort(int device_id, const double *xyz, size_t n, double voxel_size, size_t n_hint, size_t n_bound, uint64_t *keys) {
    if ((n && !xyz) || (n && !keys) || !(voxel_size > 0.0) || n_bound < n || n_bound > ((size_t)1 << 24)) return KICP_ERR_INVALID_ARG;
    KICP_TRY(check_device(device_id));
    if (n == 0) return KICP_OK;
    DevBuf pts, a, b, cnt;
    int s = pts.reserve(n * 3 * sizeof(double));
    if (s == KICP_OK) s = a.reserve(n_bound * sizeof(unsigned long long));
    if (s == KICP_OK) s = b.reserve(n_bound * sizeof(unsigned long long));
    if (s == KICP_OK) s = cnt.reserve(sizeof(int));
    const int n_i = (int)n;
    if (s == KICP_OK && (hipMemcpy(pts.p, xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                         hipMemcpy(cnt.p, &n_i, sizeof n_i, hipMemcpyHostToDevice) != hipSuccess))
        s = KICP_ERR_HIP;
    // as the pipeline calls it: the count on the device, the host knows a bound and a hint
    hipStream_t us = util_stream(device_id);
    if (s == KICP_OK && (!us || launch_tile_sort(pts.as<double>(), cnt.as<int>(), 0, n_bound, voxel_size, a.as<unsigned long long>(), b.as<unsigned long long>(), n_hint, us) != 0))
        s = KICP_ERR_HIP;
    if (s == KICP_OK) s = wait_stream(us, "sort self-test");
    if (s == KICP_OK && hipMemcpy(keys, b.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) s = KICP_ERR_HIP;
    pts.release();
    a.release();
    b.release();
    cnt.release();
    return s;
}

int kicp_selftest_solve(int device_id, const double *A, const double *b, size_t n, double *x) {
    if ((!A || !b || !x) && n) return KICP_ERR_INVALID_ARG;
    KICP_TRY(check_device(device_id));
    if (n == 0) return KICP_OK;
    DevBuf dA, db, dx;
    ScopedBufs sb;
    sb.v = {&dA, &db, &dx};
    KICP_TRY(dA.reserve(n * 36 * sizeof(double)));
    KICP_TRY(db.reserve(n * 6 * sizeof(double)));
    KICP_TRY(dx.reserve(n * 6 * sizeof(double)));
    KICP_HIP(hipMemcpy(dA.p, A, n * 36 * sizeof(double), hipMemcpyHostToDevice));
    KICP_HIP(hipMemcpy(db.p, b, n * 6 * sizeof(double), hipMemcpyHostToDevice));
    hipStream_t us = util_stream(device_id);
    if (!us) return KICP_ERR_HIP;
    launch_selftest_solve(dA.as<double>(), db.as<double>(), (int)n, dx.as<double>(), us);
    KICP_HIP(hipGetLastError());
    KICP_TRY(wait_stream(us, "solve self-test"));
    KICP_HIP(hipMemcpy(x, dx.p, n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    return KICP_OK;
}

int kicp_set_option(const char *name, long value) {
    if (!name) return KICP_ERR_INVALID_ARG;
    if (!strcmp(name, "icp_blocks")) {
        if (value < 0 || value > kIcpMaxBlocks) return KICP_ERR_INVALID_ARG;
        options().icp_blocks = value;
    } else if (!strcmp(name, "icp_points_per_group")) {
        if (value < 1 || value > 1024) return KICP_ERR_INVALID_ARG;
        options().icp_points_per_group = value;
    } else if (!strcmp(name, "icp_use_lds")) {
        options().icp_use_lds = value;
    } else if (!strcmp(name, "icp_bulk_fill")) {
        options().icp_bulk_fill = value;
    } else if (!strcmp(name, "icp_wide")) {
        if (value < -1 || value > 1) return KICP_ERR_INVALID_ARG;
        options().icp_wide = value;
    } else if (!strcmp(name, "icp_device_streams")) {
        if (value < 1 || value > kIcpMaxLanes) return KICP_ERR_INVALID_ARG;
        options().icp_device_streams = value;
    } else if (!strcmp(name, "icp_wide_promote_from")) {
        if (value < 0 || value > 1000) return KICP_ERR_INVALID_ARG;
        options().icp_wide_promote_from = value;
    } else if (!strcmp(name, "icp_wide_load_eighths")) {
        if (value < 2 || value > 7) return KICP_ERR_INVALID_ARG;
        options().icp_wide_load_eighths = value;
    } else if (!strcmp(name, "icp_wide_stable")) {
        options().icp_wide_stable = value != 0;
    } else if (!strcmp(name, "icp_group_stable")) {
        options().icp_group_stable = value != 0;
    } else if (!strcmp(name, "icp_weights_kernel")) {
        options().icp_weights_kernel = value != 0;
    } else if (!strcmp(name, "sort_by_rank")) {
        options().sort_by_rank = value != 0;
    } else if (!strcmp(name, "map_fused_update")) {
        options().map_fused_update = value != 0;
    } else if (!strcmp(name, "icp_wide_per_round")) {
        if (value < 1 || value > 27) return KICP_ERR_INVALID_ARG;
        options().icp_wide_per_round = value;
    } else if (!strcmp(name, "frame_events")) {
        options().frame_events = value != 0;
    } else if (!strcmp(name, "icp_wide_group_max")) {
        if (value < 0 || value > 512) return KICP_ERR_INVALID_ARG;
        options().icp_wide_group_max = value;
    } else if (!strcmp(name, "icp_wide_flat")) {
        if (value < 0 || value > 3) return KICP_ERR_INVALID_ARG;
        options().icp_wide_flat = value;
    } else if (!strcmp(name, "icp_wide_prefill")) {
        if (value < 0 || value > 8) return KICP_ERR_INVALID_ARG;
        options().icp_wide_prefill = value;
    } else if (!strcmp(name, "icp_schur_solve")) {
        options().icp_schur_solve = value != 0;
    } else if (!strcmp(name, "icp_wide_prune")) {
        if (value < 0 || value > 2) return KICP_ERR_INVALID_ARG;
        options().icp_wide_prune = value;
    } else if (!strcmp(name, "icp_profile")) {
        options().icp_profile = value;
    } else if (!strcmp(name, "icp_timing")) {
        options().icp_timing = value;
    } else if (!strcmp(name, "icp_lds_kib")) {
        if (value != 0 && (value < 96 || value > 160)) return KICP_ERR_INVALID_ARG;  // below ~88 KiB the layout has no room for a tile
        options().icp_lds_kib = value;
    } else if (!strcmp(name, "icp_reserve_cus")) {
        if (value < 0 || value > 128) return KICP_ERR_INVALID_ARG;
        options().icp_reserve_cus = value;
    } else if (!strcmp(name, "staging_threads")) {
        if (value < 0 || value > 64) return KICP_ERR_INVALID_ARG;
        options().staging_threads = value;
    } else if (!strcmp(name, "staging_f32")) {
        options().staging_f32 = value;
    } else if (!strcmp(name, "downsample_order")) {
        if (value != 0 && value != 1) return KICP_ERR_INVALID_ARG;
        options().downsample_order = value;
    } else if (!strcmp(name, "staging_zero_copy")) {
        options().staging_zero_copy = value;
    } else if (!strcmp(name, "stage_in")) {
        options().stage_in = value != 0;
    } else if (!strcmp(name, "staging_numa")) {
        if (value < 0 || value > 2) return KICP_ERR_INVALID_ARG;
        options().staging_numa = value;
    } else if (!strcmp(name, "staging_numa_pretend")) {
        if (value < -1 || value > 1023) return KICP_ERR_INVALID_ARG;
        options().staging_numa_pretend = value;
    } else if (!strcmp(name, "relaxed_backpressure")) {
        options().relaxed_backpressure = value != 0;
    } else if (!strcmp(name, "collective_timeout_ms")) {
        if (value < 0) return KICP_ERR_INVALID_ARG;
        options().collective_timeout_ms = value;
    } else if (!strcmp(name, "icp_weight_base")) {
        if (value < 1 || value > 1024) return KICP_ERR_INVALID_ARG;  // (a weight is a 32-bit granule; the prefix sums are 64-bit)
        options().icp_weight_base = value;
    } else if (!strcmp(name, "icp_weight_long_base")) {
        if (value < 1 || value > 4096) return KICP_ERR_INVALID_ARG;
        options().icp_weight_long_base = value;
    } else if (!strcmp(name, "icp_weight_long_emul")) {
        if (value < 0 || value > 64) return KICP_ERR_INVALID_ARG;
        options().icp_weight_long_emul = value;
    } else if (!strcmp(name, "icp_weight_dense_min")) {
        if (value < 0 || value > 100000) return KICP_ERR_INVALID_ARG;
        options().icp_weight_dense_min = value;
    } else if (!strcmp(name, "icp_weight_dense_div")) {
        if (value < 0 || value > 1024) return KICP_ERR_INVALID_ARG;
        options().icp_weight_dense_div = value;
    } else if (!strcmp(name, "icp_weight_quad")) {
        if (value < -1 || value > 4096) return KICP_ERR_INVALID_ARG;
        options().icp_weight_quad = value;
    } else if (!strcmp(name, "icp_inject_timeout")) {
        if (value < 0) return KICP_ERR_INVALID_ARG;
        options().icp_inject_timeout = value;
    } else if (!strcmp(name, "icp_inject_timeout_skip")) {
        if (value < 0) return KICP_ERR_INVALID_ARG;
        options().icp_inject_timeout_skip = value;
    } else if (!strcmp(name, "map_rehash_every")) {
        if (value < 0) return KICP_ERR_INVALID_ARG;
        options().map_rehash_every = value;
    } else if (!strcmp(name, "queue_depth")) {
        if (value != 0 && (value < 2 || value > kicp_pipeline::kRing - 1)) return KICP_ERR_INVALID_ARG;
        options().queue_depth = value;
    } else if (!strcmp(name, "wait_timeout_ms")) {
        if (value < 0) return KICP_ERR_INVALID_ARG;
        options().wait_timeout_ms = value;
    } else if (!strcmp(name, "inject_stall_ms")) {
        if (value < 0 || value > 60000) return KICP_ERR_INVALID_ARG;
        options().inject_stall_ms = value;
    } else if (!strcmp(name, "map_apply_threads")) {
        if (value != 256 && value != 512 && value != 1024) return KICP_ERR_INVALID_ARG;
        options().map_apply_threads = value;
    } else {
        set_error("unknown option '%s'", name);
        return KICP_ERR_INVALID_ARG;
    }
    return KICP_OK;
}

}  // extern "C"
