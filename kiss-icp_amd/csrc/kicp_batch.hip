// kicp_batch.hip -- C-ABI of the multi-stream batch mode (include/kicp.h, "batch" section): the generic driver of
// kicp_batch.hpp over HIP pipelines, with the pose exchange done by RCCL called directly (librccl resolved with
// dlopen, so libkicp.so itself does not depend on it) or by a communicator the host supplies.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <new>

#include "kicp_batch.hpp"
#include "kicp_internal.hpp"
#include "kicp_numa.hpp"

using namespace kicp;

namespace {

// ---- per-stream pipeline over the C-ABI of this library ----------------------------------------------------------
struct HipPipe {
    kicp_config cfg;
    kicp_pipeline *pipe = nullptr;
    int device = -1;
    int share = 1;  // streams of this batch on the pipeline's device
    void *d_send = nullptr, *d_recv = nullptr;
    double *h_send = nullptr, *h_recv = nullptr;  // pinned
    size_t block_bytes = 0;
    int n_total = 0;
    hipStream_t xchg = nullptr;
    std::string err;

    int note(int rc) {
        if (rc != KICP_OK) err = kicp_last_error();
        return rc;
    }
    int hip(hipError_t e, const char *what) {
        if (e == hipSuccess) return KICP_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return KICP_ERR_HIP;
    }
    int open(int /*rank*/, int dev, size_t bytes, int total) {
        device = dev;
        block_bytes = bytes;
        n_total = total;
        // (this is the stream's worker thread: it copies every scan of its stream into pinned memory -- on the CPUs next to its GPU)
        if (options().staging_numa >= 2) (void)numa::bind_thread_to_node(pthread_self(), device_numa_node(dev));
        int rc = note(kicp::pipeline_create_shared(&cfg, dev, share, &pipe));
        if (rc != KICP_OK) return rc;
        if ((rc = hip(hipSetDevice(dev), "hipSetDevice")) != KICP_OK) return rc;
        if ((rc = hip(hipStreamCreateWithFlags(&xchg, hipStreamNonBlocking), "hipStreamCreate")) != KICP_OK) return rc;
        stream_register(dev, xchg);
        if ((rc = hip(hipMalloc(&d_send, bytes), "hipMalloc")) != KICP_OK) return rc;
        if ((rc = hip(hipMalloc(&d_recv, bytes * total), "hipMalloc")) != KICP_OK) return rc;
        if ((rc = hip(hipHostMalloc((void **)&h_send, bytes, hipHostMallocDefault), "hipHostMalloc")) != KICP_OK) return rc;
        return hip(hipHostMalloc((void **)&h_recv, bytes * total, hipHostMallocDefault), "hipHostMalloc");
    }
    int enqueue(const kicp_mstream::Frame &f) {
        if (f.xyz_f32)
            return note(kicp_pipeline_register_frame_async_f32(pipe, (const float *)f.xyz, f.n, f.timestamps, f.n_timestamps));
        return note(kicp_pipeline_register_frame_async(pipe, (const double *)f.xyz, f.n, f.timestamps, f.n_timestamps));
    }
    int sync() { return note(kicp_pipeline_sync(pipe)); }
    int new_poses(double *out, size_t cap, size_t *n) { return note(kicp_pipeline_synced_poses(pipe, out, cap, n)); }
    void *send_buffer() { return d_send; }
    void *recv_buffer() { return d_recv; }
    void *stream() { return (void *)xchg; }
    int put(const void *host, size_t bytes) {
        memcpy(h_send, host, bytes);
        return hip(hipMemcpyAsync(d_send, h_send, bytes, hipMemcpyHostToDevice, xchg), "hipMemcpyAsync(send)");
    }
    int get(void *host, size_t bytes) {
        int rc = hip(hipMemcpyAsync(h_recv, d_recv, bytes, hipMemcpyDeviceToHost, xchg), "hipMemcpyAsync(recv)");
        if (rc != KICP_OK) return rc;
        // (bounded, by the deadline of everything that waits for peers: one that never joins the collective must not hang this rank's
        // host.  Past it the collective and the copy STAY queued on xchg, over d_send / h_recv: the driver marks the batch broken
        // -- kicp_batch.hpp, post_all -- so that no later sync reuses them one exchange out of step with the peers.)
        if ((rc = wait_stream_peers(xchg, "pose exchange")) != KICP_OK) {
            err = kicp_last_error();
            return rc;
        }
        memcpy(host, h_recv, bytes);
        return KICP_OK;
    }
    void close() {
        if (device >= 0) (void)hipSetDevice(device);
        if (pipe) kicp_pipeline_destroy(pipe);
        pipe = nullptr;
        // (hipFree / hipHostFree wait for the whole device: only behind a bounded wait; a device that does not answer keeps them)
        const bool gone = (!xchg || wait_stream(xchg, "batch teardown") == KICP_OK) && (device < 0 || wait_device(device, "batch teardown") == KICP_OK);
        if (gone) {
            if (d_send) (void)hipFree(d_send);
            if (d_recv) (void)hipFree(d_recv);
            if (h_send) (void)hipHostFree(h_send);
            if (h_recv) (void)hipHostFree(h_recv);
        }
        if (xchg && gone) (void)stream_destroy(xchg);  // (not gone: leaked with its buffers -- a second full wait would follow)
        d_send = d_recv = nullptr;
        h_send = h_recv = nullptr;
        xchg = nullptr;
    }
    const char *last_error() const { return err.c_str(); }
    const char *thread_error() const { return kicp_last_error(); }
};

// ---- RCCL, called directly ------------------------------------------------------------------------------------------
// The few declarations of <rccl/rccl.h> this needs, restated so that the header is not a build dependency.
struct RcclId {
    char internal[128];  // NCCL_UNIQUE_ID_BYTES
};
typedef void *RcclComm_t;
enum { kRcclChar = 0 };  // ncclInt8 / ncclChar

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(RcclComm_t *, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, RcclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    std::string err;

    // librccl.so is 570 MB of code objects, and a machine that has just booted from an image faults them in page by page
    // while ncclCommInitRank walks them: 70 s and 290 s measured for the first process of two fresh MI355X boxes
    // (profiles/r05_fresh_lease_loop.txt, run r), against 12 s for the same bytes read front to back.  So: read it front to
    // back once before the loader maps it (0.1 s when it is cached already; KICP_NO_PREFETCH=1 skips this).
    static void stream_into_page_cache(const char *path) {
        if (getenv("KICP_NO_PREFETCH")) return;
        const int fd = open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return;
        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
        std::vector<char> buf((size_t)4 << 20);
        while (read(fd, buf.data(), buf.size()) > 0) {
        }
        close(fd);
    }

    bool load() {
        if (lib) return true;
        // The ROCm installation's own library, by PATH, before the bare name: a process that has imported torch
        // already holds torch's bundled librccl under the same soname, built against another HIP runtime than the
        // one libkicp.so runs on -- a dlopen by soname would hand that one back (seen: ncclCommInitRank failing
        // in a pytest process after `import torch`).
        std::vector<std::string> names;
        if (const char *root = getenv("ROCM_PATH")) names.push_back(std::string(root) + "/lib/librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const std::string &name : names) {
            if (name[0] == '/') stream_into_page_cache(name.c_str());
            lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            err = std::string("librccl not found: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) {
            void *s = dlsym(lib, n);
            if (!s) err = std::string("librccl lacks ") + n;
            return s;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        GetVersion = (decltype(GetVersion))sym("ncclGetVersion");
        return GetUniqueId && CommInitRank && CommDestroy && AllGather && GetErrorString;
    }
};

RcclApi &rccl() {
    static RcclApi api;
    return api;
}
std::mutex g_rccl_mutex;

struct RcclCtx {
    RcclId id;
    int first_rank = 0;
    std::vector<RcclComm_t> comms;  // one per local rank
};

int rccl_fail(int rc, const char *what) {
    set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    return KICP_ERR_HIP;
}

int rccl_init(void *ctx, int rank, int n_ranks, int device) {
    RcclCtx *c = (RcclCtx *)ctx;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return KICP_ERR_HIP;
    }
    // every rank calls ncclCommInitRank from its own thread; the call returns once all n_ranks have joined
    int rc = rccl().CommInitRank(&c->comms[rank - c->first_rank], n_ranks, c->id, rank);
    return rc ? rccl_fail(rc, "ncclCommInitRank") : KICP_OK;
}

int rccl_all_gather(void *ctx, int rank, const void *d_send, void *d_recv, size_t bytes, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    int rc = rccl().AllGather(d_send, d_recv, bytes, kRcclChar, c->comms[rank - c->first_rank], (hipStream_t)stream);
    return rc ? rccl_fail(rc, "ncclAllGather") : KICP_OK;
}

int rccl_finalize(void *ctx, int rank) {
    RcclCtx *c = (RcclCtx *)ctx;
    RcclComm_t &cm = c->comms[rank - c->first_rank];
    if (cm) rccl().CommDestroy(cm);
    cm = nullptr;
    return KICP_OK;
}

}  // namespace

struct kicp_batch {
    std::unique_ptr<kicp_mstream::Driver<HipPipe>> driver;
    RcclCtx rccl_ctx;
    bool own_comm = false;
    std::vector<kicp_mstream::Frame> frames;
    double gather_seconds = 0.0;
};

extern "C" {

int kicp_batch_unique_id(unsigned char id[KICP_BATCH_ID_BYTES]) {
    if (!id) return KICP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (!rccl().load()) {
        set_error("%s", rccl().err.c_str());
        return KICP_ERR_NO_DEVICE;
    }
    RcclId rid;
    int rc = rccl().GetUniqueId(&rid);
    if (rc) return rccl_fail(rc, "ncclGetUniqueId");
    static_assert(sizeof(RcclId) == KICP_BATCH_ID_BYTES, "id size");
    memcpy(id, &rid, sizeof rid);
    return KICP_OK;
}

int kicp_batch_create(const kicp_config *cfg, const int *devices, int n_local, int first_rank, int n_total,
                      const unsigned char *unique_id, const kicp_batch_comm *comm, size_t frames_per_gather,
                      kicp_batch **out) {
    if (!cfg || !devices || !out || n_local <= 0 || first_rank < 0 || n_total < first_rank + n_local) {
        set_error("kicp_batch_create: bad arguments");
        return KICP_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (frames_per_gather == 0) frames_per_gather = 64;
    if (frames_per_gather > 4096) return KICP_ERR_INVALID_ARG;
    kicp_batch *b = new (std::nothrow) kicp_batch();
    if (!b) return KICP_ERR_OOM;
    kicp_batch_comm table;
    if (comm) {
        if (!comm->all_gather) {
            delete b;
            set_error("kicp_batch_create: communicator without all_gather");
            return KICP_ERR_INVALID_ARG;
        }
        table = *comm;
    } else {
        {
            std::lock_guard<std::mutex> lk(g_rccl_mutex);
            if (!rccl().load()) {
                set_error("%s", rccl().err.c_str());
                delete b;
                return KICP_ERR_NO_DEVICE;
            }
        }
        if (unique_id) {
            memcpy(&b->rccl_ctx.id, unique_id, sizeof(RcclId));
        } else if (n_local == n_total) {
            int rc = rccl().GetUniqueId(&b->rccl_ctx.id);
            if (rc) {
                delete b;
                return rccl_fail(rc, "ncclGetUniqueId");
            }
        } else {
            delete b;
            set_error("kicp_batch_create: ranks in several processes need the unique id of kicp_batch_unique_id()");
            return KICP_ERR_INVALID_ARG;
        }
        b->rccl_ctx.first_rank = first_rank;
        b->rccl_ctx.comms.assign(n_local, nullptr);
        table = kicp_batch_comm{&b->rccl_ctx, rccl_init, rccl_all_gather, rccl_finalize};
        b->own_comm = true;
    }
    b->frames.resize(n_local);
    b->driver.reset(new kicp_mstream::Driver<HipPipe>(n_local, first_rank, n_total, frames_per_gather, table, options().collective_timeout_ms));
    const kicp_config c = *cfg;
    // streams that share a GPU share its persistent registration grid: each pipeline is created with 1 / n of it, and
    // the device's gate lets n registrations run side by side (option "icp_device_streams", kicp_api.hip)
    int rc = b->driver->start(devices, [&](int i) {
        auto p = std::make_unique<HipPipe>();
        p->cfg = c;
        int same = 0;  // per device: a device that carries one stream of this batch gives it its whole grid
        for (int j = 0; j < n_local; ++j) same += devices[j] == devices[i];
        p->share = same < 8 ? same : 8;
        return p;
    });
    if (rc != KICP_OK) {
        set_error("%s", b->driver->last_error().c_str());
        if (b->driver->broken()) return rc;  // (a worker may still be inside the rendezvous and come back into *b: leaked)
        delete b;
        return rc;
    }
    *out = b;
    return KICP_OK;
}

int kicp_batch_destroy(kicp_batch *b) {
    if (!b) return KICP_OK;
    if (b->driver && b->driver->broken()) {
        // an exchange was given up (collective_timeout_ms): the workers that are still inside it cannot be cancelled and may return
        // into the handle at any time -- the responsive ones are shut down, the handle itself is leaked
        b->driver->stop();
        set_error("kicp_batch_destroy: the batch had been given up inside an exchange; its handle is leaked");
        return KICP_ERR_TIMEOUT;
    }
    b->driver.reset();
    delete b;
    return KICP_OK;
}

static int batch_register(kicp_batch *b, const void *const *xyz, int f32, const size_t *n, const double *const *ts,
                          const size_t *n_ts) {
    if (!b || !xyz || !n) return KICP_ERR_INVALID_ARG;
    for (int i = 0; i < b->driver->n_local(); ++i) {
        kicp_mstream::Frame &f = b->frames[i];
        f.xyz = xyz[i];
        f.xyz_f32 = f32;
        f.n = n[i];
        f.timestamps = ts ? ts[i] : nullptr;
        f.n_timestamps = (ts && n_ts) ? n_ts[i] : 0;
        f.skip = xyz[i] == nullptr;  // a stream without a frame in this round
    }
    int rc = b->driver->register_frames(b->frames.data());
    if (rc != KICP_OK) set_error("%s", b->driver->last_error().c_str());
    return rc;
}

int kicp_batch_register_frames(kicp_batch *b, const double *const *xyz, const size_t *n, const double *const *timestamps,
                               const size_t *n_timestamps) {
    return batch_register(b, (const void *const *)xyz, 0, n, timestamps, n_timestamps);
}

int kicp_batch_register_frames_f32(kicp_batch *b, const float *const *xyz, const size_t *n, const double *const *timestamps,
                                   const size_t *n_timestamps) {
    return batch_register(b, (const void *const *)xyz, 1, n, timestamps, n_timestamps);
}

int kicp_batch_sync(kicp_batch *b) {
    if (!b) return KICP_ERR_INVALID_ARG;
    int rc = b->driver->sync();
    if (rc != KICP_OK) set_error("%s", b->driver->last_error().c_str());
    return rc;
}

int kicp_batch_poses(kicp_batch *b, int rank, double *T_out, size_t cap_frames, size_t *n_frames) {
    if (!b || !n_frames || rank < 0 || rank >= b->driver->n_total()) return KICP_ERR_INVALID_ARG;
    if (b->driver->syncs() == 0) {
        *n_frames = 0;
        return KICP_OK;
    }
    const std::vector<double> &v = b->driver->poses(rank);
    *n_frames = v.size() / 16;
    const size_t c = std::min(*n_frames, cap_frames);
    if (c && T_out) memcpy(T_out, v.data(), c * 16 * sizeof(double));
    return KICP_OK;
}

int kicp_batch_pipeline(kicp_batch *b, int local_stream, kicp_pipeline **pipe) {
    if (!b || !pipe || local_stream < 0 || local_stream >= b->driver->n_local()) return KICP_ERR_INVALID_ARG;
    *pipe = b->driver->pipe(local_stream).pipe;
    return KICP_OK;
}

int kicp_batch_gather_seconds(kicp_batch *b, double *seconds) {
    if (!b || !seconds) return KICP_ERR_INVALID_ARG;
    *seconds = b->driver->last_gather_seconds();
    return KICP_OK;
}

}  // extern "C"
