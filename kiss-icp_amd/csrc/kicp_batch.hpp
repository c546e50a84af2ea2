// kicp_batch.hpp -- multi-stream batch mode, host side: S independent LiDAR streams, one pipeline and one local
// map per GPU, ONE WORKER THREAD PER STREAM (each bound to its device), and after every batch of frames one
// all-gather of the new poses so that every rank holds all S trajectories ("pose-graph sync").
//
// A stream cannot be sharded over frames (frame k needs pose k-1 and the map holding frame k-1:
// cpp/kiss_icp/pipeline/KissICP.cpp:47,61), so streams are the unit of parallelism and the data path needs no
// collective.  The reference has no such mode.
//
// This header is plain host C++ (no HIP): the driver is a template over the per-stream pipeline type so that the
// orchestration -- worker threads, ragged batches, chunked gathers, error propagation -- is the same code whether
// it drives HIP pipelines with RCCL (kicp_batch.hip) or stand-ins in a CPU test (tests/cpp/test_batch_stub.cpp).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kicp.h"

namespace kicp_mstream {

struct Frame {
    const void *xyz = nullptr;  // n x 3, float64 or float32 (xyz_f32)
    int xyz_f32 = 0;
    size_t n = 0;
    const double *timestamps = nullptr;
    size_t n_timestamps = 0;
    bool skip = false;  // this stream has no frame in this round
};

// What the driver needs from a per-stream pipeline (all called on the stream's own worker thread):
//   int  open(int rank, int device)                 bind the thread to the device, create the pipeline and two
//                                                   exchange buffers of block_bytes / n_total * block_bytes
//   int  enqueue(const Frame&)                      queue one frame, do not wait for it
//   int  sync()                                     wait for everything queued
//   int  new_poses(double*, size_t cap, size_t *n)  poses completed by the last sync, oldest first
//   void *send_buffer(), *recv_buffer(), *stream()  exchange buffers (memory the communicator can address)
//   int  put(const void *host, size_t bytes)        host -> send buffer, ordered on stream()
//   int  get(void *host, size_t bytes)              recv buffer -> host, after the stream drained
//   void close()
//   const char *last_error(), *thread_error()      text of the pipeline's last failure / of the calling thread's (what a
//                                                   communicator running on this thread reported)
// Status codes are kicp_status; the text of a failure is fetched with last_error().

// One block per rank and gather: a four-word header -- poses in this block, the rank, poses the rank completed in this
// sync altogether, the status of the rank's own pipeline sync -- and `cap` poses.  Fixed size, and everything a rank
// needs to know about the others travels IN the blocks: ranks in different processes need not agree on anything but
// the number of sync calls (the number of gather rounds is the maximum over all ranks, a rank whose pipeline failed
// still takes part -- with an empty block and its status -- instead of leaving the others inside the collective).
constexpr size_t kBlockHeader = 4;
inline size_t block_doubles(size_t cap) { return kBlockHeader + 16 * cap; }

template <class Pipe>
class Driver {
public:
    // collective_timeout_ms > 0: a step that waits for PEERS -- the communicator's rendezvous, an all-gather -- is given up
    // after that long (KICP_ERR_TIMEOUT): a rank of another process that never arrives must not hold this one for ever.
    // The handle is then `broken`: its stuck threads cannot be cancelled, every later call says so, stop() lets go of them
    // and the OWNER must leak the object instead of deleting it (the threads may still return into it).
    Driver(int n_local, int first_rank, int n_total, size_t frames_per_gather, const kicp_batch_comm &comm, long collective_timeout_ms = 0)
        : n_local_(n_local), first_rank_(first_rank), n_total_(n_total), cap_(frames_per_gather), comm_(comm),
          workers_(n_local), timeout_ms_(collective_timeout_ms) {}

    ~Driver() { stop(); }

    template <class Make>
    int start(const int *devices, Make make_pipe) {
        for (int i = 0; i < n_local_; ++i) {
            Worker &w = workers_[i];
            w.index = i;
            w.device = devices[i];
            w.pipe = make_pipe(i);
            w.block.assign(block_doubles(cap_), 0.0);
            w.gathered.assign(block_doubles(cap_) * n_total_, 0.0);
            w.thread = std::thread([this, &w] { run(w); });
        }
        started_ = true;
        // pipelines first, the communicator only when every stream has one: a rendezvous that some rank never
        // joins would not return
        int rc = post_all(Cmd::Open);
        if (rc == KICP_OK) rc = post_all(Cmd::OpenComm);
        if (rc != KICP_OK) {
            const std::string keep = error_;
            stop();
            error_ = keep;
        }
        return rc;
    }

    // one frame (or none) per local stream; returns when every stream has taken its frame
    int register_frames(const Frame *frames) {
        if (!started_) return fail(KICP_ERR_INVALID_ARG, "batch not started");
        if (broken_) return fail_broken();  // (nothing of a worker is touched any more: some are still inside the exchange)
        for (int i = 0; i < n_local_; ++i) workers_[i].frame = frames[i];
        return post_all(Cmd::Enqueue);
    }

    // wait for every queued frame of every local stream, then exchange the new poses.  Afterwards
    // poses(rank) holds what global rank `rank` completed since the previous sync.
    int sync() {
        if (!started_) return fail(KICP_ERR_INVALID_ARG, "batch not started");
        if (broken_) return fail_broken();
        // a stream whose pipeline fails here still goes through the exchange (an empty block carrying its status): its
        // peers -- possibly in other processes -- are already on their way into the collective
        const int rc_sync = post_all(Cmd::Sync);
        const std::string err_sync = error_;
        all_poses_.assign(n_total_, {});
        gather_seconds_ = 0.0;
        size_t rounds = 1;  // known after the first gather: the most frames any rank of any process completed
        int rc_remote = KICP_OK, bad_rank = -1;
        for (size_t r = 0; r < rounds; ++r) {
            for (auto &w : workers_) w.round = r;
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = post_all(Cmd::Gather);
            gather_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rc != KICP_OK) return rc;  // the exchange itself failed: nothing more can be agreed on
            // every local rank received the same blocks; rank 0 of this process is the one read out
            const Worker &w0 = workers_[0];
            for (int g = 0; g < n_total_; ++g) {
                const double *blk = w0.gathered.data() + size_t(g) * block_doubles(cap_);
                const size_t cnt = size_t(blk[0]), total = size_t(blk[2]);
                if (cnt > cap_) return fail(KICP_ERR_INVALID_ARG, "gathered block carries an impossible count");
                all_poses_[g].insert(all_poses_[g].end(), blk + kBlockHeader, blk + kBlockHeader + 16 * cnt);
                if (r == 0) {
                    rounds = std::max(rounds, (total + cap_ - 1) / cap_);
                    if (blk[3] != 0.0 && rc_remote == KICP_OK) {
                        rc_remote = int(blk[3]);
                        bad_rank = g;
                    }
                }
            }
        }
        ++syncs_;
        if (rc_sync != KICP_OK) {
            error_ = err_sync;
            return rc_sync;
        }
        if (rc_remote != KICP_OK) {
            error_ = "stream " + std::to_string(bad_rank) + " (another process): its pipeline failed in this sync";
            return rc_remote;
        }
        return KICP_OK;
    }

    int n_local() const { return n_local_; }
    int n_total() const { return n_total_; }
    int first_rank() const { return first_rank_; }
    size_t syncs() const { return syncs_; }
    const std::vector<double> &poses(int global_rank) const { return all_poses_[global_rank]; }
    // what local rank i itself received in the last gather round (the test checks all ranks agree)
    const std::vector<double> &received(int local) const { return workers_[local].gathered; }
    Pipe &pipe(int local) { return *workers_[local].pipe; }
    const std::string &last_error() const { return error_; }
    double last_gather_seconds() const { return gather_seconds_; }

    bool broken() const { return broken_; }

    void stop() {
        if (!started_) return;
        if (!broken_) post_all(Cmd::Close);
        for (auto &w : workers_) {
            bool busy;
            {
                std::lock_guard<std::mutex> lk(w.m);
                busy = w.busy;
                if (!busy) {
                    w.cmd = Cmd::Exit;
                    w.pending = true;
                }
            }
            w.cv.notify_one();
            if (!w.thread.joinable()) continue;
            if (busy)
                w.thread.detach();  // (broken: still inside a collective that may never return; the owner leaks *this)
            else
                w.thread.join();
        }
        started_ = false;
    }

private:
    enum class Cmd { None, Open, OpenComm, Enqueue, Sync, Gather, Close, Exit };

    struct Worker {
        int index = 0, device = 0;
        std::unique_ptr<Pipe> pipe;
        std::thread thread;
        std::mutex m;
        std::condition_variable cv;
        Cmd cmd = Cmd::None;
        bool pending = false, done = false;
        bool busy = false;             // between a command's post and its completion
        int rc = KICP_OK;
        std::string err;
        Frame frame;
        std::vector<double> fresh;     // poses completed by the last sync (16 doubles each)
        std::vector<double> block;     // what this rank contributes to one gather
        std::vector<double> gathered;  // n_total blocks
        size_t round = 0;
        int sync_rc = KICP_OK;         // status of this stream's own pipeline in the running sync
        bool open = false, comm_open = false;
    };

    int fail(int rc, const char *what) {
        error_ = what;
        return rc;
    }
    int fail_broken() { return fail(KICP_ERR_TIMEOUT, "the batch was given up after an exchange that did not return (a peer never arrived): destroy it"); }

    int post_all(Cmd c) {
        if (broken_) return fail_broken();
        for (auto &w : workers_) {
            {
                std::lock_guard<std::mutex> lk(w.m);
                w.cmd = c;
                w.pending = true;
                w.done = false;
                w.busy = true;
            }
            w.cv.notify_one();
        }
        // (only the steps that wait for peers: a pipeline's own waits have the library's deadline, wait_timeout_ms)
        const bool bounded = timeout_ms_ > 0 && (c == Cmd::OpenComm || c == Cmd::Gather);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(bounded ? timeout_ms_ : 0);
        int rc = KICP_OK;
        for (auto &w : workers_) {
            std::unique_lock<std::mutex> lk(w.m);
            if (bounded) {
                if (!w.cv.wait_until(lk, deadline, [&] { return w.done; })) {
                    broken_ = true;
                    if (rc == KICP_OK) {
                        rc = KICP_ERR_TIMEOUT;
                        error_ = "stream " + std::to_string(first_rank_ + w.index) + ": " + (c == Cmd::OpenComm ? "the communicator's rendezvous" : "the pose exchange") +
                                 " did not return within " + std::to_string(timeout_ms_) + " ms (collective_timeout_ms): a peer has not arrived";
                    }
                    continue;
                }
            } else {
                w.cv.wait(lk, [&] { return w.done; });
            }
            if (w.rc != KICP_OK && rc == KICP_OK) {
                rc = w.rc;
                error_ = "stream " + std::to_string(first_rank_ + w.index) + ": " + w.err;
            }
            // an exchange the worker itself gave up (a collective on its stream that did not end in time) is still queued over
            // the worker's buffers: reusing them would put this rank one exchange out of step with its peers
            if (c == Cmd::Gather && w.rc == KICP_ERR_TIMEOUT) broken_ = true;
        }
        return rc;
    }

    int step(Worker &w, Cmd c) {
        Pipe &p = *w.pipe;
        const int rank = first_rank_ + w.index;
        switch (c) {
        case Cmd::Open: {
            w.open = true;  // close() also releases what a failing open() left behind
            int rc = p.open(rank, w.device, block_doubles(cap_) * sizeof(double), n_total_);
            if (rc != KICP_OK) return rc;
            return KICP_OK;
        }
        case Cmd::OpenComm: {
            if (comm_.init) {
                int rc = comm_.init(comm_.ctx, rank, n_total_, w.device);
                if (rc != KICP_OK) {
                    w.err = std::string("communicator init failed: ") + p.thread_error();
                    return rc;
                }
            }
            w.comm_open = true;
            return KICP_OK;
        }
        case Cmd::Enqueue:
            return w.frame.skip ? KICP_OK : p.enqueue(w.frame);
        case Cmd::Sync: {
            w.fresh.clear();
            w.sync_rc = p.sync();
            if (w.sync_rc != KICP_OK) return w.sync_rc;
            size_t n = 0;
            int rc = p.new_poses(nullptr, 0, &n);
            if (rc == KICP_OK && n) {
                w.fresh.assign(16 * n, 0.0);
                rc = p.new_poses(w.fresh.data(), n, &n);
            }
            if (rc != KICP_OK) w.fresh.clear();
            w.sync_rc = rc;
            return rc;
        }
        case Cmd::Gather: {
            const size_t have = w.fresh.size() / 16;
            const size_t lo = std::min(have, w.round * cap_), hi = std::min(have, lo + cap_);
            std::fill(w.block.begin(), w.block.end(), std::numeric_limits<double>::quiet_NaN());
            w.block[0] = double(hi - lo);
            w.block[1] = double(rank);
            w.block[2] = double(have);
            w.block[3] = double(w.sync_rc);
            if (hi > lo) std::memcpy(w.block.data() + kBlockHeader, w.fresh.data() + 16 * lo, (hi - lo) * 16 * sizeof(double));
            const size_t bytes = w.block.size() * sizeof(double);
            int rc = p.put(w.block.data(), bytes);
            if (rc != KICP_OK) return rc;
            rc = comm_.all_gather(comm_.ctx, rank, p.send_buffer(), p.recv_buffer(), bytes, p.stream());
            if (rc != KICP_OK) {
                w.err = std::string("all_gather failed: ") + p.thread_error();
                return rc;
            }
            return p.get(w.gathered.data(), bytes * n_total_);
        }
        case Cmd::Close:
            if (w.comm_open && comm_.finalize) comm_.finalize(comm_.ctx, rank);
            w.comm_open = false;
            if (w.open) p.close();
            w.open = false;
            return KICP_OK;
        default:
            return KICP_OK;
        }
    }

    void run(Worker &w) {
        for (;;) {
            Cmd c;
            {
                std::unique_lock<std::mutex> lk(w.m);
                w.cv.wait(lk, [&] { return w.pending; });
                w.pending = false;
                c = w.cmd;
            }
            if (c == Cmd::Exit) return;
            w.err.clear();
            int rc = step(w, c);
            if (rc != KICP_OK && w.err.empty()) w.err = w.pipe->last_error();
            {
                std::lock_guard<std::mutex> lk(w.m);
                w.rc = rc;
                w.done = true;
                w.busy = false;
            }
            w.cv.notify_all();
        }
    }

    int n_local_, first_rank_, n_total_;
    size_t cap_;
    kicp_batch_comm comm_;
    std::vector<Worker> workers_;
    std::vector<std::vector<double>> all_poses_;
    std::string error_;
    bool started_ = false;
    long timeout_ms_ = 0;   // deadline of the steps that wait for peers (0: none)
    bool broken_ = false;   // such a step was given up: some worker may never come back
    size_t syncs_ = 0;
    double gather_seconds_ = 0.0;
};

}  // namespace kicp_mstream
