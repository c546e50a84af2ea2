// Multi-stream batch driver (kiss-icp_amd/csrc/kicp_batch.hpp) driven on the CPU with stand-in pipelines and a stand-in
// communicator: the same orchestration code the C-ABI instantiates over HIP pipelines and RCCL.  Checks the
// thread-per-stream workers, the rank-ordered gather, ragged and empty batches, batches longer than one block,
// ranks split over two "processes", and how failures of a pipeline or of the communicator surface.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>

#include "../../kiss-icp_amd/csrc/kicp_batch.hpp"

using namespace kicp_mstream;

static int g_fail = 0;
#define CHECK(c)                                                    \
    do {                                                            \
        if (!(c)) {                                                 \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); \
            ++g_fail;                                               \
        }                                                           \
    } while (0)

// pose k of stream r: recognisable numbers
static void fake_pose(int rank, size_t k, double *T) {
    for (int i = 0; i < 16; ++i) T[i] = 1000.0 * rank + double(k) + 0.01 * i;
}

struct StubPipe {
    int rank = -1, device = -1, n_total = 0;
    size_t block_bytes = 0;
    std::vector<unsigned char> send, recv;
    std::vector<double> queued_payload;  // first coordinate of each queued scan, to see the right scan reached the right stream
    size_t done = 0, queued = 0, reported = 0;
    std::thread::id thread;
    int fail_enqueue_at = -1, fail_open = 0, fail_sync_at = -1, syncs = 0;
    std::string err;

    int open(int r, int dev, size_t bytes, int total) {
        rank = r;
        device = dev;
        block_bytes = bytes;
        n_total = total;
        thread = std::this_thread::get_id();
        if (fail_open) {
            err = "no such device";
            return KICP_ERR_NO_DEVICE;
        }
        send.assign(bytes, 0);
        recv.assign(bytes * total, 0);
        return KICP_OK;
    }
    int enqueue(const Frame &f) {
        if (std::this_thread::get_id() != thread) return KICP_ERR_INVALID_ARG;  // a stream stays on its own thread
        if ((int)queued == fail_enqueue_at) {
            err = "scan rejected";
            return KICP_ERR_RANGE;
        }
        queued_payload.push_back(f.n ? (f.xyz_f32 ? double(((const float *)f.xyz)[0]) : ((const double *)f.xyz)[0]) : -1.0);
        ++queued;
        return KICP_OK;
    }
    int sync() {
        if (syncs++ == fail_sync_at) {
            err = "registration gave up";
            queued = done;  // the frames of this batch are lost
            reported = done;
            return KICP_ERR_TIMEOUT;
        }
        reported = done;
        done = queued;
        return KICP_OK;
    }
    int new_poses(double *out, size_t cap, size_t *n) {
        *n = done - reported;
        for (size_t k = 0; k < std::min(cap, *n) && out; ++k) fake_pose(rank, reported + k, out + 16 * k);
        return KICP_OK;
    }
    void *send_buffer() { return send.data(); }
    void *recv_buffer() { return recv.data(); }
    void *stream() { return this; }
    int put(const void *h, size_t bytes) {
        std::memcpy(send.data(), h, bytes);
        return KICP_OK;
    }
    int get(void *h, size_t bytes) {
        std::memcpy(h, recv.data(), bytes);
        return KICP_OK;
    }
    void close() { closed = true; }
    bool closed = false;
    const char *last_error() const { return err.c_str(); }
    const char *thread_error() const { return ""; }
};

// all ranks of all "processes" meet here: a counting barrier + a shared table of send pointers
struct StubComm {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived = 0, generation = 0;
    std::vector<const void *> sends;
    std::set<int> inited, finalized;
    int fail_init_rank = -1, fail_gather = 0;
    size_t gathers = 0;

    static int init(void *c, int rank, int n_ranks, int) {
        StubComm *s = (StubComm *)c;
        std::lock_guard<std::mutex> lk(s->m);
        if (rank == s->fail_init_rank) return KICP_ERR_HIP;
        s->n = n_ranks;
        s->sends.resize(n_ranks);
        s->inited.insert(rank);
        return KICP_OK;
    }
    void barrier(std::unique_lock<std::mutex> &lk) {
        int g = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
    static int all_gather(void *c, int rank, const void *send, void *recv, size_t bytes, void *) {
        StubComm *s = (StubComm *)c;
        std::unique_lock<std::mutex> lk(s->m);
        if (s->fail_gather) return KICP_ERR_HIP;
        s->sends[rank] = send;
        s->barrier(lk);
        for (int r = 0; r < s->n; ++r) std::memcpy((unsigned char *)recv + size_t(r) * bytes, s->sends[r], bytes);
        if (rank == 0) ++s->gathers;
        s->barrier(lk);  // nobody overwrites its send block before everyone has copied it
        return KICP_OK;
    }
    static int finalize(void *c, int rank) {
        StubComm *s = (StubComm *)c;
        std::lock_guard<std::mutex> lk(s->m);
        s->finalized.insert(rank);
        return KICP_OK;
    }
    kicp_batch_comm table() { return kicp_batch_comm{this, init, all_gather, finalize}; }
};

static std::unique_ptr<StubPipe> plain(int) { return std::make_unique<StubPipe>(); }

static void check_poses(Driver<StubPipe> &d, int rank, size_t first, size_t count) {
    const std::vector<double> &v = d.poses(rank);
    CHECK(v.size() == 16 * count);
    double T[16];
    for (size_t k = 0; k < count && v.size() == 16 * count; ++k) {
        fake_pose(rank, first + k, T);
        CHECK(std::memcmp(T, v.data() + 16 * k, sizeof T) == 0);
    }
}

static void test_one_process_four_streams() {
    StubComm comm;
    const int S = 4;
    const int devices[S] = {0, 1, 2, 3};
    Driver<StubPipe> d(S, 0, S, 8, comm.table());
    CHECK(d.start(devices, plain) == KICP_OK);
    CHECK(comm.inited.size() == 4);
    std::set<std::thread::id> threads;
    for (int i = 0; i < S; ++i) {
        threads.insert(d.pipe(i).thread);
        CHECK(d.pipe(i).device == i && d.pipe(i).rank == i);
    }
    CHECK(threads.size() == 4 && !threads.count(std::this_thread::get_id()));  // a worker thread per stream

    // three rounds, stream 2 sits out the second one (ragged batch)
    double scans[S][3];
    Frame f[S];
    for (int round = 0; round < 3; ++round) {
        for (int i = 0; i < S; ++i) {
            scans[i][0] = 10.0 * round + i;
            f[i] = Frame();
            f[i].xyz = scans[i];
            f[i].n = 1;
            f[i].skip = (round == 1 && i == 2);
        }
        CHECK(d.register_frames(f) == KICP_OK);
    }
    CHECK(d.sync() == KICP_OK);
    for (int i = 0; i < S; ++i) check_poses(d, i, 0, i == 2 ? 2 : 3);
    CHECK(d.pipe(1).queued_payload == (std::vector<double>{1.0, 11.0, 21.0}));
    CHECK(d.pipe(2).queued_payload == (std::vector<double>{2.0, 22.0}));
    // every rank received the same bytes
    for (int i = 1; i < S; ++i)  // (the padding is NaN: compare bytes)
        CHECK(std::memcmp(d.received(i).data(), d.received(0).data(), d.received(0).size() * sizeof(double)) == 0);
    // padding of a short block is NaN, never stale poses
    const double *blk2 = d.received(0).data() + 2 * block_doubles(8);
    CHECK(blk2[0] == 2.0 && blk2[1] == 2.0 && blk2[2] == 2.0 && blk2[3] == 0.0 && std::isnan(blk2[kBlockHeader + 16 * 2]));

    // an empty sync: one gather, nothing new anywhere
    size_t before = comm.gathers;
    CHECK(d.sync() == KICP_OK);
    CHECK(comm.gathers == before + 1);
    for (int i = 0; i < S; ++i) CHECK(d.poses(i).empty());

    // 19 frames with blocks of 8: three gathers, order kept
    float scan32[3] = {7.f, 0.f, 0.f};
    for (int k = 0; k < 19; ++k) {
        for (int i = 0; i < S; ++i) {
            f[i] = Frame();
            f[i].xyz = scan32;
            f[i].xyz_f32 = 1;
            f[i].n = 1;
            f[i].skip = (i == 3 && k >= 5);
        }
        CHECK(d.register_frames(f) == KICP_OK);
    }
    before = comm.gathers;
    CHECK(d.sync() == KICP_OK);
    CHECK(comm.gathers == before + 3);
    check_poses(d, 0, 3, 19);
    check_poses(d, 2, 2, 19);
    check_poses(d, 3, 3, 5);
    CHECK(d.pipe(0).queued_payload.back() == 7.0);
    d.stop();
    CHECK(comm.finalized.size() == 4);
    for (int i = 0; i < S; ++i) CHECK(d.pipe(i).closed);
}

static void test_two_processes() {
    // ranks {0,1} and {2,3,4} in two driver objects sharing one communicator, as two processes of a node would
    StubComm comm;
    const int dev_a[2] = {0, 1}, dev_b[3] = {0, 1, 2};
    Driver<StubPipe> a(2, 0, 5, 4, comm.table()), b(3, 2, 5, 4, comm.table());
    int rc_a = -1, rc_b = -1;
    std::thread ta([&] { rc_a = a.start(dev_a, plain); }), tb([&] { rc_b = b.start(dev_b, plain); });
    ta.join();
    tb.join();
    CHECK(rc_a == KICP_OK && rc_b == KICP_OK);
    double scan[3] = {1, 2, 3};
    auto drive = [&](Driver<StubPipe> &d, int frames, int *rc) {
        std::vector<Frame> f(d.n_local());
        for (auto &x : f) {
            x.xyz = scan;
            x.n = 1;
        }
        *rc = KICP_OK;
        for (int k = 0; k < frames && *rc == KICP_OK; ++k) *rc = d.register_frames(f.data());
        if (*rc == KICP_OK) *rc = d.sync();
    };
    std::thread t1([&] { drive(a, 6, &rc_a); }), t2([&] { drive(b, 6, &rc_b); });
    t1.join();
    t2.join();
    CHECK(rc_a == KICP_OK && rc_b == KICP_OK);
    for (int r = 0; r < 5; ++r) {
        check_poses(a, r, 0, 6);
        check_poses(b, r, 0, 6);
    }
    std::thread s1([&] { a.stop(); }), s2([&] { b.stop(); });
    s1.join();
    s2.join();
}

static void test_failures_surface() {
    {  // a pipeline that cannot open: start fails, names the stream, the communicator is never entered
        StubComm comm;
        const int devices[3] = {0, 1, 2};
        Driver<StubPipe> d(3, 0, 3, 4, comm.table());
        int rc = d.start(devices, [](int i) {
            auto p = std::make_unique<StubPipe>();
            p->fail_open = (i == 1);
            return p;
        });
        CHECK(rc == KICP_ERR_NO_DEVICE);
        CHECK(d.last_error().find("stream 1") != std::string::npos && d.last_error().find("no such device") != std::string::npos);
        CHECK(comm.inited.empty());
        CHECK(d.register_frames(nullptr) == KICP_ERR_INVALID_ARG);
    }
    {  // communicator init fails on one rank
        StubComm comm;
        comm.fail_init_rank = 2;
        const int devices[3] = {0, 1, 2};
        Driver<StubPipe> d(3, 0, 3, 4, comm.table());
        CHECK(d.start(devices, plain) == KICP_ERR_HIP);
        CHECK(d.last_error().find("communicator") != std::string::npos);
    }
    {  // a scan one stream rejects: the call reports it, the other streams keep their frames, later syncs work
        StubComm comm;
        const int devices[2] = {0, 0};
        Driver<StubPipe> d(2, 0, 2, 4, comm.table());
        CHECK(d.start(devices, [](int i) {
            auto p = std::make_unique<StubPipe>();
            if (i == 1) p->fail_enqueue_at = 1;
            return p;
        }) == KICP_OK);
        double scan[3] = {0, 0, 0};
        Frame f[2];
        for (auto &x : f) {
            x.xyz = scan;
            x.n = 1;
        }
        CHECK(d.register_frames(f) == KICP_OK);
        CHECK(d.register_frames(f) == KICP_ERR_RANGE);
        CHECK(d.last_error().find("stream 1: scan rejected") != std::string::npos);
        CHECK(d.sync() == KICP_OK);
        check_poses(d, 0, 0, 2);
        check_poses(d, 1, 0, 1);
        comm.fail_gather = 1;
        CHECK(d.sync() == KICP_ERR_HIP);
        CHECK(d.last_error().find("all_gather") != std::string::npos);
    }
}

static void test_a_failing_pipeline_does_not_strand_its_peers() {
    // two "processes" {0,1} and {2}; the pipeline of rank 1 fails its first sync.  Both processes return from sync()
    // (nobody is left inside the collective), both report the failure, the healthy ranks' poses are delivered, and the
    // next sync works.  The processes queue DIFFERENT numbers of frames: the round count travels in the blocks.
    StubComm comm;
    const int dev_a[2] = {0, 1}, dev_b[1] = {0};
    Driver<StubPipe> a(2, 0, 3, 4, comm.table()), b(1, 2, 3, 4, comm.table());
    int rc_a = -1, rc_b = -1;
    std::thread ta([&] {
        rc_a = a.start(dev_a, [](int i) {
            auto p = std::make_unique<StubPipe>();
            if (i == 1) p->fail_sync_at = 0;
            return p;
        });
    });
    std::thread tb([&] { rc_b = b.start(dev_b, plain); });
    ta.join();
    tb.join();
    CHECK(rc_a == KICP_OK && rc_b == KICP_OK);
    double scan[3] = {1, 2, 3};
    auto drive = [&](Driver<StubPipe> &d, int frames, int *rc) {
        std::vector<Frame> f(d.n_local());
        for (auto &x : f) {
            x.xyz = scan;
            x.n = 1;
        }
        *rc = KICP_OK;
        for (int k = 0; k < frames && *rc == KICP_OK; ++k) *rc = d.register_frames(f.data());
        if (*rc == KICP_OK) *rc = d.sync();
    };
    std::thread t1([&] { drive(a, 3, &rc_a); }), t2([&] { drive(b, 9, &rc_b); });  // 9 frames: three rounds of 4
    t1.join();
    t2.join();
    CHECK(rc_a == KICP_ERR_TIMEOUT && rc_b == KICP_ERR_TIMEOUT);
    CHECK(a.last_error().find("stream 1") != std::string::npos && a.last_error().find("gave up") != std::string::npos);
    CHECK(b.last_error().find("stream 1") != std::string::npos);
    check_poses(a, 0, 0, 3);
    check_poses(b, 0, 0, 3);
    CHECK(a.poses(1).empty() && b.poses(1).empty());
    check_poses(a, 2, 0, 9);
    check_poses(b, 2, 0, 9);
    std::thread t3([&] { drive(a, 2, &rc_a); }), t4([&] { drive(b, 2, &rc_b); });
    t3.join();
    t4.join();
    CHECK(rc_a == KICP_OK && rc_b == KICP_OK);
    check_poses(a, 0, 3, 2);
    check_poses(a, 1, 0, 2);
    check_poses(b, 2, 9, 2);
    std::thread s1([&] { a.stop(); }), s2([&] { b.stop(); });
    s1.join();
    s2.join();
}

// A peer that never arrives: two "processes" share a communicator, only one of them syncs.  With a deadline on the steps that wait
// for peers its sync comes back with KICP_ERR_TIMEOUT in time, every later call says the batch is broken, stop() returns although
// its workers are still inside the exchange (they are let go of, the object must outlive them: it is leaked here as the C-ABI
// leaks the handle), and the late peer's arrival afterwards finds valid memory.
static void test_a_peer_that_never_arrives() {
    StubComm *comm = new StubComm();  // (outlives the abandoned workers)
    const int dev_a[2] = {0, 1}, dev_b[1] = {0};
    auto *a = new Driver<StubPipe>(2, 0, 3, 4, comm->table(), 300);  // 300 ms for the peers
    Driver<StubPipe> b(1, 2, 3, 4, comm->table(), 0);
    int rc_a = -1, rc_b = -1;
    std::thread ta([&] { rc_a = a->start(dev_a, plain); }), tb([&] { rc_b = b.start(dev_b, plain); });
    ta.join();
    tb.join();
    CHECK(rc_a == KICP_OK && rc_b == KICP_OK);
    double scan[3] = {1, 2, 3};
    std::vector<Frame> f(2);
    for (auto &x : f) {
        x.xyz = scan;
        x.n = 1;
    }
    CHECK(a->register_frames(f.data()) == KICP_OK);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = a->sync();  // b never syncs: the gather's barrier never fills
    const double took = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(rc == KICP_ERR_TIMEOUT);
    CHECK(took > 0.25 && took < 1.5);
    CHECK(a->broken());
    CHECK(a->last_error().find("collective_timeout_ms") != std::string::npos);
    CHECK(a->register_frames(f.data()) == KICP_ERR_TIMEOUT && a->sync() == KICP_ERR_TIMEOUT);  // broken: says so, at once
    const auto t1 = std::chrono::steady_clock::now();
    a->stop();  // returns: the workers inside the exchange are let go of
    CHECK(std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < 1.0);
    // the late peer arrives after all: the abandoned workers finish their gather into memory that still exists
    std::vector<Frame> fb(1);
    fb[0].xyz = scan;
    fb[0].n = 1;
    CHECK(b.register_frames(fb.data()) == KICP_OK);
    CHECK(b.sync() == KICP_OK);
    check_poses(b, 2, 0, 1);
    check_poses(b, 0, 0, 1);  // (what a's streams had contributed before they were given up)
    b.stop();
    std::this_thread::sleep_for(std::chrono::milliseconds(50));  // (a's workers come home and go to sleep)
    // `a` and `comm` are leaked on purpose
}

int main() {
    test_a_peer_that_never_arrives();
    test_a_failing_pipeline_does_not_strand_its_peers();
    test_one_process_four_streams();
    test_two_processes();
    test_failures_surface();
    if (g_fail) {
        std::printf("%d check(s) failed\n", g_fail);
        return 1;
    }
    std::printf("batch driver: all checks passed\n");
    return 0;
}
