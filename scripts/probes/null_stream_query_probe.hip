// null_stream_query_probe.hip -- does hipStreamQuery(nullptr) == hipSuccess mean that what was queued on the NULL stream has run?
// Background (profiles/README.md, r05_e): a build that zero-filled new buffers with hipMemsetAsync(.., nullptr) and polled
// hipStreamQuery(nullptr) instead of hipStreamSynchronize(nullptr) died of a memory fault in its first kernels as the first
// process of a fresh box.  Here: queue a zero-fill on the null stream, poll the query until it says success, then overwrite the
// buffer with 0xFF on a non-blocking stream and wait for THAT; any zero byte found afterwards was written by the zero-fill
// AFTER the query had called it finished.  Run as the first process of a box: hipcc -O2 --offload-arch=gfx950 -o probe this.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int late_total = 0;
    for (int rep = 0; rep < 24; ++rep) {
        const size_t bytes = (size_t)(rep % 3 == 0 ? 256 : rep % 3 == 1 ? 8 : 1) << 20;
        unsigned char *p = nullptr;
        CK(hipMalloc((void **)&p, bytes));
        CK(hipMemsetAsync(p, 0, bytes, nullptr));
        long polls = 0;
        for (;;) {
            const hipError_t q = hipStreamQuery(nullptr);
            if (q == hipSuccess) break;
            (void)hipGetLastError();
            if (q != hipErrorNotReady) {
                std::printf("hipStreamQuery(nullptr): %s\n", hipGetErrorString(q));
                return 2;
            }
            ++polls;
        }
        CK(hipMemsetAsync(p, 0xFF, bytes, s));
        CK(hipStreamSynchronize(s));
        CK(hipDeviceSynchronize());
        std::vector<unsigned char> h(bytes);
        CK(hipMemcpy(h.data(), p, bytes, hipMemcpyDeviceToHost));
        size_t zeros = 0;
        for (size_t i = 0; i < bytes; ++i) zeros += h[i] == 0;
        std::printf("rep %2d  %4zu MiB  polls until 'success' %8ld  zero bytes after the 0xFF fill: %zu%s\n", rep, bytes >> 20, polls, zeros,
                    zeros ? "  <-- the zero-fill ran AFTER the query reported success" : "");
        late_total += zeros != 0;
        CK(hipFree(p));
    }
    std::printf("%s\n", late_total ? "hipStreamQuery(nullptr) is NOT a completion test" : "no late zero-fill observed");
    return late_total ? 1 : 0;
}
