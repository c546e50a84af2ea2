// Which HIP runtime call stalls for ~5 ms around the 10th frame of a fresh process?  (profiles/r03_c: hipMemcpyAsync H2D
// from pinned memory, once per process.)  Mimics the pipeline's per-frame call pattern on two streams and times every call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now_ms() { using namespace std::chrono; return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count(); }
__global__ void k_busy(double *p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double v = p[i]; for (int k = 0; k < iters; ++k) v = v * 1.0000001 + 1e-9; p[i] = v; }
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: plain pattern; 1: pre-warm with 64 tiny copies at start; 2: pre-warm with 64 full-size copies
    const int frames = 40, kernels_per_frame = 10;
    const size_t bytes = 1566648;  // one scan as float32
    hipStream_t main_s, prep_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&prep_s, hipStreamNonBlocking));
    char *pinned[4];
    for (auto &p : pinned) CK(hipHostMalloc((void **)&p, bytes));
    double *d_raw[2], *d_work;
    for (auto &p : d_raw) CK(hipMalloc((void **)&p, bytes));
    CK(hipMalloc((void **)&d_work, 1 << 20));
    hipEvent_t ev_h2d[4], ev_prep[2], ev_frame[64];
    for (hipEvent_t &e : ev_h2d) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t &e : ev_prep) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t &e : ev_frame) CK(hipEventCreate(&e));
    if (mode == 1 || mode == 2) {
        const double t0 = now_ms();
        for (int i = 0; i < 64; ++i) CK(hipMemcpyAsync(d_raw[i & 1], pinned[i & 3], mode == 1 ? 64 : bytes, hipMemcpyHostToDevice, prep_s));
        CK(hipStreamSynchronize(prep_s));
        printf("pre-warm (%s copies): %.3f ms\n", mode == 1 ? "tiny" : "full", now_ms() - t0);
    }
    for (int f = 0; f < frames; ++f) {
        const int par = f & 1;
        if (f >= 4) CK(hipEventSynchronize(ev_frame[f - 3]));   // queue depth ~4
        double t0 = now_ms();
        CK(hipMemcpyAsync(d_raw[par], pinned[f & 3], bytes, hipMemcpyHostToDevice, prep_s));
        double t1 = now_ms();
        CK(hipEventRecord(ev_h2d[f & 3], prep_s));
        double t2 = now_ms();
        for (int k = 0; k < kernels_per_frame; ++k) hipLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, prep_s, d_raw[par], 32768, 200);
        CK(hipEventRecord(ev_prep[par], prep_s));
        CK(hipStreamWaitEvent(main_s, ev_prep[par], 0));
        CK(hipEventRecord(ev_frame[f], main_s));
        hipLaunchKernelGGL(k_busy, dim3(224), dim3(512), 0, main_s, d_work, 100000, 60000);  // ~0.3 ms "registration"
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, main_s, d_work, 32768, 200);
        double t3 = now_ms();
        if (t3 - t0 > 0.5 || f < 3) printf("frame %2d: memcpyAsync %.3f ms, eventRecord %.3f ms, launches %.3f ms\n", f, t1 - t0, t2 - t1, t3 - t2);
    }
    CK(hipDeviceSynchronize());
    printf("done mode %d\n", mode);
    return 0;
}
