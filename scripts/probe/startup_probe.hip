// Measurement aid (not product code): the start-up cost of the HIP runtime in a fresh process — first call, first allocation,
// first (null-)stream operation, first kernel launch, a large device and a pinned host allocation.  On the pool's boxes: first call
// 0.05 s (0.08-0.19 s in one process out of four, whatever the environment), first stream operation 0.02 s, the rest ~0; none of
// GPU_MAX_HW_QUEUES, HSA_ENABLE_INTERRUPT=0, HSA_ENABLE_SDMA=0, HIP_INITIAL_DM_SIZE=0, ROCR_VISIBLE_DEVICES changes it (HISTORY §10).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(unsigned* p) { p[threadIdx.x] = threadIdx.x; }
int main() {
    double t = now_s(), t0 = t;
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    printf("first call %.3f", now_s() - t); t = now_s();
    unsigned* p = nullptr; (void)hipMalloc((void**)&p, 256);
    printf(" | first hipMalloc %.3f", now_s() - t); t = now_s();
    (void)hipMemsetAsync(p, 0, 256, nullptr); (void)hipStreamSynchronize(nullptr);
    printf(" | first null-stream op %.3f", now_s() - t); t = now_s();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, nullptr, p); (void)hipStreamSynchronize(nullptr);
    printf(" | first launch %.3f", now_s() - t); t = now_s();
    void* big = nullptr; (void)hipMalloc(&big, 6ull << 30);
    printf(" | hipMalloc 6 GiB %.3f", now_s() - t); t = now_s();
    void* pin = nullptr; (void)hipHostMalloc(&pin, 48u << 20, hipHostMallocDefault);
    printf(" | hipHostMalloc 48 MiB %.3f | total %.3f\n", now_s() - t, now_s() - t0);
    return 0;
}
