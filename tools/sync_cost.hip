// Host cost of cross-stream dependencies on this runtime: events vs stream memory operations vs a second
// submitting thread.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/sync_cost tools/sync_cost.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    float *a, *b;
    CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256));
    CK(hipMemset(a, 0, 256)); CK(hipMemset(b, 0, 256));
    const int n = 2000;
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, a);
    CK(hipDeviceSynchronize());
    double t = now();
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, a);
    double h = now() - t; CK(hipDeviceSynchronize());
    printf("launch                               host %6.2f us  drained %6.2f us\n", h / n, (now() - t) / n);
    t = now();
    for (int i = 0; i < n; i++) {
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, a);
        CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s2, b);
    }
    h = now() - t; CK(hipDeviceSynchronize());
    printf("launch,record,wait,launch(other)     host %6.2f us  drained %6.2f us\n", h / n, (now() - t) / n);
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue = %d\n", can);
    if (can) {
        uint32_t* sig;
        CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
        CK(hipStreamWriteValue32(s1, sig, 0, 0));
        CK(hipDeviceSynchronize());
        t = now();
        for (int i = 1; i <= n; i++) {
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, a);
            CK(hipStreamWriteValue32(s1, sig, (uint32_t)i, 0));
            CK(hipStreamWaitValue32(s2, sig, (uint32_t)i, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s2, b);
        }
        h = now() - t; CK(hipDeviceSynchronize());
        printf("launch,write32,wait32,launch(other)  host %6.2f us  drained %6.2f us\n", h / n, (now() - t) / n);
        uint64_t* sig64;
        CK(hipExtMallocWithFlags((void**)&sig64, 8, hipMallocSignalMemory));
        *(volatile uint64_t*)sig64 = 0;
        t = now();
        for (int i = 1; i <= n; i++) {
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, a);
            CK(hipStreamWriteValue64(s1, sig64, (uint64_t)i, 0));
            CK(hipStreamWaitValue64(s2, sig64, (uint64_t)i, hipStreamWaitValueGte, ~0ull));
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s2, b);
        }
        h = now() - t; CK(hipDeviceSynchronize());
        printf("launch,write64,wait64,launch(other)  host %6.2f us  drained %6.2f us  final %llu\n", h / n, (now() - t) / n, (unsigned long long)*(volatile uint64_t*)sig64);
    }
    // two submitting threads, one stream each
    for (int nt = 1; nt <= 2; nt++) {
        CK(hipDeviceSynchronize());
        t = now();
        auto work = [&](hipStream_t s, float* p) { for (int i = 0; i < 4 * n; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, p); };
        std::thread th1(work, s1, a);
        if (nt == 2) { std::thread th2(work, s2, b); th2.join(); }
        th1.join();
        h = now() - t; CK(hipDeviceSynchronize());
        printf("%d thread(s) x %d launches            host %6.2f us per launch per thread  drained %6.2f\n", nt, 4 * n, h / (4 * n), (now() - t) / (4 * n));
    }
    float ha = 0, hb = 0;
    CK(hipMemcpy(&ha, a, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost));
    printf("counters %.0f %.0f\n", ha, hb);
    return 0;
}
