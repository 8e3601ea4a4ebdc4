// When does a polling host thread SEE an event that was recorded in the middle of a busy stream?
//   stream: A (50 us) | record e | B (200 us) [| record e2 | C (50 us)]
// The side thread of the layer graph polls hipEventQuery(e) to launch dependent work on another stream the moment A has finished.
// If the runtime only learns about e when a later signalled packet completes, the dependent work starts late.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/event_poll_probe tools/event_poll_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tick(volatile int* flag, int value) { *flag = value; }
__global__ void k_spin(long long ticks, volatile int* flag, int value) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (flag && threadIdx.x == 0) { *flag = value; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));  // kHz
    const long long perUs = rate / 1000;
    printf("wall clock %d kHz\n", rate);
    volatile int* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    for (int variant = 0; variant < 4; variant++) {
        hipEvent_t e, e2;
        CK(hipEventCreateWithFlags(&e, variant & 1 ? hipEventDefault : hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        double seen = 0, flagSeen = 0, done = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; r++) {
            *flag = 0;
            CK(hipDeviceSynchronize());
            const double t0 = now();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 50 * perUs, flag, 1);
            CK(hipEventRecord(e, s1));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 200 * perUs, (volatile int*)nullptr, 0);
            if (variant & 2) {
                CK(hipEventRecord(e2, s1));
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 50 * perUs, (volatile int*)nullptr, 0);
            }
            double tf = 0, te = 0;
            while (te == 0) {
                if (tf == 0 && *flag == 1) tf = now() - t0;
                if (hipEventQuery(e) == hipSuccess) te = now() - t0;
            }
            if (tf == 0) tf = te;
            CK(hipStreamSynchronize(s1));
            const double td = now() - t0;
            if (r >= 2) { seen += te; flagSeen += tf; done += td; }
        }
        printf("event %-14s %s: A's flag seen at %6.1f us, hipEventQuery(e) ready at %6.1f us, stream drained at %6.1f us\n",
               variant & 1 ? "(timing)" : "(disableTiming)", variant & 2 ? "A|e|B|e2|C" : "A|e|B      ", flagSeen / reps, seen / reps, done / reps);
        CK(hipEventDestroy(e)); CK(hipEventDestroy(e2));
    }
    {   // the same question for a one-thread kernel that writes a host-visible flag instead of an event
        double seen = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; r++) {
            *flag = 0;
            CK(hipDeviceSynchronize());
            const double t0 = now();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 50 * perUs, (volatile int*)nullptr, 0);
            hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, s1, flag, 7);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 200 * perUs, (volatile int*)nullptr, 0);
            while (*flag != 7) {}
            const double tf = now() - t0;
            CK(hipStreamSynchronize(s1));
            if (r >= 2) seen += tf;
        }
        printf("A|tick kernel|B: flag seen at %6.1f us\n", seen / reps);
    }
    // device-side cost of a recorded event between two kernels of one stream: 20 kernels of 10 us back to back
    hipEvent_t ev[20];
    for (auto& x : ev) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    for (int mode = 0; mode < 4; mode++) {
        double tot = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; r++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int k = 0; k < 20; k++) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 10 * perUs, (volatile int*)nullptr, 0);
                if (mode == 1 || mode == 2) CK(hipEventRecord(ev[k], s1));
                if (mode == 3) hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, s1, flag, k);
                if (mode == 2) CK(hipStreamWaitEvent(s2, ev[k], 0));
            }
            CK(hipStreamSynchronize(s1));
            if (r >= 2) tot += now() - t0;
        }
        printf("20 x 10 us kernels, %-34s: %6.1f us\n", mode == 0 ? "back to back" : mode == 1 ? "an event recorded after each" : mode == 2 ? "event + wait on another stream" : "a tick kernel after each", tot / reps);
    }
    return 0;
}
