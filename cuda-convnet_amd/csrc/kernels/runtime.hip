// Runtime entry points of the C ABI: device / memory / stream / event plumbing
// and the lazily loaded RCCL communicator (include/mi_convnet.h).
#include <dlfcn.h>
#include <stdarg.h>
#include <string.h>
#include "common.h"
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>

static thread_local char g_err[512] = "";

void mcn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mcn_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return MCN_OK;
    mcn_set_error("%s: %s", what, hipGetErrorString(e));
    return MCN_EHIP;
}

static thread_local struct {
    int* flag = nullptr;
    int* gate = nullptr;
    int value = 0;
    hipStream_t stream = nullptr;
} t_tick;
__global__ void k_stream_tick(volatile int* flag, int value, volatile int* gate);
static std::atomic<long> g_tickSent[4];
static void tick_send_now(int why = 0) {
    if (!t_tick.flag) return;
    static const bool dbg = getenv("MCN_TICK_DEBUG") != nullptr;
    if (dbg) {
        static bool reg = (atexit([] { fprintf(stderr, "tick kernels: behind a launch %ld, flush %ld, re-arm %ld, stream_tick %ld\n", g_tickSent[0].load(), g_tickSent[1].load(), g_tickSent[2].load(), g_tickSent[3].load()); }), true);
        (void)reg;
        g_tickSent[why]++;
    }
    MCN_LAUNCH(k_stream_tick, dim3(1), dim3(1), 0, t_tick.stream, (volatile int*)t_tick.flag, t_tick.value, (volatile int*)t_tick.gate);
    t_tick.flag = nullptr;
}
McnTick mcn_take_tick(hipStream_t stream) {
    McnTick t = {nullptr, nullptr, 0};
    if (t_tick.flag && t_tick.stream == stream) {
        t.flag = t_tick.flag;
        t.gate = t_tick.gate;
        t.value = t_tick.value;
        t_tick.flag = nullptr;
    }
    return t;
}

static thread_local size_t t_launches = 0;
thread_local uint64_t t_mcnLaunchNs = 0;
int mcn_launch_check(const char* what) {
    t_launches++;
    tick_send_now();  // (a tick the launch above did not carry: right behind it)
    // MCN_LAUNCH_DELAY_NS (diagnostic): busy-wait after every launch -- if a step slows down by launches x delay the step
    // is bound by the submitting threads, if it does not the device is
    static const long delayNs = getenv("MCN_LAUNCH_DELAY_NS") ? atol(getenv("MCN_LAUNCH_DELAY_NS")) : 0;
    if (delayNs > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() < delayNs) {}
    }
    return mcn_check_hip(hipGetLastError(), what);
}
MCN_API size_t mcn_launch_count(void) { return t_launches; }
MCN_API uint64_t mcn_launch_ns(void) { return t_mcnLaunchNs; }

MCN_API const char* mcn_last_error(void) { return g_err; }

MCN_API int mcn_device_count(int* count) { return mcn_check_hip(hipGetDeviceCount(count), "hipGetDeviceCount"); }
MCN_API int mcn_set_device(int device) { return mcn_check_hip(hipSetDevice(device), "hipSetDevice"); }
MCN_API int mcn_alloc(void** ptr, size_t bytes) { return mcn_check_hip(hipMalloc(ptr, bytes ? bytes : 4), "hipMalloc"); }
MCN_API int mcn_free(void* ptr) { return mcn_check_hip(hipFree(ptr), "hipFree"); }
MCN_API int mcn_host_alloc(void** ptr, size_t bytes) { return mcn_check_hip(hipHostMalloc(ptr, bytes ? bytes : 4, hipHostMallocDefault), "hipHostMalloc"); }
MCN_API int mcn_host_free(void* ptr) { return mcn_check_hip(hipHostFree(ptr), "hipHostFree"); }
MCN_API int mcn_memset(void* ptr, int value, size_t bytes, void* stream) {
    return mcn_check_hip(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream), "hipMemsetAsync");
}
MCN_API int mcn_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    return mcn_check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync(h2d)");
}
MCN_API int mcn_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    return mcn_check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync(d2h)");
}
MCN_API int mcn_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    return mcn_check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(d2d)");
}
MCN_API int mcn_copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                       int kind, void* stream) {
    hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    return mcn_check_hip(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, k, (hipStream_t)stream), "hipMemcpy2DAsync");
}
// Every minibatch column slice of a resident data batch in one pass: dst = [minibatch index][rows][columns of that
// minibatch], so that the per-step slice (data.cu:72-98) is a pointer.  One thread per column (VEC = 4: per 4 columns).
template <int VEC>
__global__ __launch_bounds__(256) void k_tile_minibatches(const float* __restrict__ src, float* __restrict__ dst, int rows, long numCases,
                                                           int minibatch) {
    const long c = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (c >= numCases) return;
    const long mi = c / minibatch, c0 = mi * minibatch;
    const long n = numCases - c0 < minibatch ? numCases - c0 : minibatch;  // columns of this (maybe last, ragged) minibatch
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const float* sp = src + (size_t)r * numCases + c;
        float* dp = dst + (size_t)c0 * rows + (size_t)r * n + (c - c0);
        if (VEC == 4) *reinterpret_cast<float4*>(dp) = *reinterpret_cast<const float4*>(sp);
        else *dp = *sp;
    }
}
MCN_API int mcn_tile_minibatches(const float* src, float* dst, int rows, size_t numCasesSz, int minibatch, void* stream) {
    const long numCases = (long)numCasesSz;
    MCN_REQUIRE(src && dst && rows > 0 && numCases > 0 && minibatch > 0, "tileMinibatches: bad argument");
    const bool vec = minibatch % 4 == 0 && numCases % 4 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0;
    const long cols = vec ? numCases / 4 : numCases;
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)std::min(rows, 4096));
    if (vec) MCN_LAUNCH((k_tile_minibatches<4>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, rows, numCases, minibatch);
    else MCN_LAUNCH((k_tile_minibatches<1>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, rows, numCases, minibatch);
    return MCN_LAUNCH_CHECK("tileMinibatches");
}
MCN_API int mcn_stream_create(void** stream) {
    hipStream_t s;
    int rc = mcn_check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    *stream = (void*)s;
    return rc;
}
// A stream on a hardware queue of its own: the runtime hands out its queues per priority level, and streams of one level
// share them once there are more streams than queues (measured with rocprofv3 on MI355X / ROCm 7.2: a fifth normal stream
// landed on the first one's queue, and the exchange stream's collectives ran in line with the main stream's kernels).
MCN_API int mcn_stream_create_priority(void** stream, int level) {  // level < 0: lowest, > 0: highest, 0: normal
    int least = 0, greatest = 0;
    int rc = mcn_check_hip(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    if (rc) return rc;
    hipStream_t s;
    rc = mcn_check_hip(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, level < 0 ? least : level > 0 ? greatest : 0), "hipStreamCreateWithPriority");
    *stream = (void*)s;
    return rc;
}
MCN_API int mcn_stream_destroy(void* stream) { return mcn_check_hip(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy"); }
MCN_API int mcn_stream_sync(void* stream) { return mcn_check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"); }
MCN_API int mcn_device_sync(void) { return mcn_check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize"); }
MCN_API int mcn_event_create(void** event) {
    hipEvent_t e;
    int rc = mcn_check_hip(hipEventCreate(&e), "hipEventCreate");
    *event = (void*)e;
    return rc;
}
MCN_API int mcn_event_create_sync(void** event) {  // ordering only: no timestamps
    hipEvent_t e;
    int rc = mcn_check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
    *event = (void*)e;
    return rc;
}
MCN_API int mcn_event_destroy(void* event) { return mcn_check_hip(hipEventDestroy((hipEvent_t)event), "hipEventDestroy"); }
MCN_API int mcn_event_record(void* event, void* stream) {
    return mcn_check_hip(hipEventRecord((hipEvent_t)event, (hipStream_t)stream), "hipEventRecord");
}
MCN_API int mcn_event_sync(void* event) { return mcn_check_hip(hipEventSynchronize((hipEvent_t)event), "hipEventSynchronize"); }
// 1: everything recorded before the event has completed; 0: not yet; < 0: error
MCN_API int mcn_event_query(void* event) {
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return 0;
    }
    return mcn_check_hip(e, "hipEventQuery");
}
MCN_API int mcn_stream_wait_event(void* stream, void* event) {
    return mcn_check_hip(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0), "hipStreamWaitEvent");
}
MCN_API int mcn_event_elapsed_ms(float* ms, void* start, void* stop) {
    return mcn_check_hip(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
}

// ---------------------------------------------------------------------------
// Step graphs: everything issued on `stream` (and on streams that join it through events) between begin and end becomes
// one executable graph; a replay costs one call instead of one per kernel / event / wait.
// ---------------------------------------------------------------------------
MCN_API int mcn_graph_begin_capture(void* stream) {
    return mcn_check_hip(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
}
MCN_API int mcn_graph_end_capture(void* stream, void** graphExec, int* numNodes) {
    MCN_REQUIRE(graphExec, "graphEndCapture: null pointer");
    *graphExec = nullptr;
    hipGraph_t g = nullptr;
    int rc = mcn_check_hip(hipStreamEndCapture((hipStream_t)stream, &g), "hipStreamEndCapture");
    if (rc) return rc;
    if (numNodes) {
        size_t n = 0;
        (void)hipGraphGetNodes(g, nullptr, &n);
        *numNodes = (int)n;
    }
    hipGraphExec_t e = nullptr;
    rc = mcn_check_hip(hipGraphInstantiate(&e, g, nullptr, nullptr, 0), "hipGraphInstantiate");
    (void)hipGraphDestroy(g);
    if (rc) return rc;
    *graphExec = (void*)e;
    return MCN_OK;
}
MCN_API int mcn_graph_launch(void* graphExec, void* stream) {
    return mcn_check_hip(hipGraphLaunch((hipGraphExec_t)graphExec, (hipStream_t)stream), "hipGraphLaunch");
}
MCN_API int mcn_graph_destroy(void* graphExec) {
    return graphExec ? mcn_check_hip(hipGraphExecDestroy((hipGraphExec_t)graphExec), "hipGraphExecDestroy") : MCN_OK;
}

// ---------------------------------------------------------------------------
// Stream ticks: a one-thread kernel that writes a sequence number into host-visible memory when the stream reaches it.
// A submitting thread that POLLS for "the main stream has passed this point" (the layer graph's side thread) sees the tick
// 5 us before hipEventQuery reports an event recorded at the same place, and the tick costs the stream 1.5 us against 3.0 us
// for the event's marker packet (tools/event_poll_probe.hip, MI355X / ROCm 7.2: 20 x 10 us kernels take 225 us back to
// back, 255 us with a tick after each, 285 us with an event after each).
// ---------------------------------------------------------------------------
__global__ void k_stream_tick(volatile int* flag, int value, volatile int* gate) {
    if (gate) *gate = value;
    *flag = value;
}
// The tick rides on this thread's NEXT launch: in the first instructions of a kernel that has a tick slot (the classifier
// products, the multi-input backward-data product, imgActs, the strip weight gradient), else in a one-thread kernel behind
// it.  mcn_flush_tick sends an armed tick at once (before the caller blocks on whoever waits for it).
MCN_API int mcn_arm_tick(int* flag, int value, void* stream) { return mcn_arm_tick_gate(flag, nullptr, value, stream); }
MCN_API int mcn_arm_tick_gate(int* flag, int* gate, int value, void* stream) {
    MCN_REQUIRE(flag, "armTick: null pointer");
    tick_send_now(2);  // (an older one first: ticks of a stream stay in order)
    t_tick.flag = flag;
    t_tick.gate = gate;
    t_tick.value = value;
    t_tick.stream = (hipStream_t)stream;
    return MCN_OK;
}
MCN_API int mcn_flush_tick(void) {
    tick_send_now(1);
    return mcn_check_hip(hipGetLastError(), "flushTick");
}
MCN_API int mcn_tick_alloc(int** flag) {
    MCN_REQUIRE(flag, "tickAlloc: null pointer");
    int rc = mcn_check_hip(hipHostMalloc((void**)flag, 64, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc (tick)");
    if (rc == MCN_OK) **flag = 0;
    return rc;
}
MCN_API int mcn_tick_free(int* flag) { return flag ? mcn_check_hip(hipHostFree(flag), "hipHostFree (tick)") : MCN_OK; }
MCN_API int mcn_stream_tick(int* flag, int value, void* stream) {
    MCN_REQUIRE(flag, "streamTick: null pointer");
    tick_send_now(3);  // (an armed one first: ticks stay in order)
    MCN_LAUNCH(k_stream_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, (volatile int*)flag, value, (volatile int*)nullptr);
    return mcn_check_hip(hipGetLastError(), "streamTick");
}

// ---------------------------------------------------------------------------
// Stream gates: cross-stream ordering by a spinning one-thread kernel instead of a barrier packet (mi_convnet.h).
// ---------------------------------------------------------------------------
__global__ void k_stream_signal(int* gate, int value) {
    __hip_atomic_store(gate, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_stream_gate(int* gate, int value, long long timeoutTicks) {
    const long long t0 = wall_clock64();  // 100 MHz constant clock
    for (;;) {
        const int v = __hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - value) >= 0) return;
        if (wall_clock64() - t0 > timeoutTicks) {  // the producer never came: end anyway, say so
            __hip_atomic_store(gate + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
MCN_API int mcn_gate_alloc(int** gate) {
    MCN_REQUIRE(gate, "gateAlloc: null pointer");
    int rc = mcn_check_hip(hipMalloc((void**)gate, 64), "hipMalloc (gate)");
    if (rc) return rc;
    return mcn_check_hip(hipMemset(*gate, 0, 64), "hipMemset (gate)");
}
MCN_API int mcn_gate_free(int* gate) { return gate ? mcn_check_hip(hipFree(gate), "hipFree (gate)") : MCN_OK; }
MCN_API int mcn_stream_signal(int* gate, int value, void* stream) {
    MCN_REQUIRE(gate, "streamSignal: null pointer");
    MCN_LAUNCH(k_stream_signal, dim3(1), dim3(1), 0, (hipStream_t)stream, gate, value);
    return mcn_check_hip(hipGetLastError(), "streamSignal");
}
MCN_API int mcn_stream_gate(int* gate, int value, int timeoutMs, void* stream) {
    MCN_REQUIRE(gate && timeoutMs > 0, "streamGate: bad argument");
    tick_send_now(3);  // (an armed tick of this thread first: whoever polls for it may be the one that signals this gate)
    MCN_LAUNCH(k_stream_gate, dim3(1), dim3(1), 0, (hipStream_t)stream, gate, value, (long long)timeoutMs * 100000ll);
    return mcn_check_hip(hipGetLastError(), "streamGate");
}
MCN_API int mcn_gate_timed_out(int* gate, int* timedOut, void* stream) {
    MCN_REQUIRE(gate && timedOut, "gateTimedOut: null pointer");
    int rc = mcn_check_hip(hipMemcpyAsync(timedOut, gate + 1, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync (gate)");
    if (rc) return rc;
    return mcn_check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize (gate)");
}
MCN_API int mcn_gate_selftest(int* gate, void* waiter, void* signaller, int* concurrent) {
    MCN_REQUIRE(gate && concurrent && waiter != signaller, "gateSelftest: bad argument");
    *concurrent = 0;
    int rc = mcn_check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (rc) return rc;
    rc = mcn_check_hip(hipMemset(gate, 0, 64), "hipMemset (gate)");
    if (rc) return rc;
    // the gate first, its signal only once the gate is certainly spinning: streams that share a hardware queue never
    // reach the signal, the gate runs into its (short) timeout and says so
    MCN_LAUNCH(k_stream_gate, dim3(1), dim3(1), 0, (hipStream_t)waiter, gate, 1, 20ll * 100000ll);
    const uint64_t t0 = mcn_now_ns();
    while (mcn_now_ns() - t0 < 2000000ull) {}
    MCN_LAUNCH(k_stream_signal, dim3(1), dim3(1), 0, (hipStream_t)signaller, gate, 1);
    rc = mcn_check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (rc) return rc;
    int h[2] = {0, 0};
    rc = mcn_check_hip(hipMemcpy(h, gate, sizeof(h), hipMemcpyDeviceToHost), "hipMemcpy (gate)");
    if (rc) return rc;
    *concurrent = (h[0] == 1 && h[1] == 0) ? 1 : 0;
    return mcn_check_hip(hipMemset(gate, 0, 64), "hipMemset (gate)");
}

// ---------------------------------------------------------------------------
// RCCL, loaded on first use so that single-GPU users never pay for it.
// Only the entry points below are bound (signatures from rccl.h).
// ---------------------------------------------------------------------------
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
static struct {
    void* handle;
    int (*GetUniqueId)(rcclUniqueId*);
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
    int (*CommDestroy)(rcclComm_t);
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*CommCount)(rcclComm_t, int*);
    int (*GetVersion)(int*);
    const char* (*GetErrorString)(int);
} g_rccl;

static int rccl_load() {
    if (g_rccl.handle) return MCN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        mcn_set_error("cannot load RCCL: %s", dlerror());
        return MCN_ERCCL;
    }
#define BIND(field, sym)                                        \
    *(void**)(&g_rccl.field) = dlsym(h, sym);                   \
    if (!g_rccl.field) {                                        \
        mcn_set_error("RCCL symbol %s missing", sym);           \
        return MCN_ERCCL;                                       \
    }
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllReduce, "ncclAllReduce");
    BIND(AllGather, "ncclAllGather");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    *(void**)(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");  // (optional: only reported)
    *(void**)(&g_rccl.GetVersion) = dlsym(h, "ncclGetVersion");
    g_rccl.handle = h;
    return MCN_OK;
}

static int rccl_check(int rc, const char* what) {
    if (rc == 0) return MCN_OK;
    mcn_set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return MCN_ERCCL;
}

// ---------------------------------------------------------------------------
// Rehearsal communicator for ranks that SHARE one GPU (MCN_COMM=shm) -- TEST BUILD ONLY (-DMCN_TEST_HOOKS,
// libmiconvnet_hooks.so): the product library does not contain it.  RCCL refuses a second rank on a device
// ("duplicate GPU"), so a one-GPU box can never run the data-parallel step with world > 1.  This transport lets it:
// same entry points, same stream-ordered, ASYNCHRONOUS meaning as RCCL -- a call only queues work on the stream and
// returns: D2H into the rank's slot -> host function (barrier; every rank adds the slots in rank order) -> H2D ->
// host function (barrier: nobody overwrites a slot before every rank has read it).  Nothing in it blocks the calling
// thread, so an ordering hazard between the exchange stream and the compute streams shows up exactly as it would with RCCL.
// The sum has a fixed order, so all ranks end with bit-identical buffers, as with RCCL's ring on identical inputs.
// ---------------------------------------------------------------------------
struct ShmComm;
struct McnComm {
    rcclComm_t rccl = nullptr;
    ShmComm* shm = nullptr;
};
#ifdef MCN_TEST_HOOKS
struct ShmHeader {
    std::atomic<uint32_t> arrived, generation, attached;
    uint32_t numRanks;
    uint64_t slotBytes;
};
struct ShmComm {
    int rank = 0, numRanks = 1;
    size_t slotBytes = 0, mapBytes = 0;
    ShmHeader* hdr = nullptr;
    char* slots = nullptr;
    float* bounce = nullptr;  // page-locked, slotBytes
    char name[96] = {0};
    std::atomic<int> failed{0};  // a barrier inside a host function timed out: reported by the next call
    float* stage = nullptr;      // page-locked, slotBytes: this rank's contribution on its way to the segment
    // collectives of one communicator are ordered even when they are issued on different streams (as RCCL orders them):
    // each one waits for the previous one's completion event before it touches the staging buffers
    hipEvent_t lastDone = nullptr;
    hipStream_t lastStream = nullptr;
    bool any = false;
    std::mutex order;
};
static const char kShmTag[] = "MCNSHM:";
static const size_t kShmSlotBytes = 8u << 20;
static const size_t kShmHeaderBytes = 4096;

static int shm_barrier(ShmComm* c, const char* what) {
    const uint32_t gen = c->hdr->generation.load(std::memory_order_acquire);
    if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->numRanks) {
        c->hdr->arrived.store(0, std::memory_order_relaxed);
        c->hdr->generation.fetch_add(1, std::memory_order_acq_rel);
        return MCN_OK;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    for (unsigned spin = 0; c->hdr->generation.load(std::memory_order_acquire) == gen; spin++) {
        if ((spin & 1023) == 1023) {
            if (std::chrono::steady_clock::now() > deadline) {
                mcn_set_error("%s: a peer did not reach the collective within 120 s (shared-memory communicator)", what);
                return MCN_ERCCL;
            }
            sched_yield();
        }
    }
    return MCN_OK;
}

static int shm_init(McnComm* mc, int numRanks, int rank, const char* name) {
    ShmComm* c = new ShmComm;
    c->rank = rank;
    c->numRanks = numRanks;
    c->slotBytes = kShmSlotBytes;
    c->mapBytes = kShmHeaderBytes + (size_t)numRanks * c->slotBytes;
    snprintf(c->name, sizeof(c->name), "%s", name);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)c->mapBytes) != 0) { close(fd); fd = -1; }
    } else {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        struct stat st;
        while (true) {  // wait until rank 0 has created AND sized the segment
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->mapBytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (std::chrono::steady_clock::now() > deadline) break;
            usleep(2000);
        }
    }
    if (fd < 0) {
        mcn_set_error("shared-memory communicator: cannot open %s (rank %d): %s", c->name, rank, strerror(errno));
        delete c;
        return MCN_ERCCL;
    }
    void* m = mmap(nullptr, c->mapBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        mcn_set_error("shared-memory communicator: mmap failed: %s", strerror(errno));
        delete c;
        return MCN_ERCCL;
    }
    c->hdr = (ShmHeader*)m;  // a fresh segment is zero-filled: arrived = generation = attached = 0
    c->slots = (char*)m + kShmHeaderBytes;
    if (rank == 0) {
        c->hdr->numRanks = (uint32_t)numRanks;
        c->hdr->slotBytes = c->slotBytes;
    }
    c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    while (c->hdr->attached.load(std::memory_order_acquire) < (uint32_t)numRanks) {
        if (std::chrono::steady_clock::now() > deadline) {
            mcn_set_error("shared-memory communicator: %u of %d ranks attached within 120 s", c->hdr->attached.load(), numRanks);
            munmap(m, c->mapBytes);
            if (rank == 0) shm_unlink(c->name);
            delete c;
            return MCN_ERCCL;
        }
        usleep(1000);
    }
    if (rank == 0) shm_unlink(c->name);  // everyone holds a mapping: the name can go
    // The device only ever copies to and from PAGE-LOCKED buffers of this process (`stage` out, `bounce` in): a copy that
    // touches pageable memory -- the shared segment cannot be registered -- is staged by the runtime when the call is MADE,
    // not when the stream reaches it, and the H2D of a collective would read the slots before the peers have filled them.
    // The segment itself is read and written by the host functions only.
    int rc = mcn_check_hip(hipHostMalloc((void**)&c->bounce, c->slotBytes, hipHostMallocDefault), "hipHostMalloc");
    if (!rc) rc = mcn_check_hip(hipHostMalloc((void**)&c->stage, c->slotBytes, hipHostMallocDefault), "hipHostMalloc");
    if (rc) {
        if (c->bounce) (void)hipHostFree(c->bounce);
        munmap(m, c->mapBytes);
        delete c;
        return rc;
    }
    mc->shm = c;
    return MCN_OK;
}

// Host functions of one chunk, run by the stream in order (hipLaunchHostFunc); they may block (barriers with the peers)
// but never the thread that queued them.
struct ShmStep {
    ShmComm* c;
    size_t n;
    bool gather, last;
};
static void shm_host_reduce(void* arg) {
    ShmStep* st = (ShmStep*)arg;
    ShmComm* c = st->c;
    memcpy(c->slots + (size_t)c->rank * c->slotBytes, c->stage, st->n * 4);
    if (shm_barrier(c, "collective (shm)")) c->failed.store(1);  // every rank's slot is filled
    if (st->gather) {
        for (int r = 0; r < c->numRanks; r++) memcpy(c->bounce + (size_t)r * st->n, c->slots + (size_t)r * c->slotBytes, st->n * 4);
    } else {
        const float* s0 = (const float*)c->slots;
        for (size_t i = 0; i < st->n; i++) c->bounce[i] = s0[i];
        for (int r = 1; r < c->numRanks; r++) {
            const float* sr = (const float*)(c->slots + (size_t)r * c->slotBytes);
            for (size_t i = 0; i < st->n; i++) c->bounce[i] += sr[i];
        }
    }
    if (shm_barrier(c, "collective (shm)")) c->failed.store(1);  // every rank has read the slots: they may be refilled
    delete st;
}

// one chunk of at most slotBytes: recv[0..n) <- sum over ranks (rank order) / recv <- concatenation; only queues work on s
static int shm_exchange(ShmComm* c, const float* send, float* recv, size_t n, bool gather, hipStream_t s, const char* what) {
    if (c->failed.load()) {
        mcn_set_error("%s: an earlier collective timed out waiting for a peer (shared-memory communicator)", what);
        return MCN_ERCCL;
    }
    std::lock_guard<std::mutex> lk(c->order);
    if (!c->lastDone && mcn_check_hip(hipEventCreateWithFlags(&c->lastDone, hipEventDisableTiming), what)) return MCN_EHIP;
    if (c->any && c->lastStream != s && mcn_check_hip(hipStreamWaitEvent(s, c->lastDone, 0), what)) return MCN_EHIP;
    ShmStep* st = new ShmStep{c, n, gather, true};
    int rc = mcn_check_hip(hipMemcpyAsync(c->stage, send, n * 4, hipMemcpyDeviceToHost, s), what);
    if (!rc) rc = mcn_check_hip(hipLaunchHostFunc(s, shm_host_reduce, st), what);
    else delete st;
    if (!rc) rc = mcn_check_hip(hipMemcpyAsync(recv, c->bounce, (gather ? (size_t)c->numRanks : 1) * n * 4, hipMemcpyHostToDevice, s), what);
    if (!rc) rc = mcn_check_hip(hipEventRecord(c->lastDone, s), what);
    c->lastStream = s;
    c->any = true;
    return rc;
}
#endif  // MCN_TEST_HOOKS

MCN_API int mcn_comm_unique_id(char idBytes[128]) {
    const char* kind = getenv("MCN_COMM");
#ifndef MCN_TEST_HOOKS
    if (kind && strcmp(kind, "shm") == 0) {
        mcn_set_error("MCN_COMM=shm: the rehearsal communicator is only in the test build (libmiconvnet_hooks.so)");
        return MCN_ERCCL;
    }
#else
    if (kind && strcmp(kind, "shm") == 0) {
        memset(idBytes, 0, 128);
        unsigned long long r = 0;
        FILE* f = fopen("/dev/urandom", "rb");
        if (f) { if (fread(&r, sizeof r, 1, f) != 1) r = 0; fclose(f); }
        r ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
        snprintf(idBytes, 128, "%s/mcn_%d_%016llx", kShmTag, (int)getpid(), r);
        return MCN_OK;
    }
#endif
    int rc = rccl_load();
    if (rc) return rc;
    rcclUniqueId id;
    rc = rccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(idBytes, id.internal, 128);
    return MCN_OK;
}

MCN_API int mcn_comm_init(void** comm, int numRanks, int rank, const char idBytes[128]) {
    MCN_REQUIRE(comm && idBytes && numRanks >= 1 && rank >= 0 && rank < numRanks, "mcn_comm_init: bad argument");
    McnComm* mc = new McnComm;
    int rc;
#ifdef MCN_TEST_HOOKS
    if (strncmp(idBytes, kShmTag, sizeof(kShmTag) - 1) == 0 && memchr(idBytes, 0, 128)) {
        rc = shm_init(mc, numRanks, rank, idBytes + sizeof(kShmTag) - 1);
    } else
#endif
    {
        rc = rccl_load();
        if (!rc) {
            rcclUniqueId id;
            memcpy(id.internal, idBytes, 128);
            rc = rccl_check(g_rccl.CommInitRank(&mc->rccl, numRanks, id, rank), "ncclCommInitRank");
        }
    }
    if (rc) {
        delete mc;
        *comm = nullptr;
        return rc;
    }
    *comm = mc;
    return MCN_OK;
}

MCN_API int mcn_comm_destroy(void* comm) {
    McnComm* mc = (McnComm*)comm;
    if (!mc) return MCN_OK;
    int rc = MCN_OK;
#ifdef MCN_TEST_HOOKS
    if (mc->shm) {
        (void)hipDeviceSynchronize();  // (host functions of queued collectives still hold the segment)
        if (mc->shm->bounce) (void)hipHostFree(mc->shm->bounce);
        if (mc->shm->stage) (void)hipHostFree(mc->shm->stage);
        if (mc->shm->lastDone) (void)hipEventDestroy(mc->shm->lastDone);
        munmap(mc->shm->hdr, mc->shm->mapBytes);
        delete mc->shm;
    } else
#endif
    if (mc->rccl && g_rccl.handle) {
        rc = rccl_check(g_rccl.CommDestroy(mc->rccl), "ncclCommDestroy");
    }
    delete mc;
    return rc;
}

// ncclFloat32 = 7, ncclSum = 0 (rccl.h enums)
MCN_API int mcn_allreduce_sum(void* comm, float* buf, size_t count, void* stream) {
    McnComm* mc = (McnComm*)comm;
    MCN_REQUIRE(mc && buf, "mcn_allreduce_sum: communicator not initialised");
#ifdef MCN_TEST_HOOKS
    if (mc->shm) {
        const size_t per = mc->shm->slotBytes / 4;
        for (size_t off = 0; off < count; off += per)
            if (int rc = shm_exchange(mc->shm, buf + off, buf + off, std::min(per, count - off), false, (hipStream_t)stream, "allreduce (shm)"))
                return rc;
        return MCN_OK;
    }
#endif
    MCN_REQUIRE(g_rccl.handle && mc->rccl, "mcn_allreduce_sum: communicator not initialised");
    return rccl_check(g_rccl.AllReduce(buf, buf, count, 7, 0, mc->rccl, (hipStream_t)stream), "ncclAllReduce");
}

MCN_API int mcn_allgather(void* comm, const float* send, float* recv, size_t countPerRank, void* stream) {
    McnComm* mc = (McnComm*)comm;
    MCN_REQUIRE(mc && send && recv, "mcn_allgather: communicator not initialised");
#ifdef MCN_TEST_HOOKS
    if (mc->shm) {
        MCN_REQUIRE(countPerRank * 4 * mc->shm->numRanks <= mc->shm->slotBytes, "mcn_allgather: more than %zu bytes in all (shared-memory communicator)",
                    mc->shm->slotBytes);
        return shm_exchange(mc->shm, send, recv, countPerRank, true, (hipStream_t)stream, "allgather (shm)");
    }
#endif
    MCN_REQUIRE(g_rccl.handle && mc->rccl, "mcn_allgather: communicator not initialised");
    return rccl_check(g_rccl.AllGather(send, recv, countPerRank, 7, mc->rccl, (hipStream_t)stream), "ncclAllGather");
}

// what the communicator is: ranks RCCL itself counts (ncclCommCount), RCCL's version code, 1 = RCCL / 0 = the test build's
// rehearsal transport
MCN_API int mcn_comm_info(void* comm, int* numRanks, int* version, int* isRccl) {
    McnComm* mc = (McnComm*)comm;
    MCN_REQUIRE(mc && numRanks && version && isRccl, "mcn_comm_info: bad argument");
    *numRanks = 0; *version = 0; *isRccl = 0;
#ifdef MCN_TEST_HOOKS
    if (mc->shm) {
        *numRanks = mc->shm->numRanks;
        return MCN_OK;
    }
#endif
    MCN_REQUIRE(g_rccl.handle && mc->rccl, "mcn_comm_info: communicator not initialised");
    *isRccl = 1;
    if (g_rccl.CommCount) {
        int rc = rccl_check(g_rccl.CommCount(mc->rccl, numRanks), "ncclCommCount");
        if (rc) return rc;
    }
    if (g_rccl.GetVersion) return rccl_check(g_rccl.GetVersion(version), "ncclGetVersion");
    return MCN_OK;
}

// collectives issued between the two calls are fused by RCCL into one launch
MCN_API int mcn_comm_group_begin() {
    if (!g_rccl.handle) return MCN_OK;  // shared-memory communicator: collectives run one by one, in the same order on every rank
    return rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
}

MCN_API int mcn_comm_group_end() {
    if (!g_rccl.handle) return MCN_OK;
    return rccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
}

