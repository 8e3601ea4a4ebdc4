// imgActs (backward-data) for shared-weight conv and unshared local layers on fp32 MFMA.
//
// Replaces img_acts.cu:52-188 / :217-360 / :392-518 (+ host _imgActs :900-1912).
// Gather form, like the reference: one wave owns one input pixel (y, x), 16*CV
// channels of one group and 64*NQ images, and sums over the modules that cover
// the pixel and over the group's filters:
//   dImg[c, y, x, n] = sum_{m covers (y,x)} sum_f hid[f, m, n] * W[(m,) c*fp + p(y,x,m), f]
// MFMA roles: A[row=c][k=f] from the weight rows (f is the contiguous dimension,
// i.e. the REDUCTION index), B[k=f][col=n] from hidActs rows.  The reduction
// index of an MFMA may be permuted as long as A and B agree, so each lane reads
// 4 consecutive filters of its weight row with one 16-byte load and uses
// component s in k-step s; the matching B rows are f = f16 + 4*kk + s.
// No atomics, no LDS transposes; results are deterministic.
#include <cstring>
#include "common.h"

struct BwdDataParams {
    const float* hidActs;
    const float* filters;
    float* targets;
    int N, imgY, imgX, C, F, fs, mY, mX, pad, stride, groups, conv;
    float sT, sO;
    // UNDO form: the image is the output of a 2x2 / stride 2 max pool over (2 imgY, 2 imgX) maps
    const float* uActs;  // the pool's input  (C * 4 imgPixels, N)
    const float* uMax;   // the pool's output (C * imgPixels, N)
    float* uTarget;      // gradient w.r.t. the pool's input
    int uNeuron;         // >= 0: times the gradient of the neuron that produced uActs
    float uNa, uNb;
    McnTick tick;        // stored by the launch's first thread (MCN_TICK_PROLOGUE)
    int waveCt;          // 1: the four waves of a block are four channel tiles of ONE pixel (they read the same gradient rows)
    int xcdPair;         // 1: launch order remapped so that the image tiles of a pixel block share an XCD
};

// A step = one (module, 16 filters) pair.  The steps a pixel needs are walked in batches of U: the loads of the
// next batch (second register set) are in flight while the MFMAs of the current one issue, so a wave pays one memory
// round trip per batch instead of one per step.
// KS = 2 / 4: that many waves share a pixel and take every KS-th 16-filter step; their accumulators are added through LDS.  With
// ~1.3 waves per SIMD the plain form waits on memory 46 % of the time (profiles/r01_pmc_conv2_imgacts_sq_tcp.json).
// UNDO: the pixel's gradient does not go to `targets` but straight through the max pool below (whose windows do not overlap:
// every pool input belongs to exactly one output) and the neuron under it -- what convLocalMaxUndo + the neuron's
// gradient kernel would compute from `targets` in a second and third pass (layer.cu:734, :251), bit for bit.
// UNDO: 0 off, 1 through the pool only, 2 through the pool and a relu (a run-time neuron switch per element was tried: the
// unrolled epilogue grew to 28 k instructions and the launch lost more in instruction fetch than the fusion saved)
// (A statically pipelined form for the long step chains of wide layers -- full batches only, one batch in flight behind explicit
// s_waitcnt, scheduling barriers, as in k_weight_acts_wide -- was built and measured in round 3: DeepID2+ conv2 120 -> 154 us and
// local3 115 -> 137 us alone with one wave per pixel, 86 -> 81 us with the channel-tile blocks below, and the two-stream step got
// SLOWER either way (0.770 -> 0.800 ms): its 230 VGPRs and the deeper load queue take from the weight-gradient kernels on the
// other queue what the chain gains.  Removed again.)
template <int CV, int NQ, bool VECN, int U, int KS = 1, int UNDO = 0>
__global__ __launch_bounds__(256) void k_img_acts(BwdDataParams p) {
    MCN_TICK_PROLOGUE(p);
    const int fc = p.C / p.groups, fpg = p.F / p.groups, fp = p.fs * p.fs;
    const int M = p.mY * p.mX, imgPixels = p.imgY * p.imgX;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler: scalar loop control, no exec-masked loops
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const bool wct = KS == 1 && p.waveCt;
    // unshared layers: the image tiles of a pixel block read the same weight rows -- one XCD per pixel block (mcn_xcd_group)
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcdPair) mcn_xcd_group((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, (int)gridDim.y, by, bx);
    const int pxRaw = wct ? by : by * (4 / KS) + wave / KS, ks = wave % KS;
    const bool live = pxRaw < imgPixels;
    if (KS == 1 && !live) return;  // (with KS == 2 every wave reaches the barrier below)
    const int px = live ? pxRaw : imgPixels - 1;
    const int y = px / p.imgX, x = px % p.imgX;
    const int n0 = bx * (64 * NQ);
    const int ctiles = (fc + 16 * CV - 1) / (16 * CV);
    const int zi = wct ? (int)blockIdx.z * 4 + wave : (int)blockIdx.z;
    const int grp = zi / ctiles;
    const int c0 = (zi % ctiles) * 16 * CV;  // channel within the group

    // modules covering this pixel (img_acts.cu:435-440)
    const int startY = y - p.pad < p.fs ? 0 : 1 + (y - p.pad - p.fs) / p.stride;
    const int endY = min(p.mY, 1 + (y - p.pad) / p.stride);
    const int startX = x - p.pad < p.fs ? 0 : 1 + (x - p.pad - p.fs) / p.stride;
    const int endX = min(p.mX, 1 + (x - p.pad) / p.stride);

    f32x4 acc[CV][NQ * 4];
#pragma unroll
    for (int v = 0; v < CV; v++)
#pragma unroll
        for (int t = 0; t < NQ * 4; t++) acc[v][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane weight row (channel) for each channel tile; rows beyond the group's channels are
    // clamped (their outputs are never stored)
    int crow[CV];
#pragma unroll
    for (int v = 0; v < CV; v++) crow[v] = min(c0 + 16 * v + j, fc - 1) * fp;

    const float* hb = p.hidActs + (size_t)grp * fpg * M * p.N + n0 + 4 * j;
    const float* wb = p.filters + (size_t)grp * fpg + 4 * kk;

    const int f16Steps = (fpg / 16 - ks + KS - 1) / KS;  // this wave's 16-filter steps per module
    const int total = live ? max(0, endY - startY) * max(0, endX - startX) * f16Steps : 0;
    int lmy = startY, lmx = startX, lf16 = 16 * ks;  // the next step to load
    auto load = [&](f32x4(&a)[U][CV], f32x4(&b)[U][4][NQ], int count) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < count) {
                const int m = lmy * p.mX + lmx;
                const int pf = (y - (p.pad + lmy * p.stride)) * p.fs + (x - (p.pad + lmx * p.stride));
                const float* wm = wb + ((size_t)(p.conv ? 0 : m) * fc * fp + pf) * p.F + lf16;
                const float* hm = hb + ((size_t)(lf16 + 4 * kk) * M + m) * p.N;
#pragma unroll
                for (int v = 0; v < CV; v++) a[u][v] = *reinterpret_cast<const f32x4*>(wm + (size_t)crow[v] * p.F);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float* hr = hm + (size_t)s * M * p.N;
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const int n = n0 + 64 * q + 4 * j;
                        if (VECN) {
                            b[u][s][q] = n < p.N ? *reinterpret_cast<const f32x4*>(hr + 64 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; t++) b[u][s][q][t] = n + t < p.N ? hr[64 * q + t] : 0.f;
                        }
                    }
                }
                lf16 += 16 * KS;
                if (lf16 >= fpg) {
                    lf16 = 16 * ks;
                    if (++lmx == endX) { lmx = startX; lmy++; }
                }
            }
        }
    };
    auto compute = [&](const f32x4(&a)[U][CV], const f32x4(&b)[U][4][NQ], int count) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < count) {
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int v = 0; v < CV; v++)
#pragma unroll
                        for (int q = 0; q < NQ; q++)
#pragma unroll
                            for (int t = 0; t < 4; t++) acc[v][q * 4 + t] = mfma16(a[u][v][s], b[u][s][q][t], acc[v][q * 4 + t]);
            }
        }
    };
    if constexpr (U == 1) {  // the scalar-load fallback: one step at a time
        f32x4 a0[1][CV], b0[1][4][NQ];
        for (int i = 0; i < total; i++) {
            load(a0, b0, 1);
            compute(a0, b0, 1);
        }
    } else {
        f32x4 a0[U][CV], b0[U][4][NQ], a1[U][CV], b1[U][4][NQ];
        int left = total;  // steps not loaded yet
        int cnt0 = min(U, left), cnt1 = 0;
        load(a0, b0, cnt0);
        left -= cnt0;
        while (cnt0 > 0) {
            cnt1 = min(U, left);
            load(a1, b1, cnt1);
            left -= cnt1;
            compute(a0, b0, cnt0);
            if (cnt1 == 0) break;
            cnt0 = min(U, left);
            load(a0, b0, cnt0);
            left -= cnt0;
            compute(a1, b1, cnt1);
        }
    }

    if constexpr (KS > 1) {
        __shared__ f32x4 red[4 / KS][KS - 1][CV * NQ * 4][64];
        if (ks > 0) {
#pragma unroll
            for (int v = 0; v < CV; v++)
#pragma unroll
                for (int t = 0; t < NQ * 4; t++) red[wave / KS][ks - 1][v * NQ * 4 + t][lane] = acc[v][t];
        }
        __syncthreads();
        if (ks > 0 || !live) return;
#pragma unroll
        for (int o = 0; o < KS - 1; o++)  // fixed order
#pragma unroll
            for (int v = 0; v < CV; v++)
#pragma unroll
                for (int t = 0; t < NQ * 4; t++) acc[v][t] += red[wave / KS][o][v * NQ * 4 + t][lane];
    }

    // (Requesting this epilogue's operands -- 20 16-byte loads per lane -- BEFORE the k loop, so that they arrive while it runs, was
    // measured in round 3: 204 VGPRs instead of ~120 cost a wave slot per SIMD; conv2 21.1 -> 24.1 us, local3 15.9 -> 16.8, only
    // local4's 100 blocks gained, 7.6 -> 6.6.  Left as it is.)
    if constexpr (UNDO) {  // (VECN)
        const int cX = 2 * p.imgX;
#pragma unroll
        for (int v = 0; v < CV; v++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int n = n0 + 64 * q + 4 * j;
                if (n >= p.N) continue;
                f32x4 g[4], ma[4], im[4][4];
#pragma unroll
                for (int r = 0; r < 4; r++) {  // every load of this lane's 4 channels first
                    const int c = min(c0 + 16 * v + 4 * kk + r, fc - 1);
                    const size_t prow = ((size_t)(grp * fc + c) * imgPixels + px) * p.N + n;
                    const size_t crow = ((size_t)(grp * fc + c) * 4 * imgPixels + (size_t)(2 * y) * cX + 2 * x) * p.N + n;
                    if (p.sT != 0.f) g[r] = *reinterpret_cast<const f32x4*>(p.targets + prow);
                    ma[r] = *reinterpret_cast<const f32x4*>(p.uMax + prow);
#pragma unroll
                    for (int w = 0; w < 4; w++)
                        im[r][w] = *reinterpret_cast<const f32x4*>(p.uActs + crow + (size_t)((w >> 1) * cX + (w & 1)) * p.N);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int c = c0 + 16 * v + 4 * kk + r;
                    if (c >= fc) continue;
                    const size_t crow = ((size_t)(grp * fc + c) * 4 * imgPixels + (size_t)(2 * y) * cX + 2 * x) * p.N + n;
                    f32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; t++) o[t] = p.sO * acc[v][q * 4 + t][r];
                    if (p.sT != 0.f) o += p.sT * g[r];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        f32x4 out;
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const float a = 0.f + (im[r][w][t] == ma[r][t] ? 1.f : 0.f) * o[t];  // (the undo kernel's accumulator)
                            out[t] = 1.f * a;
                            if (UNDO == 2) out[t] = neuron_grad(MCN_NEURON_RELU, out[t], im[r][w][t], 0.f, 0.f, 0.f);
                        }
                        *reinterpret_cast<f32x4*>(p.uTarget + crow + (size_t)((w >> 1) * cX + (w & 1)) * p.N) = out;
                    }
                }
            }
        }
        return;
    }
    // D row 4*kk + r of tile v <-> channel c0 + 16v + 4kk + r ; col j of MFMA t <-> image n0+64q+4j+t
#pragma unroll
    for (int v = 0; v < CV; v++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c = c0 + 16 * v + 4 * kk + r;
            if (c >= fc) continue;
            float* trow = p.targets + ((size_t)(grp * fc + c) * imgPixels + px) * p.N;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int n = n0 + 64 * q + 4 * j;
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; t++) o[t] = p.sO * acc[v][q * 4 + t][r];
                if (VECN) {
                    if (n < p.N) {
                        f32x4* dst = reinterpret_cast<f32x4*>(trow + n);
                        if (p.sT != 0.f) o += p.sT * (*dst);
                        *dst = o;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (n + t < p.N) trow[n + t] = (p.sT != 0.f ? p.sT * trow[n + t] : 0.f) + o[t];
                }
            }
        }
    }
}

template <int CV, int NQ>
static void launch_bd2(const BwdDataParams& p, dim3 grid, bool vecn, bool pipelined, int ksplit, hipStream_t s) {
    constexpr int U = CV + 4 * NQ <= 6 ? 4 : 2;  // steps per register set: (CV + 4 NQ) * 4 VGPRs each, two sets
    if (p.uTarget) {  // (vecn: checked by the caller)
#define MCN_UNDO_LAUNCH(UN)                                                                                               \
    do {                                                                                                                  \
        if (pipelined) MCN_LAUNCH((k_img_acts<CV, NQ, true, U, 1, UN>), grid, dim3(256), 0, s, p);                \
        else if (ksplit) {                                                                                                \
            if constexpr (CV == 1 && NQ == 1) {                                                                           \
                grid.y = divup(p.imgY * p.imgX, 4 / ksplit);                                                              \
                if (ksplit == 4) MCN_LAUNCH((k_img_acts<1, 1, true, 2, 4, UN>), grid, dim3(256), 0, s, p);        \
                else MCN_LAUNCH((k_img_acts<1, 1, true, 2, 2, UN>), grid, dim3(256), 0, s, p);                    \
            } else if constexpr (CV == 2 && NQ == 1) {                                                                    \
                grid.y = divup(p.imgY * p.imgX, 4 / ksplit);                                                              \
                if (ksplit == 4) MCN_LAUNCH((k_img_acts<2, 1, true, 2, 4, UN>), grid, dim3(256), 0, s, p);        \
                else MCN_LAUNCH((k_img_acts<2, 1, true, 2, 2, UN>), grid, dim3(256), 0, s, p);                    \
            }                                                                                                             \
        } else MCN_LAUNCH((k_img_acts<CV, NQ, true, 1, 1, UN>), grid, dim3(256), 0, s, p);                        \
    } while (0)
        if (p.uNeuron == MCN_NEURON_RELU) MCN_UNDO_LAUNCH(2);
        else MCN_UNDO_LAUNCH(1);
#undef MCN_UNDO_LAUNCH
        return;
    }
    if (vecn && pipelined) MCN_LAUNCH((k_img_acts<CV, NQ, true, U>), grid, dim3(256), 0, s, p);
    else if (vecn && ksplit) {
        if constexpr (CV == 2 && NQ == 1) {
            grid.y = divup(p.imgY * p.imgX, 4 / ksplit);
            // (two 16-channel tiles per wave halve the re-reads of the gradient rows; 4 steps per register set: conv2+ 92 -> 85 us)
            if (ksplit == 4) MCN_LAUNCH((k_img_acts<2, 1, true, 4, 4>), grid, dim3(256), 0, s, p);
            else MCN_LAUNCH((k_img_acts<2, 1, true, 2, 2>), grid, dim3(256), 0, s, p);
        }
        if constexpr (CV == 1 && NQ == 1) {
            grid.y = divup(p.imgY * p.imgX, 4 / ksplit);  // 4 / KS pixels per block
            // prefetch depth of the pixel-split form: 2 steps per register set (in the DeepID2 step 5 us faster than 1 -- local3
            // 14.2 -> 12.7 us alone; 4 gains nothing more).  MCN_IA_U overrides (A/B).
            static const int uexp = getenv("MCN_IA_U") ? atoi(getenv("MCN_IA_U")) : 2;
            if (ksplit == 4) {
                if (uexp == 4) MCN_LAUNCH((k_img_acts<1, 1, true, 4, 4>), grid, dim3(256), 0, s, p);
                else if (uexp == 2) MCN_LAUNCH((k_img_acts<1, 1, true, 2, 4>), grid, dim3(256), 0, s, p);
                else MCN_LAUNCH((k_img_acts<1, 1, true, 1, 4>), grid, dim3(256), 0, s, p);
            } else {
                if (uexp == 4) MCN_LAUNCH((k_img_acts<1, 1, true, 4, 2>), grid, dim3(256), 0, s, p);
                else if (uexp == 2) MCN_LAUNCH((k_img_acts<1, 1, true, 2, 2>), grid, dim3(256), 0, s, p);
                else MCN_LAUNCH((k_img_acts<1, 1, true, 1, 2>), grid, dim3(256), 0, s, p);
            }
        }
    }
    else if (vecn) MCN_LAUNCH((k_img_acts<CV, NQ, true, 1>), grid, dim3(256), 0, s, p);
    else MCN_LAUNCH((k_img_acts<CV, NQ, false, 1>), grid, dim3(256), 0, s, p);
}

template <int CV>
static void launch_bd1(const BwdDataParams& p, int nq, dim3 grid, bool vecn, bool pipelined, int ksplit, hipStream_t s) {
    if (nq == 2) launch_bd2<CV, 2>(p, grid, vecn, pipelined, 0, s);
    else launch_bd2<CV, 1>(p, grid, vecn, pipelined, ksplit, s);
}

// Per calling thread (mcn_set_thread_option "img_acts_wide"): 4 (default) = the four-waves-per-pixel form for wide layers, the
// fastest launch on an otherwise idle GPU; 0 = one wave per pixel.  A caller that keeps a second stream busy with weight
// gradients sets 0: the wide form fills every wave slot and the other queue's kernels wait (DeepID2+ step: 0.905 ms with
// the wide form on the backward-data chain, 0.817 ms with the narrow one, although each launch alone is 30 % slower).
static thread_local int t_imgActsWide = 4;
MCN_API int mcn_set_thread_option(const char* name, int value) {
    MCN_REQUIRE(name, "setThreadOption: null name");
    if (strcmp(name, "img_acts_wide") == 0) {
        MCN_REQUIRE(value == 0 || value == 2 || value == 4, "setThreadOption: img_acts_wide is 0, 2 or 4");
        t_imgActsWide = value;
        return MCN_OK;
    }
    mcn_set_error("setThreadOption: unknown option '%s'", name);
    return MCN_EINVAL;
}

struct UndoArgs {
    const float* acts;
    const float* maxActs;
    float* target;
    int neuron;
    float na, nb;
};
static int img_acts_impl(const float* hidActs, const float* filters, float* targets, int N, int imgY, int imgX,
                         int C, int F, int fs, int mY, int mX, int pad, int stride, int groups, int conv,
                         float sT, float sO, void* stream, const UndoArgs* undo = nullptr) {
    // host asserts of _imgActs (img_acts.cu:900-940)
    MCN_REQUIRE(hidActs && filters && targets, "imgActs: null pointer");
    MCN_REQUIRE(N > 0 && imgY > 0 && imgX > 0 && C > 0 && F > 0 && fs > 0 && mY > 0 && mX > 0 && stride > 0 && groups > 0,
                "imgActs: non-positive dimension");
    MCN_REQUIRE(C % groups == 0, "imgActs: numImgColors %% numGroups != 0");
    MCN_REQUIRE(F % (16 * groups) == 0, "imgActs: numFilters (%d) must be divisible by 16*numGroups", F);
    MCN_REQUIRE(groups > 1 || (C <= 3 || C % 2 == 0), "imgActs: numImgColors must be <= 3 or even");
    MCN_REQUIRE(groups == 1 || (C / groups) % 4 == 0, "imgActs: filter colors must be divisible by 4 with groups");
    MCN_REQUIRE(pad <= 0, "imgActs: paddingStart must be <= 0");
    MCN_REQUIRE(pad + (mX - 1) * stride + fs >= imgX && pad + (mY - 1) * stride + fs >= imgY,
                "imgActs: modules do not cover the image");
    MCN_REQUIRE(stride <= fs, "imgActs: moduleStride > filterSize");
    MCN_REQUIRE(((uintptr_t)filters % 16) == 0, "imgActs: filters must be 16-byte aligned");

    BwdDataParams p;
    p.hidActs = hidActs; p.filters = filters; p.targets = targets;
    p.N = N; p.imgY = imgY; p.imgX = imgX; p.C = C; p.F = F; p.fs = fs; p.mY = mY; p.mX = mX;
    p.pad = pad; p.stride = stride; p.groups = groups; p.conv = conv; p.sT = sT; p.sO = sO;
    p.uActs = nullptr; p.uMax = nullptr; p.uTarget = nullptr; p.uNeuron = -1; p.uNa = p.uNb = 0.f;
    p.tick = {nullptr, 0};
    p.waveCt = 0;
    static const int xcdPairEnv = getenv("MCN_IA_XCD_PAIR") ? atoi(getenv("MCN_IA_XCD_PAIR")) : 1;
    p.xcdPair = (xcdPairEnv & (conv ? 2 : 1)) != 0 && N > 64;  // bit 0: unshared layers, bit 1: shared-weight layers
    const int fc = C / groups, imgPixels = imgY * imgX;
    const bool vecn = (N % 4 == 0) && (((uintptr_t)hidActs | (uintptr_t)targets) % 16 == 0);
    if (undo) {
        MCN_REQUIRE(undo->acts && undo->maxActs && undo->target, "imgActsMaxUndo: null pointer");
        MCN_REQUIRE(vecn && (((uintptr_t)undo->acts | (uintptr_t)undo->maxActs | (uintptr_t)undo->target) % 16) == 0,
                    "imgActsMaxUndo: numImages must be a multiple of 4 and the matrices 16-byte aligned");
        MCN_REQUIRE(undo->neuron < 0 || undo->neuron == MCN_NEURON_IDENT || undo->neuron == MCN_NEURON_RELU,
                    "imgActsMaxUndo: neuron %d is not built into this epilogue (none, ident, relu)", undo->neuron);
        p.uActs = undo->acts; p.uMax = undo->maxActs; p.uTarget = undo->target; p.uNeuron = undo->neuron; p.uNa = undo->na; p.uNb = undo->nb;
    }
    const int ct16 = divup(fc, 16);
    int CV = ct16 >= 4 && ct16 % 4 == 0 ? 4 : ct16 % 3 == 0 ? 3 : ct16 % 2 == 0 ? 2 : 1;
    int NQ = N > 64 ? 2 : 1;
    auto waves = [&](int cv, int nq) { return (long)divup(N, 64 * nq) * imgPixels * groups * divup(fc, 16 * cv); };
    if (waves(CV, NQ) < 1024 && NQ == 2) NQ = 1;
    while (waves(CV, NQ) < 1024 && CV > 1) CV--;
    dim3 grid(divup(N, 64 * NQ), divup(imgPixels, 4), groups * divup(fc, 16 * CV));
    MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "imgActs: grid too large");
    hipStream_t s = (hipStream_t)stream;
    p.tick = mcn_take_tick(s);
    // few waves per SIMD and few steps per pixel: the launch is bound by each wave's chain of memory round trips ->
    // batched, prefetched steps (measured: local3 18.8 -> 15.6 us; long step chains and shared-weight layers do not gain)
    const int maxSteps = divup(fs, stride) * divup(fs, stride) * (F / groups / 16);
    const bool pipelined = !conv && waves(CV, NQ) < 3 * 1024 && maxSteps <= 32;
    // under two waves per SIMD: two waves per pixel (measured: conv2 22.2 -> 16.7 us, local4 9.6 -> 7.0 us; preferred over
    // the prefetched form where both apply)
    int ksplit = vecn && CV == 1 && NQ == 1 && (F / groups / 16) >= 2 && waves(CV, NQ) < 2048 && divup(imgPixels, 2) <= 65535 ? 2 : 0;
    if (ksplit && (F / groups / 16) >= 4 && waves(CV, NQ) < 1024 && imgPixels <= 65535) ksplit = 4;  // (local4: 7.0 -> 6.4 us)
    // wide layers (an even number of 16-channel tiles, >= 64 filters per group) with few pixels x images: four waves per pixel
    // with TWO channel tiles each instead of one wave per (pixel, 32-64 channels) walking all its steps alone
    // (DeepID2+ conv2: 120 -> 85 us, local3: 116 -> 70 us; MCN_IA_WIDE=0 restores the old choice)
    static const int wideEnv = getenv("MCN_IA_WIDE") ? atoi(getenv("MCN_IA_WIDE")) : -1;
    const int wideKs = wideEnv >= 0 ? wideEnv : t_imgActsWide;
    const bool wideSplit = wideKs != 0;
    if (wideSplit && vecn && ct16 % 2 == 0 && (F / groups / 16) >= 4 && imgPixels <= 65535 &&
        (long)divup(N, 64) * imgPixels * groups * (ct16 / 2) < 2048) {
        CV = 2; NQ = 1; ksplit = wideKs == 2 ? 2 : 4;
        grid = dim3(divup(N, 64), divup(imgPixels, 4), groups * (ct16 / 2));
    }
    // one wave per (pixel, channel tile) and several channel tiles: a block = the channel tiles of one pixel, so the gradient
    // rows (the larger operand, the same for every channel tile) come from L2 once per block instead of once per wave
    static const int waveCtEnv = getenv("MCN_IA_WAVECT") ? atoi(getenv("MCN_IA_WAVECT")) : 1;
    if (waveCtEnv && !ksplit && (groups * divup(fc, 16 * CV)) % 4 == 0 && imgPixels <= 65535) {
        p.waveCt = 1;
        grid = dim3(divup(N, 64 * NQ), imgPixels, groups * divup(fc, 16 * CV) / 4);
    }
    if (CV > 2 || NQ != 1) ksplit = 0;  // (the pixel-split forms exist for one or two channel tiles and 64 images: never take a tick for a launch that does not happen)
    switch (CV) {
        case 4: launch_bd1<4>(p, NQ, grid, vecn, pipelined, 0, s); break;
        case 3: launch_bd1<3>(p, NQ, grid, vecn, pipelined, 0, s); break;
        case 2: launch_bd1<2>(p, NQ, grid, vecn, pipelined && !ksplit, ksplit, s); break;
        default: launch_bd1<1>(p, NQ, grid, vecn, pipelined && !ksplit, ksplit, s); break;
    }
    return MCN_LAUNCH_CHECK("imgActs");
}

MCN_API int mcn_conv_img_acts(const float* hidActs, const float* filters, float* targets, int numImages, int imgSizeY,
                              int imgSizeX, int numImgColors, int numFilters, int filterSize, int numModulesY,
                              int numModulesX, int paddingStart, int moduleStride, int numGroups, float scaleTargets,
                              float scaleOutput, void* stream) {
    return img_acts_impl(hidActs, filters, targets, numImages, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize,
                         numModulesY, numModulesX, paddingStart, moduleStride, numGroups, 1, scaleTargets, scaleOutput, stream);
}

MCN_API int mcn_local_img_acts(const float* hidActs, const float* filters, float* targets, int numImages, int imgSizeY,
                               int imgSizeX, int numImgColors, int numFilters, int filterSize, int numModulesY,
                               int numModulesX, int paddingStart, int moduleStride, int numGroups, float scaleTargets,
                               float scaleOutput, void* stream) {
    return img_acts_impl(hidActs, filters, targets, numImages, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize,
                         numModulesY, numModulesX, paddingStart, moduleStride, numGroups, 0, scaleTargets, scaleOutput, stream);
}

// imgActs of a conv / local layer whose input is a 2x2, stride 2, start 0 max pool over maps of exactly (2 imgSizeY, 2 imgSizeX)
// pixels, continued through that pool and the neuron under it in the epilogue (see k_img_acts UNDO).  poolGrads: the pool
// output's gradient matrix, READ when scaleTargets != 0 (another consumer of the pool wrote its share before), never written.
MCN_API int mcn_img_acts_max_undo_supported(int numImages, int poolSizeX, int poolStart, int poolStride, int poolOutY, int poolOutX,
                                            int poolImgY, int poolImgX, int neuron) {
    return (neuron < 0 || neuron == MCN_NEURON_IDENT || neuron == MCN_NEURON_RELU) && numImages > 0 && numImages % 4 == 0 && poolSizeX == 2 && poolStride == 2 && poolStart == 0 && poolImgY == 2 * poolOutY &&
           poolImgX == 2 * poolOutX;
}
MCN_API int mcn_img_acts_max_undo(const float* hidActs, const float* filters, const float* poolGrads, int numImages, int imgSizeY,
                                  int imgSizeX, int numImgColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                  int paddingStart, int moduleStride, int numGroups, int conv, float scaleTargets, float scaleOutput,
                                  const float* poolInput, const float* poolOutput, float* undoTarget, int neuron, float neuronA,
                                  float neuronB, void* stream) {
    MCN_REQUIRE(scaleTargets == 0.f || poolGrads, "imgActsMaxUndo: scaleTargets != 0 needs the pool's gradient matrix");
    UndoArgs u{poolInput, poolOutput, undoTarget, neuron, neuronA, neuronB};
    // (targets is only dereferenced when scaleTargets != 0; the alignment test wants a pointer either way)
    return img_acts_impl(hidActs, filters, const_cast<float*>(poolGrads ? poolGrads : poolOutput), numImages, imgSizeY, imgSizeX, numImgColors,
                         numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups, conv, scaleTargets,
                         scaleOutput, stream, &u);
}
