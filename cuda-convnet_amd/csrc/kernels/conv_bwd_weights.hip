// weightActs (backward-weights) for shared-weight conv and unshared local layers on fp32 MFMA.
//
// Replaces weight_acts.cu:57-201 / :233-380 (+ host _weightActs :579-976) and, in
// the *_reduced entry point, the (modules/partialSum) x weights temporary plus
// addSum of ConvLayer::bpropWeights (layer.cu:476-496).
//
//   dW[(chunk,) k=(c,p), f] = sum_{m in chunk} sum_n img[c, pix(m,p), n] * hid[f, m, n]
// MFMA roles: A[row=k][kk=n] from image rows, B[kk=n][col=f] from hidActs rows.
// Both operands are contiguous along the REDUCTION index n, whose order inside
// an MFMA chain is free as long as A and B agree: every lane reads 4 consecutive
// images with one 16-byte load and uses component t in step t.  One 4-wave block owns
// 16*KV weight rows x 16*FV filters for one module chunk: its waves take the chunk's
// modules round-robin and their accumulators are summed through LDS in a fixed order.
#include <mutex>
#include "common.h"

struct BwdWParams {
    const float* images;
    const float* hidActs;
    float* targets;
    int N, imgStride, imgY, imgX, C, F, fs, mY, mX, pad, stride, groups;
    int modulesPerChunk, check;
    float sT, sO;
    McnTick tick;  // stored by the launch's first thread (MCN_TICK_PROLOGUE)
    int xcdGroup;  // k_weight_acts_wide: the row / filter blocks of one module chunk on one XCD (mcn_xcd_group)
};

// MODE 0: scalar loads (any N / alignment); 1: 16-byte loads through pointers; 2: 16-byte buffer loads -- lanes outside
// the matrix get an out-of-range offset and read zero, so the loop body has no branches (buffers below 2 GB).
template <int KV, int FV, int MODE>
__global__ __launch_bounds__(256) void k_weight_acts(BwdWParams p) {
    MCN_TICK_PROLOGUE(p);
    constexpr bool VECN = MODE >= 1;
    extern __shared__ int tbl[];
    const int fc = p.C / p.groups, fpg = p.F / p.groups, fp = p.fs * p.fs;
    const int K = fc * fp, M = p.mY * p.mX, imgPixels = p.imgY * p.imgX;
    const int ktiles = (K + 16 * KV - 1) / (16 * KV);
    const int Kpad = ktiles * 16 * KV;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        if (k < K) {
            const int c = k / fp, q = k % fp, fy = q / p.fs, fx = q % p.fs;
            tbl[2 * k] = c * imgPixels + fy * p.imgX + fx;
            tbl[2 * k + 1] = (fy << 16) | fx;
        } else {
            tbl[2 * k] = 0;
            tbl[2 * k + 1] = 0x40004000;
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler: scalar loop control, no exec-masked loops
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    // (mcn_xcd_group -- all blocks of a module chunk on one XCD -- was measured here: no change in L2 misses for conv2, and local3
    // got slower alone, 13.5 -> 15.8 us, its 18 blocks per module queueing on one XCD; only the wide form below uses it)
    const int kt = blockIdx.y, chunk = blockIdx.x, ftile = blockIdx.z;
    const int f0 = ftile * (16 * FV);
    const int grp = f0 / fpg;
    const int kb = kt * 16 * KV;
    const bool check = p.check != 0;

    int off[KV], yx[KV];
#pragma unroll
    for (int v = 0; v < KV; v++) {
        off[v] = tbl[2 * (kb + 16 * v + j)];
        yx[v] = tbl[2 * (kb + 16 * v + j) + 1];
    }
    f32x4 acc[KV][FV];
#pragma unroll
    for (int v = 0; v < KV; v++)
#pragma unroll
        for (int u = 0; u < FV; u++) acc[v][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* ig = p.images + (ptrdiff_t)grp * fc * imgPixels * p.imgStride + 4 * kk;
    const float* hg = p.hidActs + (size_t)(f0 + j) * M * p.N + 4 * kk;

    // the four waves of the block take the chunk's modules round-robin and are summed through LDS below
    const int mBeg = chunk * p.modulesPerChunk, mEnd = min(M, mBeg + p.modulesPerChunk);
    // (chunks of fewer than four modules -- the unshared local layers -- split the images instead)
    const bool splitN = p.modulesPerChunk < 4;
    const int mStep = splitN ? 1 : 4, nStep = splitN ? 64 : 16;
    if constexpr (MODE == 2) {
        const size_t imgBytes = ((size_t)(fc * imgPixels - 1) * p.imgStride + p.N) * 4;
        const __amdgpu_buffer_rsrc_t rImg = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.images + (ptrdiff_t)grp * fc * imgPixels * p.imgStride), 0, (int)imgBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rHid = __builtin_amdgcn_make_buffer_rsrc((void*)p.hidActs, 0, (int)((size_t)p.F * M * p.N * 4), 0x00020000);
        const unsigned OOB = 0xFFFFFFF0u;
        const unsigned hstep = (unsigned)((size_t)16 * M * p.N * 4);
        // (Issuing several iterations' loads at once was measured slower, 24.7 -> 31 us on conv1: the kernel is bound by
        // cache-line requests -- every load touches 16 rows x 64 B -- not by the latency of one round trip.)
        for (int m = mBeg + (splitN ? 0 : wave); m < mEnd; m += mStep) {
            const int y0 = p.pad + (m / p.mX) * p.stride, x0 = p.pad + (m % p.mX) * p.stride;
            unsigned aoff[KV];
            bool ainb[KV];
#pragma unroll
            for (int v = 0; v < KV; v++) {
                ainb[v] = kb + 16 * v + j < K;
                if (check) ainb[v] = ainb[v] && (unsigned)(y0 + (yx[v] >> 16)) < (unsigned)p.imgY && (unsigned)(x0 + (yx[v] & 0xffff)) < (unsigned)p.imgX;
                aoff[v] = (unsigned)((off[v] + y0 * p.imgX + x0) * p.imgStride + 4 * kk) * 4u;
            }
            const unsigned hoff = (unsigned)(((size_t)(f0 + j) * M + m) * p.N + 4 * kk) * 4u;
            for (int nb = splitN ? 16 * wave : 0; nb < p.N; nb += nStep) {
                const bool nin = nb + 4 * kk < p.N;
                f32x4 a[KV], b[FV];
#pragma unroll
                for (int v = 0; v < KV; v++)
                    a[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rImg, (ainb[v] && nin) ? aoff[v] + 4u * nb : OOB, 0, 0));
#pragma unroll
                for (int u = 0; u < FV; u++)
                    b[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rHid, nin ? hoff + u * hstep + 4u * nb : OOB, 0, 0));
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int v = 0; v < KV; v++)
#pragma unroll
                        for (int u = 0; u < FV; u++) acc[v][u] = mfma16(a[v][t], b[u][t], acc[v][u]);
            }
        }
    } else {
    for (int m = mBeg + (splitN ? 0 : wave); m < mEnd; m += mStep) {
            const int y0 = p.pad + (m / p.mX) * p.stride, x0 = p.pad + (m % p.mX) * p.stride;
            const float* arow[KV];
            bool ainb[KV];
#pragma unroll
            for (int v = 0; v < KV; v++) {
                ainb[v] = kb + 16 * v + j < K;
                if (check) ainb[v] = ainb[v] && (unsigned)(y0 + (yx[v] >> 16)) < (unsigned)p.imgY && (unsigned)(x0 + (yx[v] & 0xffff)) < (unsigned)p.imgX;
                arow[v] = ig + ((ptrdiff_t)off[v] + (ptrdiff_t)y0 * p.imgX + x0) * p.imgStride;
            }
            const float* hm = hg + (size_t)m * p.N;
            for (int nb = splitN ? 16 * wave : 0; nb < p.N; nb += nStep) {
                const int n = nb + 4 * kk;
                f32x4 a[KV], b[FV];
#pragma unroll
                for (int v = 0; v < KV; v++) {
                    if (VECN) {
                        a[v] = (ainb[v] && n < p.N) ? *reinterpret_cast<const f32x4*>(arow[v] + nb) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; t++) a[v][t] = (ainb[v] && n + t < p.N) ? arow[v][nb + t] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < FV; u++) {
                    const float* hr = hm + (size_t)(16 * u) * M * p.N + nb;
                    if (VECN) {
                        b[u] = n < p.N ? *reinterpret_cast<const f32x4*>(hr) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; t++) b[u][t] = n + t < p.N ? hr[t] : 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int v = 0; v < KV; v++)
#pragma unroll
                        for (int u = 0; u < FV; u++) acc[v][u] = mfma16(a[v][t], b[u][t], acc[v][u]);
            }
        }
    }

    // fixed-order sum of the four waves' accumulators: wave 0 + wave 1 + wave 2 + wave 3
    __syncthreads();  // everyone is done with the decode table
    float* red = reinterpret_cast<float*>(tbl);
    for (int w = 1; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int v = 0; v < KV; v++)
#pragma unroll
                for (int u = 0; u < FV; u++)
                    *reinterpret_cast<f32x4*>(red + ((v * FV + u) * 64 + lane) * 4) = acc[v][u];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int v = 0; v < KV; v++)
#pragma unroll
                for (int u = 0; u < FV; u++) acc[v][u] += *reinterpret_cast<const f32x4*>(red + ((v * FV + u) * 64 + lane) * 4);
        }
        __syncthreads();
    }
    if (wave != 0) return;

    // D row 4kk+r of tile v <-> weight row kb+16v+4kk+r ; col j of tile u <-> filter f0+16u+j
#pragma unroll
    for (int v = 0; v < KV; v++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = kb + 16 * v + 4 * kk + r;
            if (k >= K) continue;
            float* trow = p.targets + ((size_t)chunk * K + k) * p.F + f0 + j;
#pragma unroll
            for (int u = 0; u < FV; u++) {
                const float o = p.sO * acc[v][u][r];
                trow[16 * u] = p.sT != 0.f ? p.sT * trow[16 * u] + o : o;
            }
        }
}

// Wide layers (128 filters or more, K a multiple of 32 KV: DeepID2+).  One wave owns 16 KV weight rows x 64 filters of one module
// chunk and walks (module, 16 images) as ONE statically pipelined sequence: no split of the images or modules over the waves
// of a block, so no LDS sum and no idle waves at the end; the four waves of a block are 2 row tiles x 2 filter tiles (each
// operand row is asked for by two waves of the same CU).  MFMA roles are swapped against k_weight_acts -- A = gradient rows,
// B = image rows -- so that a lane ends up with four CONSECUTIVE filters of one weight row: the epilogue (and the momentum
// term it reads, local layers) is 16-byte accesses.  One batch of loads in flight, explicit waits (see k_img_acts CHAIN).
template <int KV>
__global__ __launch_bounds__(256) void k_weight_acts_wide(BwdWParams p) {
    MCN_TICK_PROLOGUE(p);
    constexpr int FV = 4;
    extern __shared__ int tbl[];
    const int fp = p.fs * p.fs, K = p.C * fp, M = p.mY * p.mX, imgPixels = p.imgY * p.imgX;
    int wChunk = blockIdx.x, wKt = blockIdx.y, wFt = blockIdx.z;
    if (p.xcdGroup) {  // the blocks of a module chunk read the same gradient rows: one XCD per chunk (L2 misses 77 -> 53 MB per launch)
        int t;
        mcn_xcd_group((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)(gridDim.y * gridDim.z), (int)gridDim.x, wChunk, t);
        wFt = t / (int)gridDim.y;
        wKt = t - wFt * (int)gridDim.y;
    }
    const int kb0 = wKt * (32 * KV);
    for (int i = threadIdx.x; i < 32 * KV; i += 256) {
        const int k = kb0 + i, c = k / fp, q = k % fp, fy = q / p.fs, fx = q % p.fs;
        tbl[2 * i] = c * imgPixels + fy * p.imgX + fx;
        tbl[2 * i + 1] = (fy << 16) | fx;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int chunk = wChunk;
    const int kl = (wave & 1) * 16 * KV, kb = kb0 + kl;
    const int f0 = wFt * 128 + (wave >> 1) * 64;
    const bool check = p.check != 0;
    int off[KV], yx[KV];
#pragma unroll
    for (int v = 0; v < KV; v++) {
        off[v] = tbl[2 * (kl + 16 * v + j)];
        yx[v] = tbl[2 * (kl + 16 * v + j) + 1];
    }
    f32x4 acc[FV][KV];
#pragma unroll
    for (int u = 0; u < FV; u++)
#pragma unroll
        for (int v = 0; v < KV; v++) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const size_t imgBytes = ((size_t)(p.C * imgPixels - 1) * p.imgStride + p.N) * 4;
    const __amdgpu_buffer_rsrc_t rImg = __builtin_amdgcn_make_buffer_rsrc((void*)p.images, 0, (int)imgBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rHid = __builtin_amdgcn_make_buffer_rsrc((void*)p.hidActs, 0, (int)((size_t)p.F * M * p.N * 4), 0x00020000);
    const unsigned OOB = 0xFFFFFFF0u;
    const unsigned hstep = (unsigned)((size_t)16 * M * p.N * 4);
    const int mBeg = chunk * p.modulesPerChunk, mEnd = min(M, mBeg + p.modulesPerChunk);
    const int nIter = (p.N + 15) / 16, total = (mEnd - mBeg) * nIter;

    int lm = mBeg, lnb = 0;  // load cursor: (module, first image of the block of 16)
    auto load = [&](f32x4(&a)[FV], f32x4(&b)[KV]) {
        const int y0 = p.pad + (lm / p.mX) * p.stride, x0 = p.pad + (lm % p.mX) * p.stride;
        const bool nin = lnb + 4 * kk < p.N;
        const unsigned hoff = (unsigned)(((size_t)(f0 + j) * M + lm) * p.N + lnb + 4 * kk) * 4u;
#pragma unroll
        for (int u = 0; u < FV; u++)
            a[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rHid, nin ? hoff + u * hstep : OOB, 0, 0));
#pragma unroll
        for (int v = 0; v < KV; v++) {
            bool inb = nin;
            if (check) inb = inb && (unsigned)(y0 + (yx[v] >> 16)) < (unsigned)p.imgY && (unsigned)(x0 + (yx[v] & 0xffff)) < (unsigned)p.imgX;
            const unsigned aoff = (unsigned)((off[v] + y0 * p.imgX + x0) * p.imgStride + lnb + 4 * kk) * 4u;
            b[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rImg, inb ? aoff : OOB, 0, 0));
        }
        lnb += 16;
        if (lnb >= p.N) {
            lnb = 0;
            if (lm + 1 < mEnd) lm++;  // (past the end: the last module again, never used)
        }
    };
    auto compute = [&](const f32x4(&a)[FV], const f32x4(&b)[KV]) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < FV; u++)
#pragma unroll
                for (int v = 0; v < KV; v++) acc[u][v] = mfma16(a[u][t], b[v][t], acc[u][v]);
    };
    f32x4 a0[FV], b0[KV], a1[FV], b1[KV];
    load(a0, b0);
    for (int it = 0; it < total; it += 2) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): set 0 has arrived, nothing else is in flight
        __builtin_amdgcn_sched_barrier(0);
        load(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        load(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < total) compute(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // acc[u][v][r] = dW[k = kb + 16 v + j][f = f0 + 16 u + 4 kk + r]
    float* tb = p.targets + ((size_t)chunk * K + kb + j) * p.F + f0 + 4 * kk;
    if (p.sT != 0.f) {
        f32x4 old[FV][KV];
#pragma unroll
        for (int u = 0; u < FV; u++)
#pragma unroll
            for (int v = 0; v < KV; v++) old[u][v] = *reinterpret_cast<const f32x4*>(tb + (size_t)16 * v * p.F + 16 * u);
#pragma unroll
        for (int u = 0; u < FV; u++)
#pragma unroll
            for (int v = 0; v < KV; v++) *reinterpret_cast<f32x4*>(tb + (size_t)16 * v * p.F + 16 * u) = p.sT * old[u][v] + p.sO * acc[u][v];
    } else {
#pragma unroll
        for (int u = 0; u < FV; u++)
#pragma unroll
            for (int v = 0; v < KV; v++) *reinterpret_cast<f32x4*>(tb + (size_t)16 * v * p.F + 16 * u) = p.sO * acc[u][v];
    }
}

// column sums of the (chunks x cols) partials in a fixed order: grad = sT*grad + sum
// One block per 16 columns: thread (r = tid/16, c = tid%16) sums rows r, r+16, ... in four interleaved
// running sums, then the 16 row groups are combined through LDS in a fixed tree.
__global__ __launch_bounds__(256) void k_reduce_chunks(const float* __restrict__ part, float* __restrict__ grad,
                                                       int chunks, int cols, float sT) {
    __shared__ float sh[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        // (60 blocks for conv1's 960 weights: a launch of pure latency.  All rows of a thread -- up to 32 per trip, conv1 has
        // 468 strips = 30 -- are requested before the first add; four interleaved sums, rows in index order)
        for (int r = rg; r < chunks; r += 512) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; u++) v[u] = r + 16 * u < chunks ? part[(size_t)(r + 16 * u) * cols + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 32; u += 4) {
                s0 += v[u];
                s1 += v[u + 1];
                s2 += v[u + 2];
                s3 += v[u + 3];
            }
        }
    }
    sh[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 8; o > 0; o >>= 1) {
        if (rg < o) sh[rg][cl] += sh[rg + o][cl];
        __syncthreads();
    }
    if (rg == 0 && c < cols) grad[c] = sT != 0.f ? sT * grad[c] + sh[0][cl] : sh[0][cl];
}

template <int KV, int FV>
static void launch_bw2(const BwdWParams& p, dim3 grid, size_t lds, bool vecn, hipStream_t s) {
    const size_t imgBytes = (size_t)p.C * p.imgY * p.imgX * p.imgStride * 4, hidBytes = (size_t)p.F * p.mY * p.mX * p.N * 4;
    const bool buf = vecn && imgBytes < (1u << 31) && hidBytes < (1u << 31);
    if (buf) MCN_LAUNCH((k_weight_acts<KV, FV, 2>), grid, dim3(256), lds, s, p);
    else if (vecn) MCN_LAUNCH((k_weight_acts<KV, FV, 1>), grid, dim3(256), lds, s, p);
    else MCN_LAUNCH((k_weight_acts<KV, FV, 0>), grid, dim3(256), lds, s, p);
}

static int weight_acts_check(const float* images, const float* hidActs, float* targets, int N, int imgStride,
                             int imgY, int imgX, int C, int F, int fs, int mY, int mX, int pad, int stride, int groups) {
    // host asserts of _weightActs (weight_acts.cu:579-620)
    MCN_REQUIRE(images && hidActs && targets, "weightActs: null pointer");
    MCN_REQUIRE(N > 0 && imgY > 0 && imgX > 0 && C > 0 && F > 0 && fs > 0 && mY > 0 && mX > 0 && stride > 0 && groups > 0,
                "weightActs: non-positive dimension");
    MCN_REQUIRE(imgStride >= N, "weightActs: imgStride (%d) < numImages (%d)", imgStride, N);
    MCN_REQUIRE(C % groups == 0, "weightActs: numImgColors %% numGroups != 0");
    MCN_REQUIRE(F % (16 * groups) == 0, "weightActs: numFilters (%d) must be divisible by 16*numGroups", F);
    MCN_REQUIRE(groups > 1 || (C <= 3 || C % 2 == 0), "weightActs: numImgColors must be <= 3 or even");
    MCN_REQUIRE(groups == 1 || (C / groups) % 4 == 0, "weightActs: filter colors must be divisible by 4 with groups");
    MCN_REQUIRE(pad <= 0, "weightActs: paddingStart must be <= 0");
    MCN_REQUIRE(pad + (mX - 1) * stride + fs >= imgX && pad + (mY - 1) * stride + fs >= imgY,
                "weightActs: modules do not cover the image");
    MCN_REQUIRE(stride <= fs, "weightActs: moduleStride > filterSize");
    return MCN_OK;
}

// k_weight_acts_wide applies (given 16-byte operands): 16 KV rows per wave, 0 = no
static int wide_kv(int K, int F, int groups) {
    static const int off = getenv("MCN_WA_WIDE") ? !atoi(getenv("MCN_WA_WIDE")) : 0;
    if (off || groups != 1 || F % 128 != 0) return 0;
    return K % 96 == 0 ? 3 : K % 128 == 0 ? 4 : 0;
}

// modulesPerChunk modules are summed per output slab; targets has divup(M, modulesPerChunk) slabs.
static int weight_acts_launch(const float* images, const float* hidActs, float* targets, int N, int imgStride,
                              int imgY, int imgX, int C, int F, int fs, int mY, int mX, int pad, int stride,
                              int groups, int modulesPerChunk, float sT, float sO, hipStream_t s) {
    BwdWParams p;
    p.images = images; p.hidActs = hidActs; p.targets = targets;
    p.N = N; p.imgStride = imgStride; p.imgY = imgY; p.imgX = imgX; p.C = C; p.F = F; p.fs = fs;
    p.mY = mY; p.mX = mX; p.pad = pad; p.stride = stride; p.groups = groups;
    p.modulesPerChunk = modulesPerChunk; p.sT = sT; p.sO = sO;
    p.tick = {nullptr, 0};
    static const int xcdGroupEnv = getenv("MCN_WA_XCD_GROUP") ? atoi(getenv("MCN_WA_XCD_GROUP")) : 1;
    p.xcdGroup = xcdGroupEnv;
    const int fc = C / groups, K = fc * fs * fs, M = mY * mX, fpg = F / groups;
    p.check = (pad < 0) || (pad + (mX - 1) * stride + fs > imgX) || (pad + (mY - 1) * stride + fs > imgY);
    const bool vecn = (N % 4 == 0) && (imgStride % 4 == 0) && (((uintptr_t)images | (uintptr_t)hidActs) % 16 == 0);
    const int chunks = divup(M, modulesPerChunk);
    const int ftiles = fpg / 16;
    int FV = ftiles % 2 == 0 ? 2 : 1;
    const int kt16 = divup(K, 16);
    int KV = kt16 >= 3 ? 3 : kt16;
    auto blocks = [&](int kv, int fv) { return (long)chunks * divup(K, 16 * kv) * (F / (16 * fv)); };
    while (blocks(KV, FV) < 512 && KV > 1) KV--;
    if (blocks(KV, FV) < 512 && FV == 2) FV = 1;
    const int ktiles = divup(K, 16 * KV);
    dim3 grid(chunks, ktiles, F / (16 * FV));
    MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "weightActs: grid too large");
    size_t lds = (size_t)ktiles * 16 * KV * 8;  // decode table, reused for the cross-wave reduction
    if (lds < (size_t)KV * FV * 64 * 16) lds = (size_t)KV * FV * 64 * 16;
    MCN_REQUIRE(lds <= 64 * 1024, "weightActs: filter too large for the decode table (%zu bytes)", lds);
    p.tick = mcn_take_tick(s);
    const int wkv = wide_kv(K, F, groups);
    if (wkv && (long)chunks * (K / (32 * wkv)) * (F / 128) >= 128) {  // (fewer blocks than that: the split forms below fill the chip better)
        const size_t imgBytes = (size_t)C * imgY * imgX * imgStride * 4, hidBytes = (size_t)F * M * N * 4;
        if (vecn && imgBytes < (1u << 31) && hidBytes < (1u << 31) && ((uintptr_t)targets % 16) == 0) {
            dim3 wgrid(chunks, K / (32 * wkv), F / 128);
            MCN_REQUIRE(wgrid.y <= 65535 && wgrid.z <= 65535, "weightActs: grid too large");
            if (wkv == 3) MCN_LAUNCH((k_weight_acts_wide<3>), wgrid, dim3(256), 32 * 3 * 8, s, p);
            else MCN_LAUNCH((k_weight_acts_wide<4>), wgrid, dim3(256), 32 * 4 * 8, s, p);
            return MCN_LAUNCH_CHECK("weightActs");
        }
    }
    if (KV == 3 && FV == 2) launch_bw2<3, 2>(p, grid, lds, vecn, s);
    else if (KV == 3) launch_bw2<3, 1>(p, grid, lds, vecn, s);
    else if (KV == 2 && FV == 2) launch_bw2<2, 2>(p, grid, lds, vecn, s);
    else if (KV == 2) launch_bw2<2, 1>(p, grid, lds, vecn, s);
    else if (FV == 2) launch_bw2<1, 2>(p, grid, lds, vecn, s);
    else launch_bw2<1, 1>(p, grid, lds, vecn, s);
    return MCN_LAUNCH_CHECK("weightActs");
}

MCN_API int mcn_conv_weight_acts(const float* images, const float* hidActs, float* targets, int numImages,
                                 int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                 int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                 int numGroups, int partialSum, float scaleTargets, float scaleOutput, void* stream) {
    int rc = weight_acts_check(images, hidActs, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                               numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups);
    if (rc) return rc;
    const int M = numModulesY * numModulesX;
    if (partialSum <= 0) partialSum = M;
    MCN_REQUIRE(M % partialSum == 0, "convWeightActs: partialSum (%d) must divide the number of modules (%d)", partialSum, M);
    return weight_acts_launch(images, hidActs, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                              numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups,
                              partialSum, scaleTargets, scaleOutput, (hipStream_t)stream);
}

MCN_API int mcn_local_weight_acts(const float* images, const float* hidActs, float* targets, int numImages,
                                  int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                  int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                  int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    int rc = weight_acts_check(images, hidActs, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                               numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups);
    if (rc) return rc;
    return weight_acts_launch(images, hidActs, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                              numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups,
                              1, scaleTargets, scaleOutput, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Shared-weight conv, strip form.  The per-module kernel above reads every 4x4 (3x3) image patch again for every
// module: 66 MB of patch rows for conv1 against 4.6 MB of image (PMC: 58 MB of HBM-side traffic per launch).  Here a
// block owns a STRIP of MS consecutive modules of one module row, CB channels and 16*FV filters, and walks the images
// in stages of 32: the (CB * fs * W) image rows the strip touches (W = (MS-1)*stride + fs columns) and the strip's
// gradient rows are loaded once per stage with full 128-byte lines into LDS, and every module's MFMA operands are
// read from there.  Next stage's global loads are in flight (registers) while the current stage computes.
// Waves take the strip's modules round-robin; their accumulators are summed through LDS in a fixed order.
// Output: one (K, F) slab per strip, reduced by k_reduce_chunks.
struct StripParams {
    const float* images;
    const float* hidActs;
    float* targets;
    int N, imgStride, imgY, imgX, C, F, fs, mY, mX, pad, stride;
    int MS, CB, W, stripsPerRow;
    float sO;
    McnTick tick;  // stored by the launch's first thread (MCN_TICK_PROLOGUE)
    // optional: the strip's share of the shared-bias gradient, sB * sum over the strip's modules and all images of the
    // gradient rows it stages anyway -> biasPart[strip][F] (the row sum over the whole gradient matrix -- 22 MB for DeepID's
    // conv1 -- then never runs; the update launch adds the strips' shares, mcn_sgd_plan_build_fused kind 1)
    float* biasPart;
    float sB;
};

// (Two stages of global loads in flight -- two register sets, unconditional out-of-range-masked loads so that the compiler waits for
// the older set alone, which it then does -- were measured in round 3: conv1 17.0 -> 19.7 us, conv1+ 25.0 -> 29.3.  The stages are not
// waiting for a round trip; left at one stage in flight.)
template <int KT, int FV>
__global__ __launch_bounds__(256) void k_conv_wgrad_strip(const StripParams p) {
    extern __shared__ float lds[];
    MCN_TICK_PROLOGUE(p);
    constexpr int RS = 36;     // floats per LDS row: 32 images + 4 of padding (bank spread)
    constexpr int PI = 8;      // image-row load passes (32 rows each) a thread can hold
    constexpr int PH = 4 * FV; // gradient-row load passes (MS <= 8)
    const int fp = p.fs * p.fs, Kb = p.CB * fp, K = p.C * fp, M = p.mY * p.mX, imgPixels = p.imgY * p.imgX;
    const int R = p.CB * p.fs * p.W;  // image rows of the tile; row R stays zero
    const int HR = 16 * FV * p.MS;
    float* imgT = lds;
    float* hidT = lds + (size_t)(R + 1) * RS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int strip = blockIdx.x, my = strip / p.stripsPerRow, mx0 = (strip % p.stripsPerRow) * p.MS;
    const int msCount = min(p.MS, p.mX - mx0);
    const int c0 = blockIdx.y * p.CB, f0 = blockIdx.z * (16 * FV);
    const int y0 = p.pad + my * p.stride, x0 = p.pad + mx0 * p.stride;

    const __amdgpu_buffer_rsrc_t rImg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.images, 0, (int)(((size_t)(p.C * imgPixels - 1) * p.imgStride + p.N) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rHid = __builtin_amdgcn_make_buffer_rsrc((void*)p.hidActs, 0, (int)((size_t)p.F * M * p.N * 4), 0x00020000);
    const unsigned OOB = 0xFFFFFFF0u;

    // global -> LDS: thread (row r0 + 32*pass, 16-byte column l8) of the two tiles
    const int l8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    unsigned ioff[PI], hoff[PH];
#pragma unroll
    for (int q = 0; q < PI; q++) {
        const int r = r0 + 32 * q;
        ioff[q] = OOB;
        if (r < R) {
            const int c = r / (p.fs * p.W), rem = r % (p.fs * p.W), fy = rem / p.W, col = rem % p.W;
            const int y = y0 + fy, x = x0 + col;
            if ((unsigned)y < (unsigned)p.imgY && (unsigned)x < (unsigned)p.imgX)
                ioff[q] = (unsigned)(((c0 + c) * imgPixels + y * p.imgX + x) * p.imgStride + 4 * l8) * 4u;
        }
    }
#pragma unroll
    for (int q = 0; q < PH; q++) {
        const int r = r0 + 32 * q;
        hoff[q] = OOB;
        if (r < HR) {
            const int f = r / p.MS, ms = r % p.MS;
            if (ms < msCount) hoff[q] = (unsigned)(((size_t)(f0 + f) * M + my * p.mX + mx0 + ms) * p.N + 4 * l8) * 4u;
        }
    }
    if (threadIdx.x < RS) imgT[(size_t)R * RS + threadIdx.x] = 0.f;

    // MFMA operand rows: tile v, lane row j <-> weight row k = 16v + j of this channel block
    int rb[KT];
    bool rv[KT];
#pragma unroll
    for (int v = 0; v < KT; v++) {
        const int k = 16 * v + j;
        rv[v] = k < Kb;
        const int c = k / fp, q = k % fp, fy = q / p.fs, fx = q % p.fs;
        rb[v] = (c * p.fs + fy) * p.W + fx;
    }
    f32x4 acc[KT][FV];
#pragma unroll
    for (int v = 0; v < KT; v++)
#pragma unroll
        for (int u = 0; u < FV; u++) acc[v][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float bsum = 0.f;  // thread r < HR: running sum of gradient row r (filter r / MS, module r % MS) over the images
    const bool doBias = p.biasPart != nullptr && blockIdx.y == 0;
    f32x4 ireg[PI], hreg[PH];
    auto load_stage = [&](int st) {
        const unsigned nbyte = (unsigned)st * 128u;
#pragma unroll
        for (int q = 0; q < PI; q++)
            if (32 * q < R)  // block-uniform
                ireg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rImg, ioff[q] == OOB ? OOB : ioff[q] + nbyte, 0, 0));
#pragma unroll
        for (int q = 0; q < PH; q++)
            if (32 * q < HR)
                hreg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rHid, hoff[q] == OOB ? OOB : hoff[q] + nbyte, 0, 0));
    };
    const int stages = p.N / 32;
    load_stage(0);
    for (int st = 0; st < stages; st++) {
        __syncthreads();  // the previous stage's operands have been read
#pragma unroll
        for (int q = 0; q < PI; q++)
            if (r0 + 32 * q < R) *reinterpret_cast<f32x4*>(imgT + (size_t)(r0 + 32 * q) * RS + 4 * l8) = ireg[q];
#pragma unroll
        for (int q = 0; q < PH; q++)
            if (r0 + 32 * q < HR) *reinterpret_cast<f32x4*>(hidT + (size_t)(r0 + 32 * q) * RS + 4 * l8) = hreg[q];
        __syncthreads();
        if (st + 1 < stages) load_stage(st + 1);
        if (doBias && (int)threadIdx.x < HR) {
            const float* hr = hidT + (size_t)threadIdx.x * RS;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(hr + 4 * q);
                bsum += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
        for (int ms = wave; ms < msCount; ms += 4) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f32x4 a[KT], b[FV];
#pragma unroll
                for (int v = 0; v < KT; v++)
                    a[v] = *reinterpret_cast<const f32x4*>(imgT + (size_t)(rv[v] ? rb[v] + ms * p.stride : R) * RS + 16 * i + 4 * kk);
#pragma unroll
                for (int u = 0; u < FV; u++)
                    b[u] = *reinterpret_cast<const f32x4*>(hidT + (size_t)((16 * u + j) * p.MS + ms) * RS + 16 * i + 4 * kk);
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int v = 0; v < KT; v++)
#pragma unroll
                        for (int u = 0; u < FV; u++) acc[v][u] = mfma16(a[v][t], b[u][t], acc[v][u]);
            }
        }
    }

    if (doBias) {  // modules of the strip in index order
        __syncthreads();
        if ((int)threadIdx.x < HR) lds[threadIdx.x] = bsum;
        __syncthreads();
        if ((int)threadIdx.x < 16 * FV) {
            float sum = 0.f;
            for (int ms = 0; ms < msCount; ms++) sum += lds[threadIdx.x * p.MS + ms];
            p.biasPart[(size_t)strip * p.F + f0 + threadIdx.x] = p.sB * sum;
        }
    }
    // fixed-order sum of the four waves' accumulators: wave 0 + wave 1 + wave 2 + wave 3
    __syncthreads();
    for (int w = 1; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int v = 0; v < KT; v++)
#pragma unroll
                for (int u = 0; u < FV; u++) *reinterpret_cast<f32x4*>(lds + ((v * FV + u) * 64 + lane) * 4) = acc[v][u];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int v = 0; v < KT; v++)
#pragma unroll
                for (int u = 0; u < FV; u++) acc[v][u] += *reinterpret_cast<const f32x4*>(lds + ((v * FV + u) * 64 + lane) * 4);
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int v = 0; v < KT; v++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kl = 16 * v + 4 * kk + r;
            if (kl >= Kb) continue;
            float* trow = p.targets + ((size_t)strip * K + c0 * fp + kl) * p.F + f0 + j;
#pragma unroll
            for (int u = 0; u < FV; u++) trow[16 * u] = p.sO * acc[v][u][r];
        }
}

struct StripPlan { bool ok; int MS, CB, FV, KT, W, stripsPerRow, strips; size_t lds; };
static StripPlan strip_plan(int C, int F, int fs, int mY, int mX, int stride, int groups) {
    StripPlan pl = {};
    if (groups != 1 || F % 16 != 0) return pl;
    pl.MS = std::min(6, mX);  // measured on conv1 (tools/kbench.py): 8 -> 18.9 us, 6 -> 18.1, 4 -> 18.9, 2 -> 26.7
    pl.W = (pl.MS - 1) * stride + fs;
    pl.stripsPerRow = divup(mX, pl.MS);
    pl.strips = mY * pl.stripsPerRow;
    // all channels in one block: with channel blocks the gradient rows are read once per block and the 16-row tiles
    // fill badly (conv2: 19.3 us per module vs 25.8 us in strips), so deeper layers stay with the per-module kernel
    for (int fv = ((F / 16) % 2 == 0 ? 2 : 1); fv >= 1 && !pl.ok; fv--)
        for (int cb = C; cb >= C; cb--) {
            const int Kb = cb * fs * fs, KT = divup(Kb, 16), R = cb * fs * pl.W;
            const size_t lds = std::max((size_t)(R + 1 + 16 * fv * pl.MS) * 36 * 4, (size_t)KT * fv * 64 * 16);
            if (KT > 5 || R > 256 || lds > 60 * 1024) continue;
            pl.ok = true; pl.CB = cb; pl.FV = fv; pl.KT = KT; pl.lds = lds;
            break;
        }
    return pl;
}

template <int KT>
static void launch_strip(const StripParams& p, const StripPlan& pl, dim3 grid, hipStream_t s) {
    if (pl.FV == 2) MCN_LAUNCH((k_conv_wgrad_strip<KT, 2>), grid, dim3(256), pl.lds, s, p);
    else MCN_LAUNCH((k_conv_wgrad_strip<KT, 1>), grid, dim3(256), pl.lds, s, p);
}

// module split of the fused reduction: enough chunks to fill the chip, few enough to keep the
// second pass negligible
static int reduced_modules_per_chunk(int M, int K, int F) {
    if (const int wkv = wide_kv(K, F, 1)) {  // a block of the wide form = 32 KV rows x 128 filters of a chunk; every wave walks the whole chunk
        static const int wantBlocks = getenv("MCN_WA_WIDE_BLOCKS") ? atoi(getenv("MCN_WA_WIDE_BLOCKS")) : 384;
        const int perChunk = (K / (32 * wkv)) * (F / 128);
        int chunks = std::max(1, std::min(M, divup(wantBlocks, perChunk)));
        return divup(M, chunks);
    }
    const long tilesPerChunk = (long)divup(K, 48) * (F / 16);
    long want = divup(1024, (int)(tilesPerChunk > 0 ? tilesPerChunk : 1));  // chunks wanted (a chunk is a 4-wave block per tile)
    if (want > M / 4) want = M / 4 > 0 ? M / 4 : 1;                         // at least one module per wave (measured: conv2 23.1 -> 19.4 us vs two)
    if (want < 1) want = 1;
    if (want > M) want = M;
    return divup(M, (int)want);
}

MCN_API size_t mcn_conv_weight_acts_reduced_workspace_bytes(int numImgColors, int numFilters, int filterSize,
                                                            int numModulesY, int numModulesX, int numGroups) {
    const int K = (numImgColors / numGroups) * filterSize * filterSize, M = numModulesY * numModulesX;
    const int mpc = reduced_modules_per_chunk(M, K, numFilters);
    const StripPlan pl = strip_plan(numImgColors, numFilters, filterSize, numModulesY, numModulesX, 1, numGroups);
    const size_t chunks = std::max<size_t>(divup(M, mpc), pl.ok ? pl.strips : 0);  // either form may be chosen at launch
    return chunks * K * numFilters * sizeof(float);
}

static int conv_weight_acts_reduced_impl(const float* images, const float* hidActs, float* grad, void* workspace,
                                         size_t workspaceBytes, int numImages, int imgStride, int imgSizeY,
                                         int imgSizeX, int numImgColors, int numFilters, int filterSize,
                                         int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                         int numGroups, float scaleTargets, float scaleOutput, void* stream, int* chunksOut = nullptr,
                                         float* biasPart = nullptr, float biasScale = 0.f, int* biasDone = nullptr) {
    // chunksOut: first stage only -- the partials stay in `workspace`, their count goes to *chunksOut, grad is not touched
    if (biasDone) *biasDone = 0;
    int rc = weight_acts_check(images, hidActs, grad ? grad : (float*)workspace, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                               numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups);
    if (rc) return rc;
    const int K = (numImgColors / numGroups) * filterSize * filterSize, M = numModulesY * numModulesX;
    const int cols = K * numFilters;
    // strip form: unit-stride conv, 16-byte image accesses, whole stages of 32 images, buffers within 2 GB
    const StripPlan pl = strip_plan(numImgColors, numFilters, filterSize, numModulesY, numModulesX, moduleStride, numGroups);
    const bool vecn = (numImages % 4 == 0) && (imgStride % 4 == 0) && (((uintptr_t)images | (uintptr_t)hidActs) % 16 == 0);
    const size_t imgBytes = (size_t)numImgColors * imgSizeY * imgSizeX * imgStride * 4, hidBytes = (size_t)numFilters * M * numImages * 4;
    if (pl.ok && moduleStride == 1 && vecn && numImages % 32 == 0 && imgBytes < (1u << 31) && hidBytes < (1u << 31)) {
        const size_t needS = (size_t)pl.strips * cols * sizeof(float);
        if (workspaceBytes < needS || !workspace) {
            mcn_set_error("convWeightActsReduced: workspace %zu < %zu bytes", workspaceBytes, needS);
            return MCN_EWORKSPACE;
        }
        StripParams sp;
        sp.images = images; sp.hidActs = hidActs; sp.targets = (float*)workspace;
        sp.N = numImages; sp.imgStride = imgStride; sp.imgY = imgSizeY; sp.imgX = imgSizeX; sp.C = numImgColors; sp.F = numFilters;
        sp.fs = filterSize; sp.mY = numModulesY; sp.mX = numModulesX; sp.pad = paddingStart; sp.stride = moduleStride;
        sp.MS = pl.MS; sp.CB = pl.CB; sp.W = pl.W; sp.stripsPerRow = pl.stripsPerRow; sp.sO = scaleOutput;
        sp.biasPart = (chunksOut && biasDone) ? biasPart : nullptr; sp.sB = biasScale;
        if (sp.biasPart) *biasDone = 1;
        dim3 grid(pl.strips, numImgColors / pl.CB, numFilters / (16 * pl.FV));
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "convWeightActsReduced: grid too large");
        hipStream_t s = (hipStream_t)stream;
        sp.tick = mcn_take_tick(s);
        switch (pl.KT) {
            case 1: launch_strip<1>(sp, pl, grid, s); break;
            case 2: launch_strip<2>(sp, pl, grid, s); break;
            case 3: launch_strip<3>(sp, pl, grid, s); break;
            case 4: launch_strip<4>(sp, pl, grid, s); break;
            default: launch_strip<5>(sp, pl, grid, s); break;
        }
        rc = MCN_LAUNCH_CHECK("convWeightActsReduced");
        if (rc) return rc;
        if (chunksOut) { *chunksOut = pl.strips; return MCN_OK; }
        MCN_LAUNCH(k_reduce_chunks, dim3(divup(cols, 16)), dim3(256), 0, s, (const float*)workspace, grad, pl.strips, cols, scaleTargets);
        return MCN_LAUNCH_CHECK("convWeightActsReduced");
    }
    const int mpc = reduced_modules_per_chunk(M, K, numFilters);
    const int chunks = divup(M, mpc);
    const size_t need = (size_t)chunks * K * numFilters * sizeof(float);
    if (workspaceBytes < need || !workspace) {
        mcn_set_error("convWeightActsReduced: workspace %zu < %zu bytes", workspaceBytes, need);
        return MCN_EWORKSPACE;
    }
    rc = weight_acts_launch(images, hidActs, (float*)workspace, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                            numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups,
                            mpc, 0.f, scaleOutput, (hipStream_t)stream);
    if (rc) return rc;
    if (chunksOut) { *chunksOut = chunks; return MCN_OK; }
    MCN_LAUNCH(k_reduce_chunks, dim3(divup(cols, 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, grad, chunks, cols, scaleTargets);
    return MCN_LAUNCH_CHECK("convWeightActsReduced");
}

MCN_API int mcn_conv_weight_acts_reduced(const float* images, const float* hidActs, float* grad, void* workspace,
                                         size_t workspaceBytes, int numImages, int imgStride, int imgSizeY,
                                         int imgSizeX, int numImgColors, int numFilters, int filterSize,
                                         int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                         int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return conv_weight_acts_reduced_impl(images, hidActs, grad, workspace, workspaceBytes, numImages, imgStride, imgSizeY, imgSizeX,
                                         numImgColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                                         numGroups, scaleTargets, scaleOutput, stream);
}

// number of slabs mcn_conv_weight_acts_partials produces for 16-byte aligned operands of this shape
MCN_API int mcn_conv_weight_acts_partials_count(int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                                int filterSize, int numModulesY, int numModulesX, int moduleStride, int numGroups,
                                                int* biasCapable) {
    if (biasCapable) *biasCapable = 0;
    if (numImages <= 0 || numImgColors <= 0 || numFilters <= 0 || filterSize <= 0 || numModulesY <= 0 || numModulesX <= 0 || numGroups <= 0 ||
        numImgColors % numGroups != 0)
        return 0;
    const int K = (numImgColors / numGroups) * filterSize * filterSize, M = numModulesY * numModulesX;
    const StripPlan pl = strip_plan(numImgColors, numFilters, filterSize, numModulesY, numModulesX, moduleStride, numGroups);
    const bool vecn = (numImages % 4 == 0) && (imgStride % 4 == 0);
    const size_t imgBytes = (size_t)numImgColors * imgSizeY * imgSizeX * imgStride * 4, hidBytes = (size_t)numFilters * M * numImages * 4;
    if (pl.ok && moduleStride == 1 && vecn && numImages % 32 == 0 && imgBytes < (1u << 31) && hidBytes < (1u << 31)) {
        if (biasCapable) *biasCapable = 1;
        return pl.strips;
    }
    return divup(M, reduced_modules_per_chunk(M, K, numFilters));
}

MCN_API int mcn_conv_weight_acts_partials(const float* images, const float* hidActs, void* partials, size_t partialsBytes,
                                          int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                          int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                          int numGroups, float scaleOutput, int* numChunks, float* biasPartials, float biasScale,
                                          int* biasDone, void* stream) {
    MCN_REQUIRE(partials && numChunks, "convWeightActsPartials: null pointer");
    MCN_REQUIRE(!biasPartials || biasDone, "convWeightActsPartials: biasPartials without biasDone");
    return conv_weight_acts_reduced_impl(images, hidActs, nullptr, partials, partialsBytes, numImages, imgStride, imgSizeY, imgSizeX,
                                         numImgColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                                         numGroups, 0.f, scaleOutput, stream, numChunks, biasPartials, biasScale, biasDone);
}

// ---------------------------------------------------------------------------------------------
// The same call for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): the module-chunk partials go to a file-scope device array instead of a caller workspace.  Every GPU the
// library runs on holds its own copy of the array at its own address until the process ends; the address is looked up
// once per device (hipGetSymbolAddress, at the first call with that device current) and only that pointer is ever
// handed to kernels -- the array's name is a host-side stand-in, not a device address.
// Nothing has to start at zero: both partial kernels write every (chunk, k, f) slot the reduce pass reads, so the
// contents the previous call left are simply overwritten.  Calls that overlap on one GPU race on the array.
#define MCN_WA_SCRATCH_BYTES (4u << 20)
#define MCN_MAX_DEVICES 64
static __device__ __align__(256) float scratch[MCN_WA_SCRATCH_BYTES / sizeof(float)];
static std::mutex g_scratchMutex;
static float* g_scratchPtr[MCN_MAX_DEVICES];  // per device; null = not looked up yet

static int scratch_for_current_device(float** out) {
    int dev = -1;
    int rc = mcn_check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    MCN_REQUIRE(dev >= 0 && dev < MCN_MAX_DEVICES, "convWeightActsReducedStatic: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_scratchMutex);
    if (!g_scratchPtr[dev]) {
        void* ptr = nullptr;
        rc = mcn_check_hip(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
        if (rc) return rc;
        MCN_REQUIRE(ptr != nullptr, "convWeightActsReducedStatic: no device address for the scratch array");
        g_scratchPtr[dev] = (float*)ptr;
    }
    *out = g_scratchPtr[dev];
    return MCN_OK;
}

MCN_API size_t mcn_conv_weight_acts_reduced_static_max_bytes(void) { return MCN_WA_SCRATCH_BYTES; }

MCN_API int mcn_conv_weight_acts_reduced_static(const float* images, const float* hidActs, float* grad, int numImages,
                                                int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                                int filterSize, int numModulesY, int numModulesX, int paddingStart,
                                                int moduleStride, int numGroups, float scaleTargets, float scaleOutput,
                                                void* stream) {
    MCN_REQUIRE(numImgColors > 0 && numFilters > 0 && filterSize > 0 && numModulesY > 0 && numModulesX > 0 && numGroups > 0,
                "convWeightActsReducedStatic: non-positive dimension");
    const size_t need = mcn_conv_weight_acts_reduced_workspace_bytes(numImgColors, numFilters, filterSize, numModulesY, numModulesX, numGroups);
    if (need > MCN_WA_SCRATCH_BYTES) {
        mcn_set_error("convWeightActsReducedStatic: this shape needs %zu bytes of partials, the static scratch holds %u "
                      "(use mcn_conv_weight_acts_reduced with a caller workspace)", need, (unsigned)MCN_WA_SCRATCH_BYTES);
        return MCN_EWORKSPACE;
    }
    float* ptr = nullptr;
    int rc = scratch_for_current_device(&ptr);
    if (rc) return rc;
    return conv_weight_acts_reduced_impl(images, hidActs, grad, ptr, MCN_WA_SCRATCH_BYTES, numImages, imgStride, imgSizeY, imgSizeX,
                                         numImgColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                                         numGroups, scaleTargets, scaleOutput, stream);
}
