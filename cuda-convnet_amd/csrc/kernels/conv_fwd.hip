// filterActs for shared-weight conv and unshared local layers on fp32 MFMA.
//
// Replaces filter_acts.cu:56-192 / :223-368 (+ host _filterActs :565-1180) and,
// through the fused epilogue, the bias addVector of layer.cu:453-461 / :540-542
// and the in-place neuron of layer.cu:249.
//
// Implicit GEMM per module m:  out[f, n] = sum_k W[(m,) k, f] * patch_m[k, n],
// k = (c, fy, fx).  One wave owns (one module) x (16*FV filters) x (64*NQ images); for layers with few
// modules (local layers) the four waves of a block split K of one such tile and are summed through LDS.
// v_mfma_f32_16x16x4_f32 operands come straight from the reference layouts:
//   A[row=f][k]  : lane (i=l&15, kk=l>>4) reads FV consecutive filters of weight row k0+kk
//   B[k][col=n]  : lane (j=l&15, kk=l>>4) reads 4 consecutive images of patch row k0+kk
// Because both the M (filter) and N (image) index of an MFMA may be permuted
// freely, component u of the A vector / t of the B vector feed MFMA (u,t): 16-byte
// loads along the contiguous dimensions and 16-byte stores, no LDS transposes.
// The k -> (channel,fy,fx) decode is tabulated once per workgroup in LDS.
#include <algorithm>
#include "common.h"

struct FwdParams {
    const float* images;
    const float* filters;
    const float* biases;
    float* targets;
    int N, imgStride, imgY, imgX, C, F, fs, mY, mX, pad, stride, groups;
    int conv, check, biasPerModule, neuron, mw;
    float sT, sO, na, nb;
    int xcdPair;  // k_filter_acts, K-split forms: launch order remapped so that a module's two image tiles share an XCD
    float* pool;  // k_conv_fwd_fast<.., POOL>: (F, mY/2, mX/2, N) max over the 2 x 2 module quads, written by the same launch
};

template <int FV>
struct AVec { float v[FV]; };

// k-steps (of 4) loaded per batch: all loads of a batch are issued before its MFMAs and the next batch is
// prefetched into a second register set while the current one is multiplied (the shapes here have at most a
// few waves per SIMD, so load latency has to be hidden inside the wave)
template <int FV, int NQ, int U, bool KSPLIT, bool VECN, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_filter_acts(const FwdParams p) {
    extern __shared__ int tbl[];
    constexpr int KB = 4 * U, T = FV * NQ * 4;
    const int fc = p.C / p.groups, fpg = p.F / p.groups, fp = p.fs * p.fs;
    const int K = fc * fp, Kpad = (K + KB - 1) / KB * KB, M = p.mY * p.mX;
    const int imgPixels = p.imgY * p.imgX;
    for (int k = threadIdx.x; k < Kpad; k += 64 * NW) {
        if (k < K) {
            const int c = k / fp, q = k % fp, fy = q / p.fs, fx = q % p.fs;
            tbl[2 * k] = c * imgPixels + fy * p.imgX + fx;
            tbl[2 * k + 1] = (fy << 16) | fx;
        } else {
            tbl[2 * k] = 0;
            tbl[2 * k + 1] = 0x40004000;  // far out of bounds: contributes zero
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler: scalar loop control, no exec-masked loops
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    // KSPLIT: the NW waves of the block share one module and take its k-batches round-robin (summed
    // through LDS below); otherwise each of four waves owns its own module.
    // KSPLIT with two image tiles (N = 128): both blocks of a module read the module's weight slab, the larger operand.  Blocks
    // go to the eight XCDs round-robin in launch order, so (tile 0, m) and (tile 1, m) -- neighbours in launch order -- sat on
    // different XCDs and both missed in their L2.  Launch order is remapped in groups of 16: block L of a group works on
    // module 8 g + L % 8, tile L / 8 -- the two tiles of a module are 8 launches apart, on the SAME XCD.
    int bx = blockIdx.x, by = blockIdx.y;
    if (KSPLIT && gridDim.x == 2 && p.xcdPair) {
        const int L = (int)blockIdx.x + 2 * (int)blockIdx.y, g = L >> 4, r = L & 15;
        if (g * 8 + 8 <= M) { by = g * 8 + (r & 7); bx = r >> 3; }
    }
    const int m = KSPLIT ? by : by * 4 + wave;
    if (!KSPLIT && m >= M) return;
    const int n0 = bx * (64 * NQ);
    const int f0 = blockIdx.z * (16 * FV);
    const int grp = f0 / fpg;
    const int y0 = p.pad + (m / p.mX) * p.stride, x0 = p.pad + (m % p.mX) * p.stride;
    const bool check = p.check != 0;

    const float* ib = p.images + ((ptrdiff_t)grp * fc * imgPixels + (ptrdiff_t)y0 * p.imgX + x0) * p.imgStride + n0 + 4 * j;
    const float* wb = p.filters + (size_t)(p.conv ? 0 : m) * K * p.F + f0 + FV * j;
    const float* safe = p.images;  // always-valid address for lanes whose element is outside the image / batch

    f32x4 acc[FV][NQ * 4];
#pragma unroll
    for (int u = 0; u < FV; u++)
#pragma unroll
        for (int t = 0; t < NQ * 4; t++) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // biases first: their latency hides behind the whole k-loop instead of stalling the epilogue
    float biasv[FV][4];
#pragma unroll
    for (int u = 0; u < FV; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = f0 + FV * (4 * kk + r) + u;
            biasv[u][r] = p.biases ? p.biases[p.biasPerModule ? (size_t)f * M + m : f] : 0.f;
        }

    AVec<FV> a0[U], a1[U];
    f32x4 b0[U][NQ], b1[U][NQ];
    auto load = [&](int kbase, AVec<FV>(&a)[U], f32x4(&b)[U][NQ]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int k = kbase + 4 * s + kk;
            const int off = tbl[2 * k], yx = tbl[2 * k + 1];
            const int kc = k < K ? k : K - 1;
            a[s] = *reinterpret_cast<const AVec<FV>*>(wb + (size_t)kc * p.F);
            bool inb = true;
            if (check) inb = (unsigned)(y0 + (yx >> 16)) < (unsigned)p.imgY && (unsigned)(x0 + (yx & 0xffff)) < (unsigned)p.imgX;
            const float* src = ib + (ptrdiff_t)off * p.imgStride;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int n = n0 + 64 * q + 4 * j;
                if (VECN) {
                    const bool ok = inb && n < p.N;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? src + 64 * q : safe);
                    b[s][q] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const bool ok = inb && n + t < p.N;
                        const float v = *(ok ? src + 64 * q + t : safe);
                        b[s][q][t] = ok ? v : 0.f;
                    }
                }
            }
        }
    };
    auto compute = [&](const AVec<FV>(&a)[U], const f32x4(&b)[U][NQ]) {
#pragma unroll
        for (int s = 0; s < U; s++)
#pragma unroll
            for (int u = 0; u < FV; u++)
#pragma unroll
                for (int q = 0; q < NQ; q++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[u][q * 4 + t] = mfma16(a[s].v[u], b[s][q][t], acc[u][q * 4 + t]);
    };
    const int kFirst = KSPLIT ? wave * KB : 0, kStep = KSPLIT ? NW * KB : KB;
    if (kFirst < K) load(kFirst, a0, b0);
    for (int kb = kFirst; kb < K; kb += 2 * kStep) {
        const bool more1 = kb + kStep < K, more2 = kb + 2 * kStep < K;
        if (more1) load(kb + kStep, a1, b1);
        compute(a0, b0);
        if (more1) {
            if (more2) load(kb + 2 * kStep, a0, b0);
            compute(a1, b1);
        }
    }
    // (an explicit one-batch-in-flight form -- unconditional loads, s_waitcnt vmcnt(0) before each issue, scheduling barriers, as
    // in k_weight_acts_wide -- was measured slower here: local3 10.7 -> 15.7 us, local4 7.4 -> 9.2, DeepID2+ local3 60.3 -> 62.9.  A
    // wave's share of K is one to four batches; the compiler's step-by-step waits inside a batch matter more than overlap.)

    bool mine[FV * NQ];  // which (filter vector, image quad) groups this wave stores
#pragma unroll
    for (int g = 0; g < FV * NQ; g++) mine[g] = true;
    if (KSPLIT) {
        // fixed-order sum of the NW waves' accumulators; group g = u*NQ + q is finished and stored by wave g % NW
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(tbl);
#pragma unroll
        for (int u = 0; u < FV; u++)
#pragma unroll
            for (int t = 0; t < NQ * 4; t++) red[(wave * T + u * NQ * 4 + t) * 64 + lane] = acc[u][t];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < FV; u++)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                mine[u * NQ + q] = ((u * NQ + q) % NW) == wave;
                if (mine[u * NQ + q]) {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int tile = u * NQ * 4 + q * 4 + t;
                        f32x4 sum = (red[(0 * T + tile) * 64 + lane] + red[(1 * T + tile) * 64 + lane]) +
                                    (red[(2 * T + tile) * 64 + lane] + red[(3 * T + tile) * 64 + lane]);
                        if (NW == 8)
                            sum += (red[(4 * T + tile) * 64 + lane] + red[(5 * T + tile) * 64 + lane]) +
                                   (red[(6 * T + tile) * 64 + lane] + red[(7 * T + tile) * 64 + lane]);
                        acc[u][q * 4 + t] = sum;
                    }
                }
            }
    }

    // D row (4*kk + r) of MFMA u  <->  filter f0 + FV*(4*kk + r) + u ; col j, MFMA t <-> image n0+64q+4j+t
#pragma unroll
    for (int u = 0; u < FV; u++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = f0 + FV * (4 * kk + r) + u;
            const float bias = biasv[u][r];
            float* trow = p.targets + ((size_t)f * M + m) * p.N;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                if (!mine[u * NQ + q]) continue;
                const int n = n0 + 64 * q + 4 * j;
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = p.sO * acc[u][q * 4 + t][r] + bias;
                if (p.neuron != MCN_NEURON_IDENT) {
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = neuron_apply(p.neuron, v[t], p.na, p.nb);
                }
                if (VECN) {
                    if (n < p.N) {
                        f32x4* dst = reinterpret_cast<f32x4*>(trow + n);
                        if (p.sT != 0.f) v += p.sT * (*dst);
                        *dst = v;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (n + t < p.N) trow[n + t] = (p.sT != 0.f ? p.sT * trow[n + t] : 0.f) + v[t];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shared-weight conv, fast path.  One wave owns 16*FV filters x 64 images x a strip of `mw` consecutive
// modules and walks (module, k-batch) as one software-pipelined sequence: the loads of step t+1 are in
// flight while step t multiplies, across module boundaries too.  All global accesses are buffer
// instructions (128-bit resource in SGPRs + 32-bit per-lane byte offset + scalar offset), so the per-lane
// offsets are computed once per wave and a load / store costs ONE instruction: at one or two waves per SIMD
// a wave retires a vector instruction every 4-8 cycles, and address arithmetic, not MFMA, was the bound.
// Out-of-range offsets (images beyond N, padded k, padding pixels in CHECK mode) read as zero / are dropped
// by the hardware range check.
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (hipcc of ROCm 7.2 lowers the b64 / b96 buffer-load builtins to ONE dword replicated into every component,
// so only the b32 and b128 forms are used)
template <int FV>
__device__ __forceinline__ AVec<FV> buf_load_avec(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    AVec<FV> a;
    if constexpr (FV == 4) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
#pragma unroll
        for (int i = 0; i < 4; i++) a.v[i] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < FV; i++) a.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + 4 * i, soff, 0));
    }
    return a;
}

//
// POOL: the launch also writes the 2 x 2 / stride 2 max pool of its output (MaxPooler of conv_util.cuh through convLocalPool,
// conv_util.cu; PoolLayer::fpropActs layer.cu:718-726), which otherwise is a launch of its own that reads the whole conv output
// again.  Modules are then numbered in quad order -- index g <-> quad g >> 2 (row-major over the (mY/2, mX/2) pooled map),
// corner g & 3 -- so that a strip of mw = 4 or 8 modules is one or two whole quads of ONE wave (running maximum in registers),
// and with mw = 1 or 2 the 4 / mw waves that share a quad sit in the same workgroup and combine through LDS after their strips.
// The conv output itself is stored as before (the backward pass needs it); max is exact, so both outputs are bitwise those of the
// two separate launches.
template <int FV, int UB, bool CHECK, bool RELU, bool POOL = false>
__global__ __launch_bounds__(256) void k_conv_fwd_fast(const FwdParams p) {
    extern __shared__ int tbl[];
    constexpr int KB = 4 * UB;
    const int fc = p.C / p.groups, fpg = p.F / p.groups, fp = p.fs * p.fs;
    const int K = fc * fp, nb = (K + KB - 1) / KB, Kpad = nb * KB, M = p.mY * p.mX;
    const int imgPixels = p.imgY * p.imgX;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        if (k < K) {
            const int c = k / fp, q = k % fp, fy = q / p.fs, fx = q % p.fs;
            tbl[2 * k] = (c * imgPixels + fy * p.imgX + fx) * p.imgStride * 4;  // byte offset of the patch row
            tbl[2 * k + 1] = (fy << 16) | fx;
        } else {
            tbl[2 * k] = 0;  // the weight row of a padded k is out of range and reads as zero
            tbl[2 * k + 1] = 0;
        }
    }
    int* arrived = tbl + 2 * Kpad + (p.mw == 2 ? 8 : 4) * FV * 4 * 64 * 4;  // POOL: per quad of this workgroup, how many of its waves have finished
    if (POOL && threadIdx.x < 4) arrived[threadIdx.x] = 0;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    // (mcn_xcd_group over the image / filter tiles of a module block was measured: conv2+ 29.7 -> 21.9 MB of L2 misses, no change in
    // time; DeepID2's conv1 / conv2 fetched MORE, 7.9 -> 9.5 MB, and the step lost 0.7 us.  Plain launch order.)
    const int mBeg = (blockIdx.y * 4 + wave) * p.mw, mEnd = min(M, mBeg + p.mw);
    if (mBeg >= M) return;
    const int n0 = blockIdx.x * 64, f0 = blockIdx.z * (16 * FV), grp = f0 / fpg;
    const int n = n0 + 4 * j;

    // resources: this group's image channels, the filters, the targets
    const size_t imgBytes = ((size_t)(fc * imgPixels - 1) * p.imgStride + p.N) * 4;
    const __amdgpu_buffer_rsrc_t rImg =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.images + (size_t)grp * fc * imgPixels * p.imgStride), 0, (int)imgBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.filters, 0, K * p.F * 4, 0x00020000);
    const unsigned OOB = 0xFFFFFFF0u;
    const unsigned laneB = n < p.N ? (unsigned)n * 4u : OOB;           // image quad of this lane
    const unsigned laneA = (unsigned)(kk * p.F + f0 + FV * j) * 4u;     // weight row k0+kk, filters f0+FV*j..

    float biasv[FV][4];
    unsigned outOff[FV][4];
#pragma unroll
    for (int u = 0; u < FV; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = f0 + FV * (4 * kk + r) + u;
            biasv[u][r] = p.biases ? p.biases[f] : 0.f;
            outOff[u][r] = (unsigned)(((size_t)f * M * p.N + n) * 4);
        }

    f32x4 acc[FV][4];
#pragma unroll
    for (int u = 0; u < FV; u++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    AVec<FV> a0[UB], a1[UB];
    f32x4 b0[UB], b1[UB];
    // load cursor (module, batch) and compute cursor; module coordinates are wave-uniform
    int lm = mBeg, lb = 0, lmy, lmx;
    int cm = mBeg, cb = 0, cmy, cmx;
    const int qX = p.mX >> 1;
    if (POOL) {
        const int quad = mBeg >> 2, sub = mBeg & 3;
        lmy = 2 * (quad / qX) + (sub >> 1);
        lmx = 2 * (quad % qX) + (sub & 1);
    } else {
        lmy = mBeg / p.mX;
        lmx = mBeg % p.mX;
    }
    cmy = lmy;
    cmx = lmx;
    // next module of the walk: row-major, or the corners of a quad and then the quad to the right / the first of the next quad row
    // (plain value arithmetic: with branches that update my OR mx the compiler selected between their ADDRESSES and moved the
    // cursors to scratch memory)
    auto advance = [&](int& g, int& my, int& mx) {
        g++;
        int nx, ny;
        if (POOL) {
            const int sub = g & 3;
            nx = mx + (sub == 2 ? -1 : 1);
            ny = my + (sub == 2 ? 1 : (sub == 0 ? -1 : 0));
            const bool wrap = sub == 0 && nx == p.mX;
            ny = wrap ? ny + 2 : ny;
            nx = wrap ? 0 : nx;
        } else {
            const bool wrap = mx + 1 == p.mX;
            nx = wrap ? 0 : mx + 1;
            ny = wrap ? my + 1 : my;
        }
        mx = nx;
        my = ny;
    };
    // POOL: per-module maxima go through LDS slot sets of FV * 4 entries (filter rows) x 64 lanes (in registers they would cost
    // 16 FV VGPRs across the whole k loop: conv1+ dropped from two waves per SIMD to one)
    f32x4* ex = reinterpret_cast<f32x4*>(tbl + 2 * Kpad) + lane;
    auto load = [&](AVec<FV>(&a)[UB], f32x4(&b)[UB]) {
        const int y0 = p.pad + lmy * p.stride, x0 = p.pad + lmx * p.stride;
        const int modBase = (y0 * p.imgX + x0) * p.imgStride * 4;  // may be negative with padding (CHECK)
        const int kbase = lb * KB;
#pragma unroll
        for (int s = 0; s < UB; s++) {
            const int k = kbase + 4 * s + kk;
            a[s] = buf_load_avec<FV>(rW, laneA, (unsigned)((kbase + 4 * s) * p.F * 4));
            unsigned voff = laneB + (unsigned)tbl[2 * k];
            if (CHECK) {
                const int yx = tbl[2 * k + 1];
                const bool inb = (unsigned)(y0 + (yx >> 16)) < (unsigned)p.imgY && (unsigned)(x0 + (yx & 0xffff)) < (unsigned)p.imgX;
                voff = (inb && laneB != OOB) ? voff + (unsigned)modBase : OOB;
                b[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rImg, voff, 0, 0));
            } else {
                b[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rImg, voff, (unsigned)modBase, 0));
            }
        }
        if (++lb == nb) {
            lb = 0;
            advance(lm, lmy, lmx);
        }
    };
    auto compute = [&](const AVec<FV>(&a)[UB], const f32x4(&b)[UB]) {
        // the first k-step of a module starts from C = 0 (an inline constant of the MFMA: no accumulator reset)
        if (cb == 0) {
#pragma unroll
            for (int u = 0; u < FV; u++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[u][t] = mfma16(a[0].v[u], b[0][t], (f32x4){0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
            for (int u = 0; u < FV; u++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[u][t] = mfma16(a[0].v[u], b[0][t], acc[u][t]);
        }
#pragma unroll
        for (int s = 1; s < UB; s++)
#pragma unroll
            for (int u = 0; u < FV; u++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[u][t] = mfma16(a[s].v[u], b[s][t], acc[u][t]);
        if (++cb == nb) {
            // module finished: bias, neuron, store (D row 4kk+r of MFMA u <-> filter f0+FV*(4kk+r)+u; col j, MFMA t <-> image n+t)
            // plain global stores (uniform module base + the lane's 32-bit offset): with buffer stores hipcc 7.2
            // reused the data registers in the very next VALU instruction and the last lanes stored the next row
            char* mbase = reinterpret_cast<char*>(p.targets + (size_t)(POOL ? cmy * p.mX + cmx : cm) * p.N);
            // POOL (all wave-uniform): strips of 4 / 8 modules keep the running maximum of the quad in the wave's slot set and store
            // the pooled rows with the quad's last module; strips of 1 / 2 modules write one slot set per module, no read-back
            const bool quadDone = POOL && p.mw >= 4 && (cm & 3) == 3;
            const bool rmw = POOL && p.mw >= 4 && (cm & 3) != 0;
            const int setBase = POOL ? (p.mw >= 4 ? wave : wave * p.mw + (cm - mBeg)) * (FV * 4) : 0;
            char* pbase = POOL ? reinterpret_cast<char*>(p.pool + (size_t)(cm >> 2) * p.N) : nullptr;
            const unsigned n4 = (unsigned)n * 4u;
            if (n < p.N) {
#pragma unroll
                for (int u = 0; u < FV; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        f32x4 v;
#pragma unroll
                        for (int t = 0; t < 4; t++) v[t] = p.sO * acc[u][t][r] + biasv[u][r];
                        if (RELU) {  // other neurons run as their own kernel after this one (host side)
#pragma unroll
                            for (int t = 0; t < 4; t++) v[t] = fmaxf(v[t], 0.f);
                        }
                        f32x4* dst = reinterpret_cast<f32x4*>(mbase + outOff[u][r]);
                        if (p.sT != 0.f) v += p.sT * *dst;
                        *dst = v;
                        if (POOL) {
                            f32x4* slot = ex + (setBase + u * 4 + r) * 64;
                            if (rmw) {
                                const f32x4 o = *slot;
#pragma unroll
                                for (int t = 0; t < 4; t++) v[t] = fmaxf(v[t], o[t]);
                            }
                            // (pooled map: M / 4 entries per filter -- the lane's offset follows from the conv output's)
                            if (quadDone) *reinterpret_cast<f32x4*>(pbase + (((outOff[u][r] - n4) >> 2) + n4)) = v;
                            else *slot = v;
                        }
                    }
            }
            cb = 0;
            advance(cm, cmy, cmx);
        }
    };
    {
        load(a0, b0);
        // (the explicit one-batch-in-flight loop of k_weight_acts_wide -- unconditional loads, s_waitcnt vmcnt(0) before each issue --
        // was measured here too: conv1 12.6 -> 14.6 us, conv2 14.4 -> 15.8, DeepID2+ conv2 unchanged; the compiler's step-by-step
        // waits inside a batch are worth more than the overlap between batches)
        while (true) {
            if (lm < mEnd) load(a1, b1);
            compute(a0, b0);
            if (cm >= mEnd) break;
            if (lm < mEnd) load(a0, b0);
            compute(a1, b1);
            if (cm >= mEnd) break;
        }
    }
    if constexpr (POOL) {
        if (p.mw >= 4) return;
        // mw = 1 / 2: the 4 / mw waves of a quad (consecutive waves of this workgroup: strips are aligned) have left the quad's four
        // modules in four consecutive slot sets, and whichever of them finishes LAST takes their maximum and stores the pooled rows.
        // (A barrier instead keeps every wave of the workgroup resident until the slowest one arrives, and conv1+ ran 5 us longer.)
        // LDS instructions execute in issue order, so a wave's slot writes are complete before its counter increment is.
        const int q = p.mw == 1 ? 0 : wave >> 1;  // which quad of this workgroup
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int before = 0;
        if (lane == 0) before = __hip_atomic_fetch_add(arrived + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        before = __builtin_amdgcn_readfirstlane(before);
        asm volatile("" ::: "memory");
        if (before != 4 / p.mw - 1 || n >= p.N) return;
        const f32x4* sets = ex + q * 4 * (FV * 4) * 64;
        char* pbase = reinterpret_cast<char*>(p.pool + (size_t)(mBeg >> 2) * p.N);
        const unsigned n4 = (unsigned)n * 4u;
#pragma unroll
        for (int u = 0; u < FV; u++) {
            f32x4 c[4][4];  // every read of the component in flight before the first maximum
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int g = 0; g < 4; g++) c[r][g] = sets[(g * (FV * 4) + u * 4 + r) * 64];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                f32x4 m4;
#pragma unroll
                for (int t = 0; t < 4; t++) m4[t] = fmaxf(fmaxf(c[r][0][t], c[r][1][t]), fmaxf(c[r][2][t], c[r][3][t]));
                *reinterpret_cast<f32x4*>(pbase + (((outOff[u][r] - n4) >> 2) + n4)) = m4;
            }
        }
    }
}

template <int FV, int UB>
static void launch_fast2(const FwdParams& p, dim3 grid, size_t lds, hipStream_t s) {
    const bool relu = p.neuron == MCN_NEURON_RELU;
    if (p.pool) {  // (never with check: the host takes this form only when the modules tile the image exactly)
        if (relu) MCN_LAUNCH((k_conv_fwd_fast<FV, UB, false, true, true>), grid, dim3(256), lds, s, p);
        else MCN_LAUNCH((k_conv_fwd_fast<FV, UB, false, false, true>), grid, dim3(256), lds, s, p);
    } else if (p.check) {
        if (relu) MCN_LAUNCH((k_conv_fwd_fast<FV, UB, true, true>), grid, dim3(256), lds, s, p);
        else MCN_LAUNCH((k_conv_fwd_fast<FV, UB, true, false>), grid, dim3(256), lds, s, p);
    } else {
        if (relu) MCN_LAUNCH((k_conv_fwd_fast<FV, UB, false, true>), grid, dim3(256), lds, s, p);
        else MCN_LAUNCH((k_conv_fwd_fast<FV, UB, false, false>), grid, dim3(256), lds, s, p);
    }
}

template <int FV, int NQ, int U>
static void launch_fwd3(const FwdParams& p, dim3 grid, size_t lds, bool vecn, bool ksplit, hipStream_t s) {
    if (ksplit && vecn && p.mw >= 8) {  // (mw doubles as wave count + batch size of the split form: 8 + U)
        if constexpr (NQ == 1 && U <= 8 && FV <= 3) {
            const int u8 = p.mw - 8;
            if (u8 == 9) MCN_LAUNCH((k_filter_acts<FV, NQ, 9, true, true, 8>), grid, dim3(512), lds, s, p);
            else if (u8 == 6) MCN_LAUNCH((k_filter_acts<FV, NQ, 6, true, true, 8>), grid, dim3(512), lds, s, p);
            else if (u8 == 8) MCN_LAUNCH((k_filter_acts<FV, NQ, 8, true, true, 8>), grid, dim3(512), lds, s, p);
            else MCN_LAUNCH((k_filter_acts<FV, NQ, 4, true, true, 8>), grid, dim3(512), lds, s, p);
            return;
        }
    }
    if (ksplit) {
        if (vecn) MCN_LAUNCH((k_filter_acts<FV, NQ, U, true, true>), grid, dim3(256), lds, s, p);
        else MCN_LAUNCH((k_filter_acts<FV, NQ, U, true, false>), grid, dim3(256), lds, s, p);
    } else {
        if (vecn) MCN_LAUNCH((k_filter_acts<FV, NQ, U, false, true>), grid, dim3(256), lds, s, p);
        else MCN_LAUNCH((k_filter_acts<FV, NQ, U, false, false>), grid, dim3(256), lds, s, p);
    }
}

// k-steps per batch: 12 when the tile is small enough to hold two such register sets and K divides (conv1's
// K = 48 then needs a single round trip), else 8, or 4 for the widest tiles
static int fwd_batch(int FV, int NQ, int K) {
    if (FV * NQ >= 4) return 4;
    if (FV * NQ <= 2 && K % 48 == 0 && K % 32 != 0) return 12;
    return 8;
}

template <int FV, int NQ>
static void launch_fwd2(const FwdParams& p, int U, dim3 grid, size_t lds, bool vecn, bool ksplit, hipStream_t s) {
    if constexpr (FV * NQ >= 4) {
        launch_fwd3<FV, NQ, 4>(p, grid, lds, vecn, ksplit, s);
    } else if constexpr (FV * NQ <= 2) {
        if (U == 12) launch_fwd3<FV, NQ, 12>(p, grid, lds, vecn, ksplit, s);
        else launch_fwd3<FV, NQ, 8>(p, grid, lds, vecn, ksplit, s);
    } else {
        launch_fwd3<FV, NQ, 8>(p, grid, lds, vecn, ksplit, s);
    }
}

template <int FV>
static void launch_fwd1(const FwdParams& p, int nq, int U, dim3 grid, size_t lds, bool vecn, bool ksplit, hipStream_t s) {
    if (nq == 2) launch_fwd2<FV, 2>(p, U, grid, lds, vecn, ksplit, s);
    else launch_fwd2<FV, 1>(p, U, grid, lds, vecn, ksplit, s);
}

static int filter_acts_impl(const float* images, const float* filters, const float* biases, float* targets,
                            int N, int imgStride, int imgY, int imgX, int C, int F, int fs, int mY, int mX,
                            int pad, int stride, int groups, int conv, float sT, float sO, int biasPerModule,
                            int neuron, float na, float nb, void* stream, float* pool = nullptr, bool probe = false) {
    // host asserts of _filterActs (filter_acts.cu:577-614), reported instead of aborting
    MCN_REQUIRE(images && filters && targets, "filterActs: null pointer");
    MCN_REQUIRE(N > 0 && imgY > 0 && imgX > 0 && C > 0 && F > 0 && fs > 0 && mY > 0 && mX > 0 && stride > 0 && groups > 0,
                "filterActs: non-positive dimension");
    MCN_REQUIRE(imgStride >= N, "filterActs: imgStride (%d) < numImages (%d)", imgStride, N);
    MCN_REQUIRE(C % groups == 0, "filterActs: numImgColors %% numGroups != 0");
    MCN_REQUIRE(F % (16 * groups) == 0, "filterActs: numFilters (%d) must be divisible by 16*numGroups", F);
    MCN_REQUIRE(groups > 1 || (C <= 3 || C % 2 == 0), "filterActs: numImgColors must be <= 3 or even");
    MCN_REQUIRE(groups == 1 || (C / groups) % 2 == 0, "filterActs: filter colors must be even with groups");
    MCN_REQUIRE(pad <= 0, "filterActs: paddingStart must be <= 0");
    MCN_REQUIRE(pad + (mX - 1) * stride + fs >= imgX && pad + (mY - 1) * stride + fs >= imgY,
                "filterActs: modules do not cover the image");
    MCN_REQUIRE(stride <= fs, "filterActs: moduleStride > filterSize");
    MCN_REQUIRE(neuron_valid(neuron), "filterActs: bad neuron type %d", neuron);

    FwdParams p;
    p.images = images; p.filters = filters; p.biases = biases; p.targets = targets;
    p.N = N; p.imgStride = imgStride; p.imgY = imgY; p.imgX = imgX; p.C = C; p.F = F; p.fs = fs;
    p.mY = mY; p.mX = mX; p.pad = pad; p.stride = stride; p.groups = groups;
    p.conv = conv; p.biasPerModule = biasPerModule; p.neuron = neuron;
    p.sT = sT; p.sO = sO; p.na = na; p.nb = nb;
    static const int xcdPairEnv = getenv("MCN_FA_XCD_PAIR") ? atoi(getenv("MCN_FA_XCD_PAIR")) : 1;
    p.xcdPair = xcdPairEnv;
    p.pool = nullptr;
    const int fc = C / groups, K = fc * fs * fs, M = mY * mX, fpg = F / groups;
    p.check = (pad < 0) || (pad + (mX - 1) * stride + fs > imgX) || (pad + (mY - 1) * stride + fs > imgY) || (K % 4 != 0);

    const bool vecn = (N % 4 == 0) && (imgStride % 4 == 0) && (((uintptr_t)images | (uintptr_t)targets) % 16 == 0);
    const int ftiles = fpg / 16;
    p.mw = 1;
    // fast path (shared weights, 16-byte image accesses, one bias per filter, offsets within 2 GB)
    if (conv && vecn && !(biases && biasPerModule) && ((uintptr_t)filters % 16 == 0) &&
        (neuron == MCN_NEURON_IDENT || neuron == MCN_NEURON_RELU) &&
        (size_t)C * imgY * imgX * imgStride * 4 < (1u << 31) && (size_t)F * M * N * 4 < (1u << 31) && K * F * 4 < (1 << 30)) {
        p.check = (pad < 0) || (pad + (mX - 1) * stride + fs > imgX) || (pad + (mY - 1) * stride + fs > imgY);
        const int ntiles = divup(N, 64);
        int FV = 1;
        for (int fv = 4; fv > 1; fv--)
            if (ftiles % fv == 0 && (long)ntiles * M * (F / (16 * fv)) >= 2048) { FV = fv; break; }
        const long waves1 = (long)ntiles * M * (F / (16 * FV));
        p.mw = (int)std::max<long>(1, std::min<long>(8, waves1 / 2560));
        if (pool || probe) {  // the fused 2 x 2 max pool: whole quads, strips of 1, 2, 4 or 8 modules in quad order
            if (p.check || (mY & 1) || (mX & 1) || sT != 0.f) {
                if (probe) return 0;
                MCN_REQUIRE(false, "filterActs + pool: the modules must tile the image exactly in an even number of rows and columns");
            }
            if (probe) return 1;
            p.pool = pool;
            p.mw = p.mw >= 8 ? 8 : p.mw >= 4 ? 4 : p.mw;
            if (p.mw == 3) p.mw = 2;
        }
        const int S = divup(K, 4);
        const int UB = (FV <= 2 && S % 12 == 0) ? 12 : (FV == 4 ? 6 : (FV == 1 ? 16 : 8));
        dim3 grid(ntiles, divup(divup(M, p.mw), 4), F / (16 * FV));
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "filterActs: grid too large");
        size_t lds = (size_t)divup(K, 4 * UB) * 4 * UB * 8;
        if (p.pool) lds += (size_t)(p.mw == 2 ? 8 : 4) * FV * 4 * 64 * 16 + 16;  // one slot set per module of the block's quads (per wave for whole-quad strips) + the arrival counters
        MCN_REQUIRE(lds <= 136 * 1024, "filterActs: filter too large for the decode table (%zu bytes)", lds);
        hipStream_t s = (hipStream_t)stream;
        if (FV == 4) launch_fast2<4, 6>(p, grid, lds, s);
        else if (FV == 3) launch_fast2<3, 8>(p, grid, lds, s);
        else if (FV == 2 && UB == 12) launch_fast2<2, 12>(p, grid, lds, s);
        else if (FV == 2) launch_fast2<2, 8>(p, grid, lds, s);
        else if (UB == 12) launch_fast2<1, 12>(p, grid, lds, s);
        else launch_fast2<1, 16>(p, grid, lds, s);
        return MCN_LAUNCH_CHECK("filterActs");
    }
    if (probe) return 0;
    MCN_REQUIRE(!pool, "filterActs + pool: shape outside the fused form (mcn_filter_acts_pool_supported)");
    auto best_fv = [&](int cap) {
        for (int fv = cap; fv > 1; fv--)
            if (ftiles % fv == 0) return fv;
        return 1;
    };
    int FV = best_fv(4), NQ = N > 64 ? 2 : 1;
    bool ksplit = false;
    auto waves = [&](int fv, int nq) { return (long)divup(N, 64 * nq) * M * (F / (16 * fv)); };
    // big unshared layers (DeepID2+ local3: 100 modules x 128 filters, K = 1152): 64 filters x 64 images per block, K split over its
    // four waves -- 16-byte weight loads and 16 MFMAs per k-step and load pair instead of 4 (measured 75.9 -> 60.3 us; the
    // eight-wave split with 32 filters: 81 us).  MCN_FA_BIG_LOCAL=0 restores the one-wave tiles.
    static const bool bigLocalOn = !(getenv("MCN_FA_BIG_LOCAL") && atoi(getenv("MCN_FA_BIG_LOCAL")) == 0);
    const bool bigLocal = bigLocalOn && !conv && vecn && K >= 512 && ftiles % 4 == 0 && waves(1, 1) * 4 > 4096 && waves(1, 1) * 4 <= 8192;
    if ((waves(1, 1) * 4 <= 4096 && K >= 128) || bigLocal) {
        // few modules (the unshared local layers): one block per (module, filter vector, 64 images), its four
        // waves split K.  Wide filter vectors keep the weight rows read in long contiguous pieces.
        ksplit = true;
        NQ = 1;
        FV = bigLocal ? 4 : best_fv(3);
    } else {
        // tile choice: largest filter vector that divides the group's filter tiles, then shrink until
        // the launch has at least one wave per SIMD (1024 on MI355X)
        if (waves(FV, NQ) < 1024 && NQ == 2) NQ = 1;
        while (waves(FV, NQ) < 1024 && FV > 1) FV = best_fv(FV - 1);
    }
    int U = fwd_batch(FV, NQ, K);
    // split form with 8 waves and batches of 4 k-steps: a wave's chain of dependent round trips is what a launch this small
    // costs (local3: 200 workgroups, K = 288 -> 18 batches over 8 waves; measured 11.5 -> see DESIGN.md)
    static const bool wide = getenv("MCN_FA_NW4") == nullptr;  // (A/B switch)
    const bool eight = ksplit && vecn && wide && U <= 8 && K >= 128 && FV <= 3;
    if (eight) {
        // batch = a wave's whole share of K when that fits the registers (one round trip per wave), else 8 / 4
        const int share = divup(divup(K, 4), 8);
        U = share <= 4 ? 4 : share <= 6 ? 6 : share <= 9 ? 9 : (divup(K, 4) % 64 == 0 ? 8 : 9);
        p.mw = 8 + U;
    }
    const int KB = 4 * U;
    if (K % KB != 0) p.check = 1;  // the padded k of the last batch must load zeros
    dim3 grid(divup(N, 64 * NQ), ksplit ? M : divup(M, 4), F / (16 * FV));
    MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "filterActs: grid too large");
    size_t lds = (size_t)divup(K, KB) * KB * 8;
    if (ksplit) lds = std::max(lds, (size_t)(eight ? 8 : 4) * FV * NQ * 4 * 64 * 16);
    MCN_REQUIRE(lds <= (eight ? 128 : 64) * 1024, "filterActs: filter too large for the decode table (%zu bytes)", lds);
    hipStream_t s = (hipStream_t)stream;
    switch (FV) {
        case 4: launch_fwd1<4>(p, NQ, U, grid, lds, vecn, ksplit, s); break;
        case 3: launch_fwd1<3>(p, NQ, U, grid, lds, vecn, ksplit, s); break;
        case 2: launch_fwd1<2>(p, NQ, U, grid, lds, vecn, ksplit, s); break;
        default: launch_fwd1<1>(p, NQ, U, grid, lds, vecn, ksplit, s); break;
    }
    return MCN_LAUNCH_CHECK("filterActs");
}

MCN_API int mcn_conv_filter_acts(const float* images, const float* filters, float* targets, int numImages,
                                 int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                 int filterSize, int numModulesY, int numModulesX, int paddingStart,
                                 int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return filter_acts_impl(images, filters, nullptr, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                            numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups, 1,
                            scaleTargets, scaleOutput, 0, MCN_NEURON_IDENT, 0.f, 0.f, stream);
}

MCN_API int mcn_local_filter_acts(const float* images, const float* filters, float* targets, int numImages,
                                  int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                  int filterSize, int numModulesY, int numModulesX, int paddingStart,
                                  int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return filter_acts_impl(images, filters, nullptr, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                            numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups, 0,
                            scaleTargets, scaleOutput, 0, MCN_NEURON_IDENT, 0.f, 0.f, stream);
}

MCN_API int mcn_filter_acts_bias_neuron(const float* images, const float* filters, const float* biases, float* targets,
                                        int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                        int numFilters, int filterSize, int numModulesY, int numModulesX,
                                        int paddingStart, int moduleStride, int numGroups, int conv,
                                        int biasPerModule, int neuron, float neuronA, float neuronB, void* stream) {
    return filter_acts_impl(images, filters, biases, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                            numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups,
                            conv ? 1 : 0, 0.f, 1.f, biasPerModule, neuron, neuronA, neuronB, stream);
}

// filterActs + bias + neuron + the 2 x 2 / stride 2 / start 0 max pool of the result in one launch (shared-weight conv layers whose
// modules tile the image; ConvLayer::fpropActs layer.cu:443-462 followed by PoolLayer::fpropActs layer.cu:718-726).  targets gets
// the (numFilters, modules, numImages) conv output, poolTargets the (numFilters, numModulesY/2, numModulesX/2, numImages) maxima.
MCN_API int mcn_filter_acts_pool_supported(const float* images, const float* filters, const float* biases, const float* targets,
                                           int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                           int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                           int numGroups, int conv, int biasPerModule, int neuron) {
    if (!images || !filters || !targets || numImages <= 0 || numGroups <= 0 || numImgColors % numGroups != 0 ||
        numFilters % (16 * numGroups) != 0 || !neuron_valid(neuron))
        return 0;
    return filter_acts_impl(images, filters, biases, const_cast<float*>(targets), numImages, imgStride, imgSizeY, imgSizeX, numImgColors,
                            numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups, conv ? 1 : 0, 0.f,
                            1.f, biasPerModule, neuron, 0.f, 0.f, nullptr, nullptr, true) == 1 ? 1 : 0;
}

MCN_API int mcn_filter_acts_bias_neuron_pool(const float* images, const float* filters, const float* biases, float* targets,
                                             float* poolTargets, int numImages, int imgStride, int imgSizeY, int imgSizeX,
                                             int numImgColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                             int paddingStart, int moduleStride, int numGroups, int conv, int biasPerModule, int neuron,
                                             float neuronA, float neuronB, void* stream) {
    MCN_REQUIRE(poolTargets && ((uintptr_t)poolTargets % 16) == 0, "filterActs + pool: pool target missing or not 16-byte aligned");
    return filter_acts_impl(images, filters, biases, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors, numFilters,
                            filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups, conv ? 1 : 0, 0.f, 1.f,
                            biasPerModule, neuron, neuronA, neuronB, stream, poolTargets, false);
}
